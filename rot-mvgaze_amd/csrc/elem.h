// Element access for the HBM-bound passes that run in both storage types: fp32 (the parity path) and
// bf16 (config C5's path: activations and their gradients are bf16 in HBM, arithmetic stays fp32).
// A "vec" is 4 consecutive channels: a float4 (16 bytes) or 4 packed bf16 (8 bytes).
#pragma once
#include <hip/hip_runtime.h>

namespace mvg {

typedef unsigned short bf16_t;      // storage type of a bf16 value

__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xFFFF0000u); }
__device__ __forceinline__ unsigned bf16_pack2(float a, float b) {       // round-to-nearest-even; a NaN stays a NaN
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

template <typename T>
struct Elem;

// W = float4 groups per 16-byte access: the streaming passes move 16 bytes per lane in both storage types
// (fp32: 4 channels, bf16: 8 channels); ldw / stw address W-group units.
template <>
struct Elem<float> {
  static constexpr double kBytes = 4.0;
  static constexpr int W = 1;
  static __device__ __forceinline__ void ldw(const float *p, long long iw, float4 (&v)[1]) { v[0] = reinterpret_cast<const float4 *>(p)[iw]; }
  static __device__ __forceinline__ void stw(float *p, long long iw, const float4 (&v)[1]) { reinterpret_cast<float4 *>(p)[iw] = v[0]; }
  static __device__ __forceinline__ float4 ld4(const float *p, long long i4) { return reinterpret_cast<const float4 *>(p)[i4]; }
  static __device__ __forceinline__ void st4(float *p, long long i4, float4 v) { reinterpret_cast<float4 *>(p)[i4] = v; }
  static __device__ __forceinline__ float ld1(const float *p, long long i) { return p[i]; }
};

template <>
struct Elem<bf16_t> {
  static constexpr double kBytes = 2.0;
  static constexpr int W = 2;
  static __device__ __forceinline__ void ldw(const bf16_t *p, long long iw, float4 (&v)[2]) {
    const uint4 u = reinterpret_cast<const uint4 *>(p)[iw];
    v[0] = make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
    v[1] = make_float4(bf16_lo(u.z), bf16_hi(u.z), bf16_lo(u.w), bf16_hi(u.w));
  }
  static __device__ __forceinline__ void stw(bf16_t *p, long long iw, const float4 (&v)[2]) {
    reinterpret_cast<uint4 *>(p)[iw] = make_uint4(bf16_pack2(v[0].x, v[0].y), bf16_pack2(v[0].z, v[0].w), bf16_pack2(v[1].x, v[1].y),
                                                  bf16_pack2(v[1].z, v[1].w));
  }
  static __device__ __forceinline__ float4 ld4(const bf16_t *p, long long i4) {
    const uint2 u = reinterpret_cast<const uint2 *>(p)[i4];
    return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
  }
  static __device__ __forceinline__ void st4(bf16_t *p, long long i4, float4 v) {
    reinterpret_cast<uint2 *>(p)[i4] = make_uint2(bf16_pack2(v.x, v.y), bf16_pack2(v.z, v.w));
  }
  static __device__ __forceinline__ float ld1(const bf16_t *p, long long i) { return __uint_as_float((unsigned)p[i] << 16); }
};

// "sp" storage (conv_split.hip's operand format): an fp32 value v - times a per-tensor power-of-two scale chosen by
// the producer, so that the pieces stay inside fp16's range - as TWO fp16 pieces, h1 = fp16(v), h2 = fp16(v - h1)
// (round to nearest even; v - h1 is exact in fp32).  |v - h1 - h2| <= 2^-23 |v| whenever h2 is a normal fp16 number:
// at most the LAST of the 24 significand bits is lost (the remainder v - h1 has up to 12 significant bits, fp16 keeps
// 11), and three values in four are represented exactly; rms error 0.74 * 2^-24 |v|.  Channels in chunks of 8 with the two pieces of a chunk
// adjacent (32 bytes, 4 bytes per element).  The streaming passes address 4-channel groups like fp32: group i4 is
// half (i4 & 1) of chunk (i4 >> 1), i.e. two 8-byte accesses 16 bytes apart.
struct sp_t {
  unsigned short v;
};
constexpr int SP_NP = 2;            // pieces per value
constexpr int SP_BYTES = 2 * SP_NP; // bytes per element

__device__ __forceinline__ void split2(float v, unsigned short &p1, unsigned short &p2) {
  const _Float16 h1 = (_Float16)v;
  const _Float16 h2 = (_Float16)(v - (float)h1);
  p1 = __builtin_bit_cast(unsigned short, h1);
  p2 = __builtin_bit_cast(unsigned short, h2);
}
__device__ __forceinline__ float h_lo(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xFFFFu)); }
__device__ __forceinline__ float h_hi(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); }

// 8 consecutive fp32 values -> the chunk's two 16-byte piece vectors
__device__ __forceinline__ void split2_chunk(const float (&v)[8], uint4 &q1, uint4 &q2) {
  unsigned short a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) split2(v[k], a[k], b[k]);
  q1 = make_uint4((unsigned)a[0] | ((unsigned)a[1] << 16), (unsigned)a[2] | ((unsigned)a[3] << 16), (unsigned)a[4] | ((unsigned)a[5] << 16),
             (unsigned)a[6] | ((unsigned)a[7] << 16));
  q2 = make_uint4((unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16), (unsigned)b[4] | ((unsigned)b[5] << 16),
             (unsigned)b[6] | ((unsigned)b[7] << 16));
}
// ... and back (the sum of the two pieces: exact in fp32)
__device__ __forceinline__ void merge2_chunk(const uint4 &q1, const uint4 &q2, float (&v)[8]) {
  const unsigned a[4] = {q1.x, q1.y, q1.z, q1.w}, b[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = h_lo(a[k]) + h_lo(b[k]);
    v[2 * k + 1] = h_hi(a[k]) + h_hi(b[k]);
  }
}

// The power-of-two scale that maps `bound` (>= every |value| of the tensor) just below 2^15: values down to 2^-17 of the
// bound then keep the 2^-23 relative accuracy; smaller ones are off by at most 2^-25 / scale = 2^-40 of the bound.
__device__ __forceinline__ float sp_scale_for(float bound) {
  if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(bound, &e);                      // bound = m * 2^e, m in [0.5, 1)
  int k = 15 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return ldexpf(1.f, k);
}

template <>
struct Elem<sp_t> {
  static constexpr double kBytes = (double)SP_BYTES;
  static constexpr int W = 1;
  static __device__ __forceinline__ float4 ld4(const sp_t *p, long long i4) {
    const uint2 *q = reinterpret_cast<const uint2 *>(p) + (i4 >> 1) * 4 + (i4 & 1);
    const uint2 a = q[0], b = q[2];
    return make_float4(h_lo(a.x) + h_lo(b.x), h_hi(a.x) + h_hi(b.x), h_lo(a.y) + h_lo(b.y), h_hi(a.y) + h_hi(b.y));
  }
  static __device__ __forceinline__ void st4(sp_t *p, long long i4, float4 v) {
    unsigned short a[4], b[4];
    split2(v.x, a[0], b[0]);
    split2(v.y, a[1], b[1]);
    split2(v.z, a[2], b[2]);
    split2(v.w, a[3], b[3]);
    uint2 *q = reinterpret_cast<uint2 *>(p) + (i4 >> 1) * 4 + (i4 & 1);
    q[0] = make_uint2((unsigned)a[0] | ((unsigned)a[1] << 16), (unsigned)a[2] | ((unsigned)a[3] << 16));
    q[2] = make_uint2((unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16));
  }
  static __device__ __forceinline__ void ldw(const sp_t *p, long long iw, float4 (&v)[1]) { v[0] = ld4(p, iw); }
  static __device__ __forceinline__ void stw(sp_t *p, long long iw, const float4 (&v)[1]) { st4(p, iw, v[0]); }
};

}  // namespace mvg
