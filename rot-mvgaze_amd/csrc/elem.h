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

// "s3" storage (conv_split.hip): an fp32 value as the exact sum of three bf16 pieces; channels in chunks of 8 with
// the three pieces of a chunk adjacent (48 bytes).  The streaming passes address 4-channel groups like fp32: group
// i4 is half (i4 & 1) of chunk (i4 >> 1), i.e. three 8-byte accesses 16 bytes apart.
struct s3_t {
  unsigned short v;
};

__device__ __forceinline__ void split3(float v, unsigned short &p1, unsigned short &p2, unsigned short &p3) {
  const __bf16 h1 = (__bf16)v;
  float r = v - (float)h1;
  const __bf16 h2 = (__bf16)r;
  r -= (float)h2;
  const __bf16 h3 = (__bf16)r;
  p1 = __builtin_bit_cast(unsigned short, h1);
  p2 = __builtin_bit_cast(unsigned short, h2);
  p3 = __builtin_bit_cast(unsigned short, h3);
}

// 8 consecutive fp32 values -> the chunk's three 16-byte piece vectors
__device__ __forceinline__ void split3_chunk(const float (&v)[8], uint4 &q1, uint4 &q2, uint4 &q3) {
  unsigned short a[8], b[8], c[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) split3(v[k], a[k], b[k], c[k]);
  q1 = make_uint4((unsigned)a[0] | ((unsigned)a[1] << 16), (unsigned)a[2] | ((unsigned)a[3] << 16), (unsigned)a[4] | ((unsigned)a[5] << 16),
             (unsigned)a[6] | ((unsigned)a[7] << 16));
  q2 = make_uint4((unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16), (unsigned)b[4] | ((unsigned)b[5] << 16),
             (unsigned)b[6] | ((unsigned)b[7] << 16));
  q3 = make_uint4((unsigned)c[0] | ((unsigned)c[1] << 16), (unsigned)c[2] | ((unsigned)c[3] << 16), (unsigned)c[4] | ((unsigned)c[5] << 16),
             (unsigned)c[6] | ((unsigned)c[7] << 16));
}

template <>
struct Elem<s3_t> {
  static constexpr double kBytes = 6.0;
  static constexpr int W = 1;
  static __device__ __forceinline__ float4 ld4(const s3_t *p, long long i4) {
    const uint2 *q = reinterpret_cast<const uint2 *>(p) + (i4 >> 1) * 6 + (i4 & 1);
    const uint2 a = q[0], b = q[2], c = q[4];
    return make_float4((bf16_lo(a.x) + bf16_lo(b.x)) + bf16_lo(c.x), (bf16_hi(a.x) + bf16_hi(b.x)) + bf16_hi(c.x),
                       (bf16_lo(a.y) + bf16_lo(b.y)) + bf16_lo(c.y), (bf16_hi(a.y) + bf16_hi(b.y)) + bf16_hi(c.y));
  }
  static __device__ __forceinline__ void st4(s3_t *p, long long i4, float4 v) {
    unsigned short a[4], b[4], c[4];
    split3(v.x, a[0], b[0], c[0]);
    split3(v.y, a[1], b[1], c[1]);
    split3(v.z, a[2], b[2], c[2]);
    split3(v.w, a[3], b[3], c[3]);
    uint2 *q = reinterpret_cast<uint2 *>(p) + (i4 >> 1) * 6 + (i4 & 1);
    q[0] = make_uint2((unsigned)a[0] | ((unsigned)a[1] << 16), (unsigned)a[2] | ((unsigned)a[3] << 16));
    q[2] = make_uint2((unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16));
    q[4] = make_uint2((unsigned)c[0] | ((unsigned)c[1] << 16), (unsigned)c[2] | ((unsigned)c[3] << 16));
  }
  static __device__ __forceinline__ void ldw(const s3_t *p, long long iw, float4 (&v)[1]) { v[0] = ld4(p, iw); }
  static __device__ __forceinline__ void stw(s3_t *p, long long iw, const float4 (&v)[1]) { st4(p, iw, v[0]); }
};

}  // namespace mvg
