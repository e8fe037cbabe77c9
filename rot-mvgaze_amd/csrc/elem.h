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

template <>
struct Elem<float> {
  static constexpr double kBytes = 4.0;
  static __device__ __forceinline__ float4 ld4(const float *p, long long i4) { return reinterpret_cast<const float4 *>(p)[i4]; }
  static __device__ __forceinline__ void st4(float *p, long long i4, float4 v) { reinterpret_cast<float4 *>(p)[i4] = v; }
  static __device__ __forceinline__ float ld1(const float *p, long long i) { return p[i]; }
};

template <>
struct Elem<bf16_t> {
  static constexpr double kBytes = 2.0;
  static __device__ __forceinline__ float4 ld4(const bf16_t *p, long long i4) {
    const uint2 u = reinterpret_cast<const uint2 *>(p)[i4];
    return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
  }
  static __device__ __forceinline__ void st4(bf16_t *p, long long i4, float4 v) {
    reinterpret_cast<uint2 *>(p)[i4] = make_uint2(bf16_pack2(v.x, v.y), bf16_pack2(v.z, v.w));
  }
  static __device__ __forceinline__ float ld1(const bf16_t *p, long long i) { return __uint_as_float((unsigned)p[i] << 16); }
};

}  // namespace mvg
