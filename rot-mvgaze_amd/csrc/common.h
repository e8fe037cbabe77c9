// Internal helpers shared by the HIP translation units of librotmvgaze_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rotmvgaze.h"

namespace mvg {

void set_error(const char *fmt, ...);
bool prof_on();
// Bracket a launch with events when profiling is enabled.
void prof_begin(int family, hipStream_t s, double flops, double bytes);
void prof_end(int family, hipStream_t s);

// The workspace the caller registered for this (device, stream) with mvg_set_scratch, if it holds `floats`
// floats; nullptr otherwise (the caller of this function then takes its scratch-free form).  Uses on a stream are
// ordered by the stream, so every kernel sequence that finishes with its scratch before the next launch on that
// stream may share it (stream-K pieces, bn_finalize slices).  The library allocates nothing.
float *stream_scratch(hipStream_t st, size_t floats);
// device CUs minus mvg_set_reserved_cus(): what stream-K grids, wgrad splits and split-K plan for
int compute_cus();

// bn.hip: merge [groups][chunks][2][c] partial sums (s1, s2) of a BatchNorm backward in fp64, in a fixed order;
// dgamma / dbeta (+)= the sums over the groups.  Host launcher shared with the fused backward-data path.
// mx != nullptr: the partials carry a third row, max |dz| per channel, merged into mx [groups][c].
// raw_mean / raw_invstd != nullptr: the partials' second row is the uncentred sum(dz * y); s2 = invstd * (it - mean * s1).
int bn_bwd_finalize_launch(const float *partial, int groups, int chunks, int c, float *s1, float *s2, float *dgamma,
                           float *dbeta, int accumulate, hipStream_t st, float *mx = nullptr, const float *raw_mean = nullptr,
                           const float *raw_invstd = nullptr, const float *gamma = nullptr, const float *invstd = nullptr,
                           long long rows = 0, float *dy_sinv = nullptr);
// (gamma, invstd, rows, dy_sinv: split path - the same launch also leaves *dy_sinv = 2^-k for the unit's dy, from the bound
// max |gamma invstd| (mx + |s1| / rows + sqrt(rows) |s2| / rows): what bn_dy_scale_kernel computes)

struct ProfScope {
  int fam;
  hipStream_t s;
  bool on;
  ProfScope(int family, hipStream_t st, double flops, double bytes) : fam(family), s(st), on(prof_on()) {
    if (on) prof_begin(fam, s, flops, bytes);
  }
  ~ProfScope() {
    if (on) prof_end(fam, s);
  }
};

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

#define MVG_REQUIRE(cond, ...)       \
  do {                               \
    if (!(cond)) {                   \
      mvg::set_error(__VA_ARGS__);   \
      return 2;                      \
    }                                \
  } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// wave-level sum over 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace mvg
