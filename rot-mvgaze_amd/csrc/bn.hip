// BatchNorm2d (train + eval) forward / backward around the conv kernels.  All HBM-bound passes:
// float4 accesses, channel index carried incrementally (no per-element division).
#include <stdlib.h>

#include "common.h"
#include "elem.h"

namespace mvg {

// ---- finalize: merge the conv epilogue's per-wave partials (sum, centred sumsq) -------------
// grid = c/8 blocks, 1024 threads = 8 channels x 128 partial-lanes.  Every group (= view) is merged
// by the same workgroup, one after the other, so that the running statistics are updated in group
// order (the reference runs the backbone on view 0, then view 1: models/rot_mv.py:204-205).
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float *__restrict__ stats,
                                                           const double *__restrict__ sliced, int slices, int groups,
                                                           int partials, int rows_per_partial, long long rows, int c,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float *running_mean,
                                                           float *running_var, float momentum, float eps,
                                                           float *mean_out, float *invstd_out, float *scale,
                                                           float *shift) {
  __shared__ double sh[3][128][8];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int ch = blockIdx.x * 8 + cl;
  const double inv_full = 1.0 / (double)rows_per_partial;
  float rm = 0.f, rv = 0.f;
  if (pl == 0 && ch < c) {
    if (running_mean) rm = running_mean[ch];
    if (running_var) rv = running_var[ch];
  }
  for (int g = 0; g < groups; ++g) {
    const float *st = stats + (long long)g * partials * 2 * c;
    double s = 0.0, q = 0.0, ss = 0.0;
    if (ch < c && sliced) {
      for (int k = pl; k < slices; k += 128) {
        const double *o = sliced + (((long long)g * slices + k) * 3) * c;
        s += o[ch];
        q += o[c + ch];
        ss += o[2 * c + ch];
      }
    } else if (ch < c) {
      // branch-free body (no early exit) so that the loads of several iterations are in flight at once
      const int valid = (int)((rows + rows_per_partial - 1) / rows_per_partial) < partials
                            ? (int)((rows + rows_per_partial - 1) / rows_per_partial) : partials;
#pragma unroll 4
      for (int p = pl; p < valid; p += 128) {
        const float sp_f = st[((long long)p * 2) * c + ch];
        const float qp_f = st[((long long)p * 2 + 1) * c + ch];
        const double sp = sp_f, qp = qp_f;
        s += sp;
        q += qp;
        ss += sp * sp;
      }
      ss *= inv_full;
      // the ragged last partial was scaled by 1/rows_per_partial above: correct it to 1/cnt
      const long long last_cnt = rows - (long long)(valid - 1) * rows_per_partial;
      if (valid > 0 && last_cnt < rows_per_partial && ((valid - 1) & 127) == pl) {
        const double sp = st[((long long)(valid - 1) * 2) * c + ch];
        ss += sp * sp * (1.0 / (double)last_cnt - inv_full);
      }
    }
    // reduce over the 128 partial lanes: lanes 8/16/32 apart inside the wave (shuffles), then the 16
    // waves through LDS - two barriers per group instead of an 8-level LDS tree
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
      ss += __shfl_xor(ss, o, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 8) {
      sh[0][wv][cl] = s;
      sh[1][wv][cl] = q;
      sh[2][wv][cl] = ss;
    }
    __syncthreads();
    if (pl == 0) {
      s = q = ss = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        s += sh[0][k][cl];
        q += sh[1][k][cl];
        ss += sh[2][k][cl];
      }
      sh[0][0][cl] = s;
      sh[1][0][cl] = q;
      sh[2][0][cl] = ss;
    }
    if (pl == 0 && ch < c) {
      const double n = (double)rows;
      const double mean = sh[0][0][cl] / n;
      double m2 = sh[1][0][cl] + (sh[2][0][cl] - sh[0][0][cl] * mean);   // Chan merge of the partials
      if (m2 < 0.0) m2 = 0.0;
      const double var = m2 / n;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float fmean = (float)mean;
      const long long o = (long long)g * c + ch;
      mean_out[o] = fmean;
      invstd_out[o] = invstd;
      const float sc = gamma[ch] * invstd;
      scale[o] = sc;
      shift[o] = beta[ch] - fmean * sc;
      rm = (1.f - momentum) * rm + momentum * fmean;
      const float unbiased = (float)(rows > 1 ? m2 / (n - 1.0) : var);
      rv = (1.f - momentum) * rv + momentum * unbiased;
    }
    __syncthreads();
  }
  if (pl == 0 && ch < c) {
    if (running_mean) running_mean[ch] = rm;
    if (running_var) running_var[ch] = rv;
  }
}

// The same merge with one workgroup per (8 channels, GROUP): grid = (c/8, groups).  bn_finalize_kernel walks the groups
// one after the other - eight dependent rounds of loads and barriers per call at V = 8 (26 us per call, 1.4 ms of a C5
// step in 53 calls) - only because the running statistics are updated in group order; here every group's statistics
// are made at once, (mean, unbiased variance) go to `mv` [groups][2][c] in fp64, and bn_running_update_kernel applies
// them to the running statistics in group order.  Same arithmetic, same order inside a group: identical results.
__global__ __launch_bounds__(1024) void bn_finalize_group_kernel(const float *__restrict__ stats, const double *__restrict__ sliced,
                                                                 int slices, int partials, int rows_per_partial, long long rows,
                                                                 int c, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float eps, float *mean_out,
                                                                 float *invstd_out, float *scale, float *shift,
                                                                 double *__restrict__ mv) {
  __shared__ double sh[3][16][8];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int ch = blockIdx.x * 8 + cl;
  const int g = blockIdx.y, groups = gridDim.y;
  const double inv_full = 1.0 / (double)rows_per_partial;
  const float *st = stats + (long long)g * partials * 2 * c;
  double s = 0.0, q = 0.0, ss = 0.0;
  if (ch < c && sliced) {
    for (int k = pl; k < slices; k += 128) {
      const double *o = sliced + (((long long)g * slices + k) * 3) * c;
      s += o[ch];
      q += o[c + ch];
      ss += o[2 * c + ch];
    }
  } else if (ch < c) {
    const int valid = (int)((rows + rows_per_partial - 1) / rows_per_partial) < partials
                          ? (int)((rows + rows_per_partial - 1) / rows_per_partial) : partials;
#pragma unroll 4
    for (int p = pl; p < valid; p += 128) {
      const float sp_f = st[((long long)p * 2) * c + ch];
      const float qp_f = st[((long long)p * 2 + 1) * c + ch];
      const double sp = sp_f, qp = qp_f;
      s += sp;
      q += qp;
      ss += sp * sp;
    }
    ss *= inv_full;
    const long long last_cnt = rows - (long long)(valid - 1) * rows_per_partial;
    if (valid > 0 && last_cnt < rows_per_partial && ((valid - 1) & 127) == pl) {
      const double sp = st[((long long)(valid - 1) * 2) * c + ch];
      ss += sp * sp * (1.0 / (double)last_cnt - inv_full);
    }
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 8) {
    sh[0][wv][cl] = s;
    sh[1][wv][cl] = q;
    sh[2][wv][cl] = ss;
  }
  __syncthreads();
  if (pl == 0 && ch < c) {
    s = q = ss = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      s += sh[0][k][cl];
      q += sh[1][k][cl];
      ss += sh[2][k][cl];
    }
    const double n = (double)rows;
    const double mean = s / n;
    double m2 = q + (ss - s * mean);                   // Chan merge of the partials
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / n;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    const long long o = (long long)g * c + ch;
    mean_out[o] = fmean;
    invstd_out[o] = invstd;
    const float sc = gamma[ch] * invstd;
    scale[o] = sc;
    shift[o] = beta[ch] - fmean * sc;
    mv[((long long)g * 2) * c + ch] = (double)fmean;
    mv[((long long)g * 2 + 1) * c + ch] = (double)(float)(rows > 1 ? m2 / (n - 1.0) : var);
  }
  (void)groups;
}

__global__ __launch_bounds__(256) void bn_running_update_kernel(const double *__restrict__ mv, int groups, int c, float momentum,
                                                                float *running_mean, float *running_var) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  float rm = running_mean ? running_mean[ch] : 0.f, rv = running_var ? running_var[ch] : 0.f;
  for (int g = 0; g < groups; ++g) {
    rm = (1.f - momentum) * rm + momentum * (float)mv[((long long)g * 2) * c + ch];
    rv = (1.f - momentum) * rv + momentum * (float)mv[((long long)g * 2 + 1) * c + ch];
  }
  if (running_mean) running_mean[ch] = rm;
  if (running_var) running_var[ch] = rv;
}

// Large partial counts (stem at C2: 25 088 per group) first collapse to <= 64 slices per group with
// all CUs busy; bn_finalize_kernel then merges the slices.  slice record: (sum, q, sum of s^2/cnt) in
// fp64, [group][slice][3][c].
__global__ __launch_bounds__(256) void bn_partials_slice_kernel(const float *__restrict__ stats, int partials,
                                                                int rows_per_partial, long long rows, int c,
                                                                int per_slice, double *__restrict__ out, int slices) {
  __shared__ double sh[3][32][8];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;                // 8 channels x 32 partial lanes
  const int ch = blockIdx.x * 8 + cl;
  const int g = blockIdx.z;
  const float *st = stats + (long long)g * partials * 2 * c;
  const int valid = (int)((rows + rows_per_partial - 1) / rows_per_partial) < partials
                        ? (int)((rows + rows_per_partial - 1) / rows_per_partial) : partials;
  const int p0 = blockIdx.y * per_slice;
  const int p1 = p0 + per_slice < valid ? p0 + per_slice : valid;
  double s = 0.0, q = 0.0, ss = 0.0;
  if (ch < c) {
    const double inv_full = 1.0 / (double)rows_per_partial;
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += 32) {
      const double sp = st[((long long)p * 2) * c + ch], qp = st[((long long)p * 2 + 1) * c + ch];
      long long cnt = rows - (long long)p * rows_per_partial;
      s += sp;
      q += qp;
      ss += sp * sp * (cnt >= rows_per_partial ? inv_full : 1.0 / (double)cnt);
    }
  }
  sh[0][pl][cl] = s;
  sh[1][pl][cl] = q;
  sh[2][pl][cl] = ss;
  __syncthreads();
  if (pl == 0 && ch < c) {
    for (int k = 1; k < 32; ++k) {
      s += sh[0][k][cl];
      q += sh[1][k][cl];
      ss += sh[2][k][cl];
    }
    double *o = out + (((long long)g * slices + blockIdx.y) * 3) * c;
    o[ch] = s;
    o[c + ch] = q;
    o[2 * c + ch] = ss;
  }
}

// ---- every group's statistics AND the running statistics in ONE workgroup per 8 channels -------------------------
// grid = c/8, 1024 threads = 8 channels x 128 lanes, the lanes divided between the groups (views): every (channel,
// group) merges its partials (or slices) with 128 / groups lanes, bn_finalize_group_kernel's arithmetic; the per-group
// (mean, unbiased variance) meet in LDS and one thread per channel applies the running-statistics updates in group
// order.  Replaces bn_finalize_group_kernel + bn_running_update_kernel (two launches per conv + BN unit, 106 per
// ResNet-50 step) without any cross-workgroup hand-off.  (Folding them into last-arriving workgroups of ONE grid was
// measured in round 4 and lost: an agent-scope release per workgroup - an L2 write-back - on grids of thousands of
// workgroups cost 37 us per call, 2.96 ms per C3 step against 1.02.)
__global__ __launch_bounds__(1024) void bn_finalize_allgroups_kernel(const float *__restrict__ stats, const double *__restrict__ sliced,
                                                                     int slices, int groups, int lpg, int partials, int rows_per_partial,
                                                                     long long rows, int c, const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta, float eps, float momentum,
                                                                     float *mean_out, float *invstd_out, float *scale, float *shift,
                                                                     float *running_mean, float *running_var) {
  __shared__ double sh[3][128][8];
  __shared__ float sh_mv[2][8][8];                 // [mean | unbiased variance][group][channel]
  const int cl = threadIdx.x & 7, l = threadIdx.x >> 3;
  const int g = l / lpg, pl = l - g * lpg;
  const int ch = blockIdx.x * 8 + cl;
  double s = 0.0, q = 0.0, ss = 0.0;
  if (ch < c && g < groups) {
    if (sliced) {
      for (int k = pl; k < slices; k += lpg) {
        const double *o = sliced + (((long long)g * slices + k) * 3) * c;
        s += o[ch];
        q += o[c + ch];
        ss += o[2 * c + ch];
      }
    } else {
      const float *st = stats + (long long)g * partials * 2 * c;
      const double inv_full = 1.0 / (double)rows_per_partial;
      const int valid = (int)((rows + rows_per_partial - 1) / rows_per_partial) < partials
                            ? (int)((rows + rows_per_partial - 1) / rows_per_partial) : partials;
#pragma unroll 4
      for (int p = pl; p < valid; p += lpg) {
        const double sp = st[((long long)p * 2) * c + ch], qp = st[((long long)p * 2 + 1) * c + ch];
        const long long cnt = rows - (long long)p * rows_per_partial;
        s += sp;
        q += qp;
        ss += sp * sp * (cnt >= rows_per_partial ? inv_full : 1.0 / (double)cnt);
      }
    }
  }
  sh[0][l][cl] = s;
  sh[1][l][cl] = q;
  sh[2][l][cl] = ss;
  __syncthreads();
  if (pl == 0 && g < groups && ch < c) {
    for (int k = 1; k < lpg; ++k) {                  // fixed order
      s += sh[0][l + k][cl];
      q += sh[1][l + k][cl];
      ss += sh[2][l + k][cl];
    }
    const double n = (double)rows;
    const double mean = s / n;
    double m2 = q + (ss - s * mean);                   // Chan merge of the partials
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / n;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    const long long o = (long long)g * c + ch;
    mean_out[o] = fmean;
    invstd_out[o] = invstd;
    const float sc = gamma[ch] * invstd;
    scale[o] = sc;
    shift[o] = beta[ch] - fmean * sc;
    sh_mv[0][g][cl] = fmean;
    sh_mv[1][g][cl] = (float)(rows > 1 ? m2 / (n - 1.0) : var);
  }
  __syncthreads();
  if (l == 0 && ch < c && (running_mean || running_var)) {
    float rm = running_mean ? running_mean[ch] : 0.f, rv = running_var ? running_var[ch] : 0.f;
    for (int gg = 0; gg < groups; ++gg) {
      rm = (1.f - momentum) * rm + momentum * sh_mv[0][gg][cl];
      rv = (1.f - momentum) * rv + momentum * sh_mv[1][gg][cl];
    }
    if (running_mean) running_mean[ch] = rm;
    if (running_var) running_var[ch] = rv;
  }
}

__global__ void bn_eval_affine_kernel(int groups, int c, const float *gamma, const float *beta, const float *rm,
                                      const float *rv, float eps, float *scale, float *shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= groups * c) return;
  const int ch = i % c;
  const float sc = gamma[ch] / sqrtf(rv[ch] + eps);
  scale[i] = sc;
  shift[i] = beta[ch] - rm[ch] * sc;
}

// ---- apply: out = [relu](y*scale + shift [+ residual]) -------------------------------------
// TO / TR: storage of the output and of the residual when they differ from y's (the split path: y fp32 from the conv,
// out in sp for the next conv, residual sp (identity) or fp32 (raw downsample output)); same access width as T.
template <typename T, typename TO = T, typename TR = T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T *__restrict__ y, const float *__restrict__ scale,
                                                       const float *__restrict__ shift,
                                                       const TR *__restrict__ residual,
                                                       const float *__restrict__ res_scale,
                                                       const float *__restrict__ res_shift, int relu,
                                                       TO *__restrict__ out, long long nw_per_group, int cwn, int c,
                                                       unsigned char *__restrict__ relu_bits) {
  typedef Elem<T> E;
  constexpr int W = E::W;                 // float4 groups per 16-byte access (fp32: 1, bf16: 2)
  static_assert(Elem<TO>::W == W && Elem<TR>::W == W, "bn_apply: mixed storage types need the same access width");
  // relu_bits (optional): one byte per 16-byte access, bit k = (element k of the access came out > 0) - the
  // ReLU mask of a residual unit for its backward reduce pass, at 1/16 of the bytes of reading `out` again
  // res_scale / res_shift: the residual is the RAW output of the block's downsample conv and its BatchNorm
  // is applied here (the normalised downsample map is never written: resnet.py:88-93,137-145)
  const float4 *rs4 = res_scale ? reinterpret_cast<const float4 *>(res_scale + (long long)blockIdx.y * c) : nullptr;
  const float4 *rh4 = res_shift ? reinterpret_cast<const float4 *>(res_shift + (long long)blockIdx.y * c) : nullptr;
  const int g = blockIdx.y;
  const float4 *sc4 = reinterpret_cast<const float4 *>(scale + (long long)g * c);
  const float4 *sh4 = reinterpret_cast<const float4 *>(shift + (long long)g * c);
  const long long base = (long long)g * nw_per_group;
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  int cq = (int)(i % cwn);
  const int step = (int)(stride % cwn);
  // a lane keeps its channel group whenever the grid stride is a multiple of the groups per row (every ResNet
  // shape): with bf16 storage (W = 2) the per-channel factors are then loaded once, outside the row loop
  // (C5: bn_bwd_apply 8.0 -> 7.5 ms)
  float4 pa[W], pb[W], pra[W], prb[W];
  auto load_factors = [&]() {
#pragma unroll
    for (int w = 0; w < W; ++w) {
      pa[w] = sc4[cq * W + w];
      pb[w] = sh4[cq * W + w];
      if (rs4) {
        pra[w] = rs4[cq * W + w];
        prb[w] = rh4[cq * W + w];
      }
    }
  };
  load_factors();
  for (; i < nw_per_group; i += stride) {
    if (W == 1 || step != 0) load_factors();      // fp32 (W = 1) measured 3-6 % faster reloading: fewer live registers
    float4 v[W], r[W], o[W];
    E::ldw(y, base + i, v);
    if (residual) Elem<TR>::ldw(residual, base + i, r);
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const float4 a = pa[w], b = pb[w];
      // explicit fma: the backward kernels rebuild the ReLU mask from y with the same expression
      o[w] = make_float4(__builtin_fmaf(v[w].x, a.x, b.x), __builtin_fmaf(v[w].y, a.y, b.y), __builtin_fmaf(v[w].z, a.z, b.z),
                         __builtin_fmaf(v[w].w, a.w, b.w));
      if (residual) {
        float4 rr = r[w];
        if (rs4) {
          const float4 ra = pra[w], rb = prb[w];
          rr = make_float4(__builtin_fmaf(rr.x, ra.x, rb.x), __builtin_fmaf(rr.y, ra.y, rb.y), __builtin_fmaf(rr.z, ra.z, rb.z),
                           __builtin_fmaf(rr.w, ra.w, rb.w));
        }
        o[w].x += rr.x;
        o[w].y += rr.y;
        o[w].z += rr.z;
        o[w].w += rr.w;
      }
      if (relu) {
        o[w].x = fmaxf(o[w].x, 0.f);
        o[w].y = fmaxf(o[w].y, 0.f);
        o[w].z = fmaxf(o[w].z, 0.f);
        o[w].w = fmaxf(o[w].w, 0.f);
      }
    }
    Elem<TO>::stw(out, base + i, o);
    if (relu_bits) {
      unsigned m = 0;
#pragma unroll
      for (int w = 0; w < W; ++w)
        m |= ((o[w].x > 0.f ? 1u : 0u) | (o[w].y > 0.f ? 2u : 0u) | (o[w].z > 0.f ? 4u : 0u) | (o[w].w > 0.f ? 8u : 0u)) << (4 * w);
      relu_bits[base + i] = (unsigned char)m;
    }
    cq += step;
    if (cq >= cwn) cq -= cwn;
  }
}

// ---- backward reduce ------------------------------------------------------------------------
// grid = (chunks, column blocks, groups); thread = one float4 column group x one row lane.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T *g, const T *__restrict__ act,
                                                            const T *__restrict__ y,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const float *__restrict__ mscale,
                                                            const float *__restrict__ mshift, long long rows,
                                                            long long rows_per_chunk, int c, int cwn, int cw,
                                                            float *__restrict__ partial, int chunks, T *dz_out,
                                                            const unsigned char *__restrict__ relu_bits,
                                                            int prows) {
  // prows: rows per partial - 2 (s1, s2) or 3 (+ max |masked gradient| per channel: mvg_bn_bwd_apply_split's bound)
  typedef Elem<T> E;
  constexpr int W = E::W;                 // float4 groups per 16-byte access; a "column" below is one such access
  __shared__ float4 sh[3][256][W];
  const int grp = blockIdx.z;
  const int rl = threadIdx.x / cw, cl = threadIdx.x % cw;
  const int nrl = 256 / cw;
  const int cq = blockIdx.y * cw + cl;
  const bool cok = cq < cwn;
  float4 s1[W], s2[W], mx[W];
#pragma unroll
  for (int w = 0; w < W; ++w) s1[w] = s2[w] = mx[w] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cok) {
    float4 mu[W], is[W], ma[W], mb[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      mu[w] = reinterpret_cast<const float4 *>(mean + (long long)grp * c)[cq * W + w];
      is[w] = reinterpret_cast<const float4 *>(invstd + (long long)grp * c)[cq * W + w];
      ma[w] = mb[w] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mscale) {
        ma[w] = reinterpret_cast<const float4 *>(mscale + (long long)grp * c)[cq * W + w];
        mb[w] = reinterpret_cast<const float4 *>(mshift + (long long)grp * c)[cq * W + w];
      }
    }
    const long long r0 = (long long)blockIdx.x * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    const long long gbase = (long long)grp * rows * cwn;
    // several rows per iteration with independent loads (the trip count is a runtime value, so the
    // compiler keeps a single row in flight otherwise: latency-bound at 2.9 TB/s)
    auto row = [&](long long r, float4 (&a1)[W], float4 (&a2)[W]) {
      const long long off = gbase + r * cwn + cq;
      float4 d[W], v[W], a[W];
      E::ldw(g, off, d);
      E::ldw(y, off, v);
      if (act) E::ldw(act, off, a);
      const unsigned mb8 = relu_bits ? (unsigned)relu_bits[off] : 0u;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        if (relu_bits) {               // the mask bn_apply recorded (bit k of the byte = element k of this access > 0)
          const unsigned m4 = mb8 >> (4 * w);
          d[w].x = (m4 & 1u) ? d[w].x : 0.f;
          d[w].y = (m4 & 2u) ? d[w].y : 0.f;
          d[w].z = (m4 & 4u) ? d[w].z : 0.f;
          d[w].w = (m4 & 8u) ? d[w].w : 0.f;
        } else if (act) {
          d[w].x = a[w].x > 0.f ? d[w].x : 0.f;
          d[w].y = a[w].y > 0.f ? d[w].y : 0.f;
          d[w].z = a[w].z > 0.f ? d[w].z : 0.f;
          d[w].w = a[w].w > 0.f ? d[w].w : 0.f;
        } else if (mscale) {       // ReLU without residual: out > 0 <=> fma(y, scale, shift) > 0 (bn_apply_kernel)
          d[w].x = __builtin_fmaf(v[w].x, ma[w].x, mb[w].x) > 0.f ? d[w].x : 0.f;
          d[w].y = __builtin_fmaf(v[w].y, ma[w].y, mb[w].y) > 0.f ? d[w].y : 0.f;
          d[w].z = __builtin_fmaf(v[w].z, ma[w].z, mb[w].z) > 0.f ? d[w].z : 0.f;
          d[w].w = __builtin_fmaf(v[w].w, ma[w].w, mb[w].w) > 0.f ? d[w].w : 0.f;
        }
        mx[w].x = fmaxf(mx[w].x, fabsf(d[w].x));
        mx[w].y = fmaxf(mx[w].y, fabsf(d[w].y));
        mx[w].z = fmaxf(mx[w].z, fabsf(d[w].z));
        mx[w].w = fmaxf(mx[w].w, fabsf(d[w].w));
        a1[w].x += d[w].x;
        a1[w].y += d[w].y;
        a1[w].z += d[w].z;
        a1[w].w += d[w].w;
        a2[w].x += d[w].x * ((v[w].x - mu[w].x) * is[w].x);
        a2[w].y += d[w].y * ((v[w].y - mu[w].y) * is[w].y);
        a2[w].z += d[w].z * ((v[w].z - mu[w].z) * is[w].z);
        a2[w].w += d[w].w * ((v[w].w - mu[w].w) * is[w].w);
      }
      if (dz_out) E::stw(dz_out, off, d);      // the masked gradient (may alias g): the apply pass then needs no mask
    };
    constexpr int U = W == 1 ? 4 : 2;          // rows in flight per thread (64 bytes of each tensor)
    float4 t1[U - 1][W], t2[U - 1][W];
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
#pragma unroll
      for (int w = 0; w < W; ++w) t1[u][w] = t2[u][w] = make_float4(0.f, 0.f, 0.f, 0.f);
    long long r = r0 + rl;
    for (; r + (U - 1) * (long long)nrl < r1; r += U * (long long)nrl) {
      row(r, s1, s2);
#pragma unroll
      for (int u = 1; u < U; ++u) row(r + u * (long long)nrl, t1[u - 1], t2[u - 1]);
    }
    for (; r < r1; r += nrl) row(r, s1, s2);
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if constexpr (U == 4) {
        s1[w].x += t1[0][w].x + (t1[1][w].x + t1[2][w].x); s1[w].y += t1[0][w].y + (t1[1][w].y + t1[2][w].y);
        s1[w].z += t1[0][w].z + (t1[1][w].z + t1[2][w].z); s1[w].w += t1[0][w].w + (t1[1][w].w + t1[2][w].w);
        s2[w].x += t2[0][w].x + (t2[1][w].x + t2[2][w].x); s2[w].y += t2[0][w].y + (t2[1][w].y + t2[2][w].y);
        s2[w].z += t2[0][w].z + (t2[1][w].z + t2[2][w].z); s2[w].w += t2[0][w].w + (t2[1][w].w + t2[2][w].w);
      } else {
        s1[w].x += t1[0][w].x; s1[w].y += t1[0][w].y; s1[w].z += t1[0][w].z; s1[w].w += t1[0][w].w;
        s2[w].x += t2[0][w].x; s2[w].y += t2[0][w].y; s2[w].z += t2[0][w].z; s2[w].w += t2[0][w].w;
      }
    }
  }
#pragma unroll
  for (int w = 0; w < W; ++w) {
    sh[0][threadIdx.x][w] = s1[w];
    sh[1][threadIdx.x][w] = s2[w];
    sh[2][threadIdx.x][w] = mx[w];
  }
  __syncthreads();
  if (rl == 0 && cok) {
    for (int k = 1; k < nrl; ++k) {
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const float4 a = sh[0][k * cw + cl][w], b = sh[1][k * cw + cl][w], m = sh[2][k * cw + cl][w];
        s1[w].x += a.x; s1[w].y += a.y; s1[w].z += a.z; s1[w].w += a.w;
        s2[w].x += b.x; s2[w].y += b.y; s2[w].z += b.z; s2[w].w += b.w;
        mx[w].x = fmaxf(mx[w].x, m.x); mx[w].y = fmaxf(mx[w].y, m.y); mx[w].z = fmaxf(mx[w].z, m.z); mx[w].w = fmaxf(mx[w].w, m.w);
      }
    }
    float4 *p = reinterpret_cast<float4 *>(partial + (((long long)grp * chunks + blockIdx.x) * prows) * c);
#pragma unroll
    for (int w = 0; w < W; ++w) {
      p[cq * W + w] = s1[w];
      p[c / 4 + cq * W + w] = s2[w];
      if (prows == 3) p[2 * (c / 4) + cq * W + w] = mx[w];
    }
  }
}

// grid = (c/16, groups) blocks, 1024 threads = 16 channels x 64 chunk-lanes: one (view) group per workgroup (a 64-channel
// layer at C3 has 4 x 3136 x 3 x 64 partials: with the groups looped inside ONE workgroup per 16 channels this kernel
// took 30-136 us per call, 1.8 ms per step); bn_bwd_dgamma_kernel then adds the groups in order (reproducible).
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int groups, int chunks,
                                                               int c, float *s1, float *s2, float *mx,
                                                               const float *__restrict__ raw_mean, const float *__restrict__ raw_invstd) {
  // mx != null: the partials have three rows (s1, s2, max |dz|) and mx [groups][c] receives the maxima
  // raw_mean != null: the second row holds the UNCENTRED sum(dz * y) (the bf16 backward-data epilogue keeps no
  // per-channel constants in registers): s2 = invstd * (sum(dz * y) - mean * s1), evaluated on the fp64 totals
  __shared__ double sh[3][64][16];
  const int prows = mx ? 3 : 2;
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  const int g = blockIdx.y;
  double a = 0.0, b = 0.0, m = 0.0;
  if (ch < c)
    for (int k = pl; k < chunks; k += 64) {
      a += partial[(((long long)g * chunks + k) * prows) * c + ch];
      b += partial[(((long long)g * chunks + k) * prows + 1) * c + ch];
      if (mx) m = fmax(m, (double)partial[(((long long)g * chunks + k) * prows + 2) * c + ch]);
    }
  sh[0][pl][cl] = a;
  sh[1][pl][cl] = b;
  sh[2][pl][cl] = m;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (pl < o) {
      sh[0][pl][cl] += sh[0][pl + o][cl];
      sh[1][pl][cl] += sh[1][pl + o][cl];
      sh[2][pl][cl] = fmax(sh[2][pl][cl], sh[2][pl + o][cl]);
    }
    __syncthreads();
  }
  if (pl == 0 && ch < c) {
    s1[(long long)g * c + ch] = (float)sh[0][0][cl];
    double t2 = sh[1][0][cl];
    if (raw_mean) t2 = (double)raw_invstd[(long long)g * c + ch] * (t2 - (double)raw_mean[(long long)g * c + ch] * sh[0][0][cl]);
    s2[(long long)g * c + ch] = (float)t2;
    if (mx) mx[(long long)g * c + ch] = (float)sh[2][0][cl];
  }
}

// dgamma (+)= sum over the groups of s2, dbeta (+)= ... of s1, in group order
__global__ __launch_bounds__(256) void bn_bwd_dgamma_kernel(const float *__restrict__ s1, const float *__restrict__ s2, int groups, int c,
                                                            float *dgamma, float *dbeta, int accumulate) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  double tg = 0.0, tb = 0.0;
  for (int g = 0; g < groups; ++g) {
    tb += (double)s1[(long long)g * c + ch];
    tg += (double)s2[(long long)g * c + ch];
  }
  if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)tg;
  if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)tb;
}

// bn_bwd_dgamma_kernel and bn_dy_scale_kernel (below) in ONE workgroup: dgamma / dbeta over the groups in group order,
// and the bound that scales the unit's dy (split path) - two 5 us launches per unit became one (53 fewer per step).
__global__ __launch_bounds__(1024) void bn_bwd_dgamma_dyscale_kernel(const float *__restrict__ s1, const float *__restrict__ s2,
                                                                     const float *__restrict__ mx, int groups, int c, float *dgamma,
                                                                     float *dbeta, int accumulate, const float *__restrict__ gamma,
                                                                     const float *__restrict__ invstd, float inv_rows, float sqrt_rows,
                                                                     float *__restrict__ dy_sinv) {
  __shared__ float sh_b[16];
  float bd = 0.f;
  for (int ch = threadIdx.x; ch < c; ch += 1024) {
    double tg = 0.0, tb = 0.0;
    for (int g = 0; g < groups; ++g) {
      const float v1 = s1[(long long)g * c + ch], v2 = s2[(long long)g * c + ch];
      tb += (double)v1;
      tg += (double)v2;
      bd = fmaxf(bd, fabsf(gamma[ch] * invstd[(long long)g * c + ch]) * (mx[(long long)g * c + ch] + fabsf(v1) * inv_rows + sqrt_rows * fabsf(v2) * inv_rows));
    }
    if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)tg;
    if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)tb;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bd = fmaxf(bd, __shfl_xor(bd, o, 64));
  if ((threadIdx.x & 63) == 0) sh_b[threadIdx.x >> 6] = bd;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) bd = fmaxf(bd, sh_b[k]);
    *dy_sinv = 1.f / sp_scale_for(bd);
  }
}

template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T *__restrict__ g, const T *__restrict__ act,
                                                           const T *__restrict__ y,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ s1, const float *__restrict__ s2,
                                                           const float *__restrict__ mscale,
                                                           const float *__restrict__ mshift,
                                                           long long nw_per_group, float inv_rows, int cwn, int c,
                                                           TO *__restrict__ dy, T *__restrict__ dz_out) {
  typedef Elem<T> E;
  constexpr int W = E::W;
  static_assert(Elem<TO>::W == W, "bn_bwd_apply: mixed storage types need the same access width");
  const int grp = blockIdx.y;
  const float4 *mu4 = reinterpret_cast<const float4 *>(mean + (long long)grp * c);
  const float4 *is4 = reinterpret_cast<const float4 *>(invstd + (long long)grp * c);
  const float4 *ga4 = reinterpret_cast<const float4 *>(gamma);
  const float4 *a4 = reinterpret_cast<const float4 *>(s1 + (long long)grp * c);
  const float4 *b4 = reinterpret_cast<const float4 *>(s2 + (long long)grp * c);
  const long long base = (long long)grp * nw_per_group;
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  int cq = (int)(i % cwn);
  const int step = (int)(stride % cwn);
  float4 pmu[W], pis[W], pga[W], psa[W], psb[W], pma[W], pmb[W];
  auto load_factors = [&]() {
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int ci = cq * W + w;
      pmu[w] = mu4[ci];
      pis[w] = is4[ci];
      pga[w] = ga4[ci];
      psa[w] = a4[ci];
      psb[w] = b4[ci];
      if (mscale) {
        pma[w] = reinterpret_cast<const float4 *>(mscale + (long long)grp * c)[ci];
        pmb[w] = reinterpret_cast<const float4 *>(mshift + (long long)grp * c)[ci];
      }
    }
  };
  load_factors();
  for (; i < nw_per_group; i += stride) {
    if (W == 1 || step != 0) load_factors();
    float4 d[W], v[W], a[W], o[W];
    E::ldw(g, base + i, d);
    E::ldw(y, base + i, v);
    if (act) E::ldw(act, base + i, a);
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if (act) {
        d[w].x = a[w].x > 0.f ? d[w].x : 0.f;
        d[w].y = a[w].y > 0.f ? d[w].y : 0.f;
        d[w].z = a[w].z > 0.f ? d[w].z : 0.f;
        d[w].w = a[w].w > 0.f ? d[w].w : 0.f;
      } else if (mscale) {
        const float4 ma = pma[w], mb = pmb[w];
        d[w].x = __builtin_fmaf(v[w].x, ma.x, mb.x) > 0.f ? d[w].x : 0.f;
        d[w].y = __builtin_fmaf(v[w].y, ma.y, mb.y) > 0.f ? d[w].y : 0.f;
        d[w].z = __builtin_fmaf(v[w].z, ma.z, mb.z) > 0.f ? d[w].z : 0.f;
        d[w].w = __builtin_fmaf(v[w].w, ma.w, mb.w) > 0.f ? d[w].w : 0.f;
      }
      const float4 mu = pmu[w], is = pis[w], ga = pga[w], sa = psa[w], sb = psb[w];
      o[w].x = ga.x * is.x * (d[w].x - sa.x * inv_rows - (v[w].x - mu.x) * is.x * (sb.x * inv_rows));
      o[w].y = ga.y * is.y * (d[w].y - sa.y * inv_rows - (v[w].y - mu.y) * is.y * (sb.y * inv_rows));
      o[w].z = ga.z * is.z * (d[w].z - sa.z * inv_rows - (v[w].z - mu.z) * is.z * (sb.z * inv_rows));
      o[w].w = ga.w * is.w * (d[w].w - sa.w * inv_rows - (v[w].w - mu.w) * is.w * (sb.w * inv_rows));
    }
    if (dz_out) E::stw(dz_out, base + i, d);
    Elem<TO>::stw(dy, base + i, o);
    cq += step;
    if (cq >= cwn) cq -= cwn;
  }
}

// ---- split path: the passes that write sp (elem.h), one 8-channel chunk per lane -------------------------------
// (fp32 tensors as two float4 per lane, the sp tensor as the chunk's two 16-byte pieces; the generic kernels
// with Elem<sp_t> move 4 channels per lane as 8-byte accesses and are slower.)  A lane's channel chunk is fixed whenever the grid stride is a multiple of the chunks per row
// (every ResNet shape): the per-channel factors are loaded once, outside the row loop.
struct F8 {
  float v[8];
};
__device__ __forceinline__ F8 ld8(const float *p, long long chunk) {
  const float4 a = reinterpret_cast<const float4 *>(p)[2 * chunk], b = reinterpret_cast<const float4 *>(p)[2 * chunk + 1];
  return F8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ F8 ld8_sp(const sp_t *p, long long chunk) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p) + SP_NP * chunk;
  F8 r;
  merge2_chunk(q[0], q[1], r.v);
  return r;
}
__device__ __forceinline__ void st8_sp(sp_t *p, long long chunk, const F8 &x) {
  uint4 q1, q2;
  split2_chunk(x.v, q1, q2);
  uint4 *q = reinterpret_cast<uint4 *>(p) + SP_NP * chunk;
  q[0] = q1;
  q[1] = q2;
}

template <bool RES_SP>
__global__ __launch_bounds__(256) void bn_apply_sp_kernel(const float *__restrict__ y, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, const void *__restrict__ residual,
                                                          const float *__restrict__ res_scale,
                                                          const float *__restrict__ res_shift, int relu, sp_t *__restrict__ out,
                                                          long long n8_per_group, int c8n, int c,
                                                          unsigned short *__restrict__ relu_bits) {
  const int g = blockIdx.y;
  const long long base = (long long)g * n8_per_group;
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  int cq = (int)(i % c8n);
  const int step = (int)(stride % c8n);
  F8 sc = ld8(scale + (long long)g * c, cq), sh = ld8(shift + (long long)g * c, cq), rs, rh;
  const bool raff = res_scale != nullptr;
  if (raff) {
    rs = ld8(res_scale + (long long)g * c, cq);
    rh = ld8(res_shift + (long long)g * c, cq);
  }
  for (; i < n8_per_group; i += stride) {
    if (step != 0) {                       // (not taken for the ResNet shapes: the stride is a multiple of c8n)
      sc = ld8(scale + (long long)g * c, cq);
      sh = ld8(shift + (long long)g * c, cq);
      if (raff) {
        rs = ld8(res_scale + (long long)g * c, cq);
        rh = ld8(res_shift + (long long)g * c, cq);
      }
    }
    const F8 v = ld8(y, base + i);
    F8 r, o;
    if (residual) r = RES_SP ? ld8_sp(reinterpret_cast<const sp_t *>(residual), base + i) : ld8(reinterpret_cast<const float *>(residual), base + i);
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float x = __builtin_fmaf(v.v[k], sc.v[k], sh.v[k]);      // the same expressions as bn_apply_kernel (mask rebuild in the backward)
      if (residual) x += raff ? __builtin_fmaf(r.v[k], rs.v[k], rh.v[k]) : r.v[k];
      if (relu) x = fmaxf(x, 0.f);
      o.v[k] = x;
      m |= (x > 0.f ? 1u : 0u) << (k + (k >= 4 ? 4 : 0));    // two bytes, low nibbles: one byte per 4 channels (bn_bwd_reduce_bits)
    }
    st8_sp(out, base + i, o);
    if (relu_bits) relu_bits[base + i] = (unsigned short)m;
    cq += step;
    if (cq >= c8n) cq -= c8n;
  }
}

// dy is stored times 2^k, k from a bound on |dy| (elem.h: sp_scale_for), per (group, channel) and then the maximum:
//   |dy| = |gamma invstd| |dz - s1/n - xhat s2/n| <= |gamma invstd| (max |dz| + |s1|/n + sqrt(n) |s2|/n)
// (a z-score of n samples is at most sqrt(n - 1)); max |dz| per channel comes from the reduce pass (mx), so a dead or
// low-variance channel - huge invstd, zero gradient - does not inflate the bound.  One workgroup; *dy_sinv = 2^-k.
__global__ __launch_bounds__(1024) void bn_dy_scale_kernel(const float *__restrict__ gamma, const float *__restrict__ invstd,
                                                           const float *__restrict__ s1, const float *__restrict__ s2,
                                                           const float *__restrict__ mx, int groups, int c, float inv_rows,
                                                           float sqrt_rows, float *__restrict__ dy_sinv) {
  __shared__ float sh_b[16];
  float bd = 0.f;
  for (int i = threadIdx.x; i < groups * c; i += 1024)
    bd = fmaxf(bd, fabsf(gamma[i % c] * invstd[i]) * (mx[i] + fabsf(s1[i]) * inv_rows + sqrt_rows * fabsf(s2[i]) * inv_rows));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bd = fmaxf(bd, __shfl_xor(bd, o, 64));
  if ((threadIdx.x & 63) == 0) sh_b[threadIdx.x >> 6] = bd;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) bd = fmaxf(bd, sh_b[k]);
    *dy_sinv = 1.f / sp_scale_for(bd);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_sp_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                              const float *__restrict__ mean, const float *__restrict__ invstd,
                                                              const float *__restrict__ gamma, const float *__restrict__ s1,
                                                              const float *__restrict__ s2, const float *__restrict__ mscale,
                                                              const float *__restrict__ mshift, long long n8_per_group,
                                                              float inv_rows, int c8n, int c, sp_t *__restrict__ dy,
                                                              const float *__restrict__ dy_sinv) {
  const float dsc = 1.f / *dy_sinv;            // 2^k from bn_dy_scale_kernel (exact: a power of two)
  const int grp = blockIdx.y;
  const long long base = (long long)grp * n8_per_group;
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  int cq = (int)(i % c8n);
  const int step = (int)(stride % c8n);
  const long long gc = (long long)grp * c;
  F8 mu = ld8(mean + gc, cq), is = ld8(invstd + gc, cq), ga = ld8(gamma, cq), sa = ld8(s1 + gc, cq), sb = ld8(s2 + gc, cq), ma, mb;
  if (mscale) {
    ma = ld8(mscale + gc, cq);
    mb = ld8(mshift + gc, cq);
  }
  for (; i < n8_per_group; i += stride) {
    if (step != 0) {
      mu = ld8(mean + gc, cq); is = ld8(invstd + gc, cq); ga = ld8(gamma, cq); sa = ld8(s1 + gc, cq); sb = ld8(s2 + gc, cq);
      if (mscale) {
        ma = ld8(mscale + gc, cq);
        mb = ld8(mshift + gc, cq);
      }
    }
    const F8 d = ld8(g, base + i), v = ld8(y, base + i);
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float dd = d.v[k];
      if (mscale) dd = __builtin_fmaf(v.v[k], ma.v[k], mb.v[k]) > 0.f ? dd : 0.f;
      o.v[k] = (ga.v[k] * is.v[k] * (dd - sa.v[k] * inv_rows - (v.v[k] - mu.v[k]) * is.v[k] * (sb.v[k] * inv_rows))) * dsc;
    }
    st8_sp(dy, base + i, o);
    cq += step;
    if (cq >= c8n) cq -= c8n;
  }
}

// ---- stem tail: BatchNorm + ReLU + MaxPool2d(3,2,1), fused --------------------------------------
// The normalised stem activation (B*V x 112 x 112 x 64 floats, 411 MB at C2) is never written:
// forward reads the conv output once and writes the pooled map + argmax; backward rebuilds the
// gradient of the BN output on the fly (gather over the <= 4 pooling windows a pixel can win) and
// the ReLU mask from y (mask = fma(y, scale, shift) > 0, the same expression as the forward).
__device__ __forceinline__ float4 bn_relu4(float4 v, float4 a, float4 b) {
  return make_float4(fmaxf(__builtin_fmaf(v.x, a.x, b.x), 0.f), fmaxf(__builtin_fmaf(v.y, a.y, b.y), 0.f),
                     fmaxf(__builtin_fmaf(v.z, a.z, b.z), 0.f), fmaxf(__builtin_fmaf(v.w, a.w, b.w), 0.f));
}

// grid = (ceil(wo*c4n / 256), images*ho): one thread = one (image, oy, ox, 4 channels), no per-thread
// divisions by runtime values except ox/cq.  First maximum in (kh, kw) scan order (ATen's rule).
template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T *__restrict__ y, const float *__restrict__ scale,
                                                                  const float *__restrict__ shift, TO *__restrict__ pooled,
                                                                  uchar4 *__restrict__ argmax, int n_per_group, int h, int w,
                                                                  int c4n, int ho, int wo) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= wo * c4n) return;
  const int ox = t / c4n, cq = t - ox * c4n;
  const int n = blockIdx.y / ho, oy = blockIdx.y - n * ho;
  const int grp = n / n_per_group;
  const float4 a = reinterpret_cast<const float4 *>(scale)[(long long)grp * c4n + cq];
  const float4 b = reinterpret_cast<const float4 *>(shift)[(long long)grp * c4n + cq];
  float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  uchar4 idx = make_uchar4(0, 0, 0, 0);
  bool first = true;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int iy = oy * 2 - 1 + kh;
    if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ix = ox * 2 - 1 + kw;
      if ((unsigned)ix >= (unsigned)w) continue;
      const float4 v = bn_relu4(Elem<T>::ld4(y, (((long long)n * h + iy) * w + ix) * c4n + cq), a, b);
      const unsigned char k = (unsigned char)(kh * 3 + kw);
      if (first) {
        best = v;
        idx = make_uchar4(k, k, k, k);
        first = false;
      } else {
        if (v.x > best.x || v.x != v.x) { best.x = v.x; idx.x = k; }
        if (v.y > best.y || v.y != v.y) { best.y = v.y; idx.y = k; }
        if (v.z > best.z || v.z != v.z) { best.z = v.z; idx.z = k; }
        if (v.w > best.w || v.w != v.w) { best.w = v.w; idx.w = k; }
      }
    }
  }
  const long long o = (((long long)n * ho + oy) * wo + ox) * c4n + cq;
  Elem<TO>::st4(pooled, o, best);
  argmax[o] = idx;
}

// grid = (chunks, column blocks, groups) and the partial layout of bn_bwd_reduce_kernel.  The sums run
// over POOLED elements: sum_pixels d = sum_windows g_w * mask(winner pixel of w), so every window
// fetches y only at its argmax pixel (one scalar per channel) - a quarter of the elements and a tenth
// of the instructions of the per-pixel gather.  A chunk is a range of pooled lines (image, oy); the
// 256/cw row lanes stride along ox.
template <typename T>
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const T *__restrict__ gp, const uchar4 *__restrict__ am,
                                                                 const T *__restrict__ y, const float *__restrict__ mean,
                                                                 const float *__restrict__ invstd,
                                                                 const float *__restrict__ scale,
                                                                 const float *__restrict__ shift, int lines_per_chunk,
                                                                 int n_per_group, int h, int w, int ho, int wo, int c, int c4n,
                                                                 int cw, float *__restrict__ partial, int chunks, int prows) {
  // prows == 3 (split path): a third partial row bounds max |gradient of a pixel| per channel - 4 x the largest masked
  // window gradient (a pixel wins at most four of the 3x3 stride-2 windows) - for the sp scale of dy
  __shared__ float4 sh[3][256];
  const int grp = blockIdx.z;
  const int rl = threadIdx.x / cw, cl = threadIdx.x % cw;
  const int nrl = 256 / cw;
  const int cq = blockIdx.y * cw + cl;
  const bool cok = cq < c4n;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, mxv[4] = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    const float4 mu4 = reinterpret_cast<const float4 *>(mean + (long long)grp * c)[cq];
    const float4 is4 = reinterpret_cast<const float4 *>(invstd + (long long)grp * c)[cq];
    const float4 sa4 = reinterpret_cast<const float4 *>(scale + (long long)grp * c)[cq];
    const float4 sb4 = reinterpret_cast<const float4 *>(shift + (long long)grp * c)[cq];
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
    const float sa[4] = {sa4.x, sa4.y, sa4.z, sa4.w}, sb[4] = {sb4.x, sb4.y, sb4.z, sb4.w};
    const int lines = n_per_group * ho;
    const int l0 = blockIdx.x * lines_per_chunk;
    const int l1 = l0 + lines_per_chunk < lines ? l0 + lines_per_chunk : lines;
    for (int l = l0; l < l1; ++l) {
      const int img = l / ho, oy = l - img * ho;
      const long long n = (long long)grp * n_per_group + img;
      const long long prow = ((n * ho + oy) * wo) * c4n + cq;
      const T *yimg = y + n * h * w * c + cq * 4;
      for (int ox = rl; ox < wo; ox += nrl) {
        const float4 g4 = Elem<T>::ld4(gp, prow + (long long)ox * c4n);
        const uchar4 k4 = am[prow + (long long)ox * c4n];
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const int k[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kh = (k[j] * 11) >> 5, kw = k[j] - 3 * kh;          // k / 3, k % 3 for k < 9
          const int iy = 2 * oy - 1 + kh, ix = 2 * ox - 1 + kw;          // the winner is an in-bounds pixel
          const float v = Elem<T>::ld1(yimg, ((long long)iy * w + ix) * c + j);
          const float d = __builtin_fmaf(v, sa[j], sb[j]) > 0.f ? g[j] : 0.f;
          mxv[j] = fmaxf(mxv[j], fabsf(d));
          s1[j] += d;
          s2[j] += d * ((v - mu[j]) * is[j]);
        }
      }
    }
  }
  sh[0][threadIdx.x] = make_float4(s1[0], s1[1], s1[2], s1[3]);
  sh[1][threadIdx.x] = make_float4(s2[0], s2[1], s2[2], s2[3]);
  sh[2][threadIdx.x] = make_float4(mxv[0], mxv[1], mxv[2], mxv[3]);
  __syncthreads();
  if (rl == 0 && cok) {
    float4 t1 = sh[0][threadIdx.x], t2 = sh[1][threadIdx.x], t3 = sh[2][threadIdx.x];
    for (int k = 1; k < nrl; ++k) {
      const float4 a = sh[0][k * cw + cl], b = sh[1][k * cw + cl], m = sh[2][k * cw + cl];
      t1.x += a.x; t1.y += a.y; t1.z += a.z; t1.w += a.w;
      t2.x += b.x; t2.y += b.y; t2.z += b.z; t2.w += b.w;
      t3.x = fmaxf(t3.x, m.x); t3.y = fmaxf(t3.y, m.y); t3.z = fmaxf(t3.z, m.z); t3.w = fmaxf(t3.w, m.w);
    }
    float4 *p = reinterpret_cast<float4 *>(partial + (((long long)grp * chunks + blockIdx.x) * prows) * c);
    p[cq] = t1;
    p[c4n + cq] = t2;
    if (prows == 3) p[2 * c4n + cq] = make_float4(4.f * t3.x, 4.f * t3.y, 4.f * t3.z, 4.f * t3.w);
  }
}

// grid = (ceil(ceil(w/2)*c4n / 256), images*ceil(h/2)): one thread = one 2 x 2 pixel quad (rows 2a, 2a+1, columns 2b, 2b+1)
// x 4 channels of the conv output.  The quad's pixels can only have won the pooling windows (a, a+1) x (b, b+1): their
// four (argmax, gradient) records are loaded once and dealt out (one thread per pixel fetched 9 records per quad and had
// a single 16-byte load of y in flight: 0.86 ms at C5 where the bytes take 0.35).  Window contributions are added in
// the order (a, b), (a, b+1), (a+1, b), (a+1, b+1).
template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const T *__restrict__ gp, const uchar4 *__restrict__ am,
                                                                const T *__restrict__ y, const float *__restrict__ mean,
                                                                const float *__restrict__ invstd,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift, const float *__restrict__ s1,
                                                                const float *__restrict__ s2, int n_per_group, int h, int w,
                                                                int ho, int wo, int c4n, float inv_rows,
                                                                TO *__restrict__ dy, const float *__restrict__ dy_sinv) {
  const int hq = (h + 1) >> 1, wq = (w + 1) >> 1;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= wq * c4n) return;
  const int qb = t / c4n, cq = t - qb * c4n;
  const int n = blockIdx.y / hq, qa = blockIdx.y - n * hq;
  const long long gq = (long long)(n / n_per_group) * c4n + cq;
  const float4 mu = reinterpret_cast<const float4 *>(mean)[gq], is = reinterpret_cast<const float4 *>(invstd)[gq];
  const float4 sa = reinterpret_cast<const float4 *>(scale)[gq], sb = reinterpret_cast<const float4 *>(shift)[gq];
  const float4 a1 = reinterpret_cast<const float4 *>(s1)[gq], a2 = reinterpret_cast<const float4 *>(s2)[gq];
  const float4 ga = reinterpret_cast<const float4 *>(gamma)[cq];
  const float dsc = dy_sinv ? 1.f / *dy_sinv : 1.f;         // sp result: times 2^k (bn_dy_scale_kernel; exact)
  // the four windows
  uchar4 wk[2][2];
  float4 wg[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = qa + i < ho && qb + j < wo;
      const long long o = (((long long)n * ho + qa + i) * wo + qb + j) * c4n + cq;
      wk[i][j] = ok ? am[o] : make_uchar4(255, 255, 255, 255);
      wg[i][j] = ok ? Elem<T>::ld4(gp, o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  float4 v[2][2];
  bool pok[2][2];
#pragma unroll
  for (int pi = 0; pi < 2; ++pi)
#pragma unroll
    for (int pj = 0; pj < 2; ++pj) {
      pok[pi][pj] = 2 * qa + pi < h && 2 * qb + pj < w;
      v[pi][pj] = pok[pi][pj] ? Elem<T>::ld4(y, (((long long)n * h + 2 * qa + pi) * w + 2 * qb + pj) * c4n + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
  for (int pi = 0; pi < 2; ++pi)
#pragma unroll
    for (int pj = 0; pj < 2; ++pj) {
      if (!pok[pi][pj]) continue;
      // pixel (2 qa + pi, 2 qb + pj) inside window (qa + i, qb + j): kh = pi + 1 - 2 i, kw = pj + 1 - 2 j (in 0..2)
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kh = pi + 1 - 2 * i;
        if (kh < 0) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int kw = pj + 1 - 2 * j;
          if (kw < 0) continue;
          const unsigned char me = (unsigned char)(kh * 3 + kw);
          const uchar4 k = wk[i][j];
          const float4 g = wg[i][j];
          if (k.x == me) d.x += g.x;
          if (k.y == me) d.y += g.y;
          if (k.z == me) d.z += g.z;
          if (k.w == me) d.w += g.w;
        }
      }
      const float4 vv = v[pi][pj];
      d.x = __builtin_fmaf(vv.x, sa.x, sb.x) > 0.f ? d.x : 0.f;
      d.y = __builtin_fmaf(vv.y, sa.y, sb.y) > 0.f ? d.y : 0.f;
      d.z = __builtin_fmaf(vv.z, sa.z, sb.z) > 0.f ? d.z : 0.f;
      d.w = __builtin_fmaf(vv.w, sa.w, sb.w) > 0.f ? d.w : 0.f;
      float4 o;
      o.x = ga.x * is.x * (d.x - a1.x * inv_rows - (vv.x - mu.x) * is.x * (a2.x * inv_rows));
      o.y = ga.y * is.y * (d.y - a1.y * inv_rows - (vv.y - mu.y) * is.y * (a2.y * inv_rows));
      o.z = ga.z * is.z * (d.z - a1.z * inv_rows - (vv.z - mu.z) * is.z * (a2.z * inv_rows));
      o.w = ga.w * is.w * (d.w - a1.w * inv_rows - (vv.w - mu.w) * is.w * (a2.w * inv_rows));
      if (dy_sinv) {
        o.x *= dsc; o.y *= dsc; o.z *= dsc; o.w *= dsc;
      }
      Elem<TO>::st4(dy, (((long long)n * h + 2 * qa + pi) * w + 2 * qb + pj) * c4n + cq, o);
    }
}

static int bwd_chunks(int groups, long long rows, int c) {
  // at least 64 rows per chunk
  const int c4n = c / 4;
  const int cw = c4n < 256 ? c4n : 256;
  const int colblocks = ceil_div(c4n, cw);
  // one resident round: the kernel's registers and LDS admit three workgroups per CU; measured at C3's shapes
  // (scripts/bn_bench.py): 256 / 384 / 512 / 768 / 1024 / 2048 / 4096 workgroups -> 13.7 / 11.2 / 9.7 / 9.4 / 10.7 /
  // 10.2 / 10.5 ms per step (1024 = a full round plus a third of one)
  int total = 3 * compute_cus();
  if (total < 64) total = 64;
  long long want = total / ((long long)groups * colblocks);
  if (want < 1) want = 1;
  long long maxc = (rows + 63) / 64;
  if (want > maxc) want = maxc;
  return (int)want;
}

static int grid_for(long long n4) {
  long long b = (n4 + 255) / 256;
  if (b > 4096) b = 4096;      // thinner grids (1024 / 2048) were tried to leave wave slots to the wgrad stream: no gain
  if (b < 1) b = 1;
  return (int)b;
}

int bn_bwd_finalize_launch(const float *partial, int groups, int chunks, int c, float *s1, float *s2, float *dgamma,
                           float *dbeta, int accumulate, hipStream_t st, float *mx, const float *raw_mean, const float *raw_invstd,
                           const float *gamma, const float *invstd, long long rows, float *dy_sinv) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(c, 16), groups), dim3(1024), 0, st, partial, groups, chunks, c, s1, s2, mx,
                     raw_mean, raw_invstd);
  if (check_launch("bn_bwd_finalize")) return 1;
  if (dy_sinv && gamma && invstd && mx && rows > 0) {       // split path: dgamma / dbeta and dy's scale in one launch
    hipLaunchKernelGGL(bn_bwd_dgamma_dyscale_kernel, dim3(1), dim3(1024), 0, st, s1, s2, mx, groups, c, dgamma, dbeta, accumulate, gamma, invstd,
                       1.0f / (float)rows, sqrtf((float)rows), dy_sinv);
    return check_launch("bn_bwd_dgamma_dyscale");
  }
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(bn_bwd_dgamma_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, st, s1, s2, groups, c, dgamma, dbeta, accumulate);
    return check_launch("bn_bwd_dgamma");
  }
  return 0;
}

}  // namespace mvg

using namespace mvg;

extern "C" {

int mvg_bn_finalize(const float *stats, int groups, int partials, int rows_per_partial, int64_t rows_per_group, int c,
                    const float *gamma, const float *beta, float *running_mean, float *running_var, float momentum,
                    float eps, float *mean, float *invstd, float *scale, float *shift, void *stream) {
  MVG_REQUIRE(groups > 0 && partials > 0 && c > 0 && rows_per_group > 0, "bn_finalize: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_BN_FINALIZE, st, 0.0, 4.0 * groups * (double)partials * 2 * c);
  const double *sliced = nullptr;
  int slices = 0;
  // scratch: [groups][2][c] doubles for the group-parallel form, then the slices
  const size_t mv_floats = (size_t)groups * 2 * c * 2;
  double *mv = groups > 1 ? (double *)stream_scratch(st, mv_floats) : nullptr;
  if (partials >= 1024) {
    slices = 64;
    const int per_slice = ceil_div(partials, slices);
    slices = ceil_div(partials, per_slice);
    float *base = stream_scratch(st, mv_floats + (size_t)groups * slices * 3 * c * 2);
    double *buf = base ? (double *)(base + mv_floats) : nullptr;
    if (buf) {
      hipLaunchKernelGGL(bn_partials_slice_kernel, dim3(ceil_div(c, 8), slices, groups), dim3(256), 0, st, stats, partials,
                         rows_per_partial, (long long)rows_per_group, c, per_slice, buf, slices);
      if (check_launch("bn_partials_slice")) return 1;
      sliced = buf;
    }
  }
  if (groups > 1 && groups <= 4 && (partials < 1024 || sliced)) {
    // every group's statistics and the running statistics in ONE launch (the workgroup's lanes divided between the groups:
    // >= 32 lanes per group; with 8 groups - C5's shapes - 16 lanes per group made the call slower than the two launches
    // below: bn_finalize 1.00 -> 1.39 ms per C5 step)
    hipLaunchKernelGGL(bn_finalize_allgroups_kernel, dim3(ceil_div(c, 8)), dim3(1024), 0, st, stats, sliced, slices, groups, 128 / groups,
                       partials, rows_per_partial, (long long)rows_per_group, c, gamma, beta, eps, momentum, mean, invstd, scale, shift,
                       running_mean, running_var);
    return check_launch("bn_finalize(all groups)");
  }
  if (mv) {
    hipLaunchKernelGGL(bn_finalize_group_kernel, dim3(ceil_div(c, 8), groups), dim3(1024), 0, st, stats, sliced, slices, partials,
                       rows_per_partial, (long long)rows_per_group, c, gamma, beta, eps, mean, invstd, scale, shift, mv);
    if (check_launch("bn_finalize(groups)")) return 1;
    if (running_mean || running_var) {
      hipLaunchKernelGGL(bn_running_update_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, st, mv, groups, c, momentum, running_mean,
                         running_var);
      return check_launch("bn_running_update");
    }
    return 0;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(c, 8)), dim3(1024), 0, st, stats, sliced, slices, groups, partials,
                     rows_per_partial, (long long)rows_per_group, c, gamma, beta, running_mean, running_var, momentum, eps, mean,
                     invstd, scale, shift);
  return check_launch("bn_finalize");
}

int mvg_bn_eval_affine(int groups, int c, const float *gamma, const float *beta, const float *running_mean,
                       const float *running_var, float eps, float *scale, float *shift, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_BN_FINALIZE, st, 0.0, 4.0 * 6 * c);
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ceil_div((long long)groups * c, 256)), dim3(256), 0, st, groups, c, gamma,
                     beta, running_mean, running_var, eps, scale, shift);
  return check_launch("bn_eval_affine");
}

}  // extern "C" (templated implementations below)

template <typename T, typename TO = T, typename TR = T>
static int bn_apply_impl(const T *y, const float *scale, const float *shift, const TR *residual, const float *res_scale,
                         const float *res_shift, int relu, TO *out, int groups, int64_t rows_per_group, int c, void *stream,
                         uint8_t *relu_bits = nullptr) {
  MVG_REQUIRE(c % 4 == 0, "bn_apply: c %% 4 != 0");
  MVG_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (residual || !res_scale),
              "bn_apply: res_scale / res_shift go together and need a residual");
  hipStream_t st = (hipStream_t)stream;
  const long long n4 = rows_per_group * (c / 4);
  constexpr int W = Elem<T>::W;
  MVG_REQUIRE(c % (4 * W) == 0, "bn_apply: c must be a multiple of %d", 4 * W);
  ProfScope ps(MVG_K_BN_APPLY, st, 0.0,
               4.0 * groups * (double)n4 * (Elem<T>::kBytes + Elem<TO>::kBytes + (residual ? Elem<TR>::kBytes : 0.0)));
  hipLaunchKernelGGL((bn_apply_kernel<T, TO, TR>), dim3(grid_for(n4 / W), groups), dim3(256), 0, st, y, scale, shift, residual, res_scale,
                     res_shift, relu, out, n4 / W, c / 4 / W, c, relu_bits);
  return check_launch("bn_apply");
}

template <typename T>
static int bn_bwd_reduce_impl(const T *g, const T *act, const T *y, const float *mean, const float *invstd,
                              const float *relu_scale, const float *relu_shift, int groups, int64_t rows_per_group, int c,
                              float *s1, float *s2, float *dgamma, float *dbeta, int accumulate, float *workspace, T *dz_out,
                              void *stream, const uint8_t *relu_bits = nullptr, float *mx = nullptr, const float *gamma = nullptr,
                              float *dy_sinv = nullptr) {
  MVG_REQUIRE(!(act && relu_scale) && !(relu_bits && (act || relu_scale)),
              "bn_bwd_reduce: give the ReLU mask ONE way: act, (relu_scale, relu_shift) or relu_bits");
  MVG_REQUIRE((relu_scale == nullptr) == (relu_shift == nullptr), "bn_bwd_reduce: relu_scale and relu_shift go together");
  MVG_REQUIRE(c % 4 == 0, "bn_bwd_reduce: c %% 4 != 0");
  MVG_REQUIRE(workspace != nullptr, "bn_bwd_reduce: workspace required");
  hipStream_t st = (hipStream_t)stream;
  constexpr int W = Elem<T>::W;
  MVG_REQUIRE(c % (4 * W) == 0, "bn_bwd_reduce: c must be a multiple of %d", 4 * W);
  const int c4n = c / 4 / W;                 // 16-byte column groups per row
  const int cw = c4n < 256 ? c4n : 256;
  MVG_REQUIRE(256 % cw == 0, "bn_bwd_reduce: c/%d must divide 256 or be a multiple of it (c=%d)", 4 * W, c);
  const int chunks = bwd_chunks(groups, rows_per_group, c);
  const long long rpc = (rows_per_group + chunks - 1) / chunks;
  ProfScope ps(MVG_K_BN_BWD_REDUCE, st, 0.0, Elem<T>::kBytes * groups * (double)rows_per_group * c * ((act ? 3 : 2) + (dz_out ? 1 : 0)));
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(chunks, ceil_div(c4n, cw), groups), dim3(256), 0, st, g, act, y, mean, invstd,
                     relu_scale, relu_shift, (long long)rows_per_group, rpc, c, c4n, cw, workspace, chunks, dz_out, relu_bits, mx ? 3 : 2);
  if (check_launch("bn_bwd_reduce")) return 1;
  return bn_bwd_finalize_launch(workspace, groups, chunks, c, s1, s2, dgamma, dbeta, accumulate, st, mx, nullptr, nullptr, gamma, invstd,
                                (long long)rows_per_group, dy_sinv);
}

template <typename T, typename TO = T>
static int bn_bwd_apply_impl(const T *g, const T *act, const T *y, const float *mean, const float *invstd, const float *gamma,
                             const float *s1, const float *s2, const float *relu_scale, const float *relu_shift, int groups,
                             int64_t rows_per_group, int c, TO *dy, T *dz_out, void *stream) {
  MVG_REQUIRE(!(act && relu_scale), "bn_bwd_apply: give the ReLU mask either as act or as (relu_scale, relu_shift)");
  MVG_REQUIRE((relu_scale == nullptr) == (relu_shift == nullptr), "bn_bwd_apply: relu_scale and relu_shift go together");
  MVG_REQUIRE(c % 4 == 0, "bn_bwd_apply: c %% 4 != 0");
  hipStream_t st = (hipStream_t)stream;
  const long long n4 = rows_per_group * (c / 4);
  constexpr int W = Elem<T>::W;
  MVG_REQUIRE(c % (4 * W) == 0, "bn_bwd_apply: c must be a multiple of %d", 4 * W);
  ProfScope ps(MVG_K_BN_BWD_APPLY, st, 0.0,
               4.0 * groups * (double)n4 * (Elem<T>::kBytes * ((act ? 3 : 2) + (dz_out ? 1 : 0)) + Elem<TO>::kBytes));
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T, TO>), dim3(grid_for(n4 / W), groups), dim3(256), 0, st, g, act, y, mean, invstd, gamma, s1, s2,
                     relu_scale, relu_shift, n4 / W, 1.0f / (float)rows_per_group, c / 4 / W, c, dy, dz_out);
  return check_launch("bn_bwd_apply");
}

template <typename T, typename TO = T>
static int bn_relu_maxpool_fwd_impl(const T *y, const float *scale, const float *shift, TO *pooled, uint8_t *argmax, int groups,
                                    int n_per_group, int h, int w, int c, int ho, int wo, void *stream) {
  MVG_REQUIRE(c % 4 == 0, "bn_relu_maxpool: c %% 4 != 0");
  MVG_REQUIRE(ho == (h + 2 - 3) / 2 + 1 && wo == (w + 2 - 3) / 2 + 1, "bn_relu_maxpool: bad output size");
  const long long total = (long long)groups * n_per_group * ho * wo * (c / 4);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_POOL, st, 0.0,
               Elem<T>::kBytes * (double)groups * n_per_group * h * w * c + Elem<TO>::kBytes * (double)total * 4 + (double)total * 4);
  MVG_REQUIRE((long long)groups * n_per_group * ho < 65536, "bn_relu_maxpool: images*ho must fit grid.y");
  hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<T, TO>), dim3(ceil_div((long long)wo * (c / 4), 256), groups * n_per_group * ho),
                     dim3(256), 0, st, y, scale, shift, pooled, (uchar4 *)argmax, n_per_group, h, w, c / 4, ho, wo);
  return check_launch("bn_relu_maxpool_fwd");
}

template <typename T>
static int bn_relu_maxpool_bwd_reduce_impl(const T *g_pooled, const uint8_t *argmax, const T *y, const float *mean,
                                           const float *invstd, const float *scale, const float *shift, int groups,
                                           int n_per_group, int h, int w, int c, int ho, int wo, float *s1, float *s2,
                                           float *dgamma, float *dbeta, int accumulate, float *workspace, void *stream,
                                           float *mx = nullptr, const float *gamma = nullptr, float *dy_sinv = nullptr) {
  MVG_REQUIRE(c % 4 == 0, "bn_relu_maxpool_bwd_reduce: c %% 4 != 0");
  MVG_REQUIRE(workspace != nullptr, "bn_relu_maxpool_bwd_reduce: workspace required");
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)n_per_group * h * w;
  const int c4n = c / 4;
  const int cw = c4n < 256 ? c4n : 256;
  MVG_REQUIRE(256 % cw == 0, "bn_relu_maxpool_bwd_reduce: c/4 must divide 256 or be a multiple of it (c=%d)", c);
  // chunk = whole image lines; never more chunks than mvg_bn_bwd_workspace_floats(groups, rows, c) sizes
  const int lines = n_per_group * ho;                 // pooled lines
  int chunks = bwd_chunks(groups, rows, c);
  if (chunks > lines) chunks = lines;
  const int lpc = (lines + chunks - 1) / chunks;
  chunks = (lines + lpc - 1) / lpc;
  ProfScope ps(MVG_K_BN_BWD_REDUCE, st, 0.0,
               Elem<T>::kBytes * groups * ((double)rows * c + (double)n_per_group * ho * wo * c * 1.25));
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel<T>, dim3(chunks, ceil_div(c4n, cw), groups), dim3(256), 0, st, g_pooled,
                     (const uchar4 *)argmax, y, mean, invstd, scale, shift, lpc, n_per_group, h, w, ho, wo, c, c4n, cw, workspace,
                     chunks, mx ? 3 : 2);
  if (check_launch("bn_relu_maxpool_bwd_reduce")) return 1;
  return bn_bwd_finalize_launch(workspace, groups, chunks, c, s1, s2, dgamma, dbeta, accumulate, st, mx, nullptr, nullptr, gamma, invstd, rows,
                                dy_sinv);
}

template <typename T, typename TO = T>
static int bn_relu_maxpool_bwd_apply_impl(const T *g_pooled, const uint8_t *argmax, const T *y, const float *mean,
                                          const float *invstd, const float *gamma, const float *scale, const float *shift,
                                          const float *s1, const float *s2, int groups, int n_per_group, int h, int w, int c,
                                          int ho, int wo, TO *dy, void *stream, const float *dy_sinv = nullptr) {
  MVG_REQUIRE(c % 4 == 0, "bn_relu_maxpool_bwd_apply: c %% 4 != 0");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_BN_BWD_APPLY, st, 0.0,
               Elem<T>::kBytes * groups * ((double)n_per_group * h * w * c * 2 + (double)n_per_group * ho * wo * c * 1.25));
  MVG_REQUIRE((long long)groups * n_per_group * ((h + 1) / 2) < 65536, "bn_relu_maxpool: images * h / 2 must fit grid.y");
  hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T, TO>), dim3(ceil_div((long long)((w + 1) / 2) * (c / 4), 256), groups * n_per_group * ((h + 1) / 2)), dim3(256),
                     0, st, g_pooled, (const uchar4 *)argmax, y, mean, invstd, gamma, scale, shift, s1, s2, n_per_group, h, w, ho,
                     wo, c / 4, 1.0f / (float)((long long)n_per_group * h * w), dy, dy_sinv);
  return check_launch("bn_relu_maxpool_bwd_apply");
}

extern "C" {

size_t mvg_bn_bwd_workspace_floats(int groups, int64_t rows_per_group, int c) {
  return (size_t)groups * bwd_chunks(groups, rows_per_group, c) * 3 * c;      // (s1, s2, and - mvg_bn_bwd_reduce_split - max |dz|)
}

#define MVG_BN_FACES(SUFFIX, T)                                                                                              \
  int mvg_bn_apply##SUFFIX(const T *y, const float *scale, const float *shift, const T *residual,                           \
                           const float *res_scale, const float *res_shift, int relu, T *out, int groups,                    \
                           int64_t rows_per_group, int c, void *stream) {                                                   \
    return bn_apply_impl<T>(y, scale, shift, residual, res_scale, res_shift, relu, out, groups, rows_per_group, c, stream);  \
  }                                                                                                                          \
  int mvg_bn_bwd_reduce##SUFFIX(const T *g, const T *act, const T *y, const float *mean, const float *invstd,               \
                                const float *relu_scale, const float *relu_shift, int groups, int64_t rows_per_group,       \
                                int c, float *s1, float *s2, float *dgamma, float *dbeta, int accumulate,                   \
                                float *workspace, T *dz_out, void *stream) {                                                \
    return bn_bwd_reduce_impl<T>(g, act, y, mean, invstd, relu_scale, relu_shift, groups, rows_per_group, c, s1, s2,        \
                                 dgamma, dbeta, accumulate, workspace, dz_out, stream);                                     \
  }                                                                                                                          \
  int mvg_bn_bwd_apply##SUFFIX(const T *g, const T *act, const T *y, const float *mean, const float *invstd,                \
                               const float *gamma, const float *s1, const float *s2, const float *relu_scale,               \
                               const float *relu_shift, int groups, int64_t rows_per_group, int c, T *dy, T *dz_out,        \
                               void *stream) {                                                                              \
    return bn_bwd_apply_impl<T>(g, act, y, mean, invstd, gamma, s1, s2, relu_scale, relu_shift, groups, rows_per_group,     \
                                c, dy, dz_out, stream);                                                                     \
  }                                                                                                                          \
  int mvg_bn_relu_maxpool_fwd##SUFFIX(const T *y, const float *scale, const float *shift, T *pooled, uint8_t *argmax,       \
                                      int groups, int n_per_group, int h, int w, int c, int ho, int wo, void *stream) {     \
    return bn_relu_maxpool_fwd_impl<T>(y, scale, shift, pooled, argmax, groups, n_per_group, h, w, c, ho, wo, stream);      \
  }                                                                                                                          \
  int mvg_bn_relu_maxpool_bwd_reduce##SUFFIX(const T *g_pooled, const uint8_t *argmax, const T *y, const float *mean,       \
                                             const float *invstd, const float *scale, const float *shift, int groups,       \
                                             int n_per_group, int h, int w, int c, int ho, int wo, float *s1, float *s2,    \
                                             float *dgamma, float *dbeta, int accumulate, float *workspace,                 \
                                             void *stream) {                                                                \
    return bn_relu_maxpool_bwd_reduce_impl<T>(g_pooled, argmax, y, mean, invstd, scale, shift, groups, n_per_group, h, w,   \
                                              c, ho, wo, s1, s2, dgamma, dbeta, accumulate, workspace, stream);             \
  }                                                                                                                          \
  int mvg_bn_relu_maxpool_bwd_apply##SUFFIX(const T *g_pooled, const uint8_t *argmax, const T *y, const float *mean,        \
                                            const float *invstd, const float *gamma, const float *scale,                    \
                                            const float *shift, const float *s1, const float *s2, int groups,               \
                                            int n_per_group, int h, int w, int c, int ho, int wo, T *dy, void *stream) {    \
    return bn_relu_maxpool_bwd_apply_impl<T>(g_pooled, argmax, y, mean, invstd, gamma, scale, shift, s1, s2, groups,        \
                                             n_per_group, h, w, c, ho, wo, dy, stream);                                     \
  }

MVG_BN_FACES(, float)
MVG_BN_FACES(_bf16, uint16_t)
#undef MVG_BN_FACES

// residual units: bn_apply also records the ReLU mask as one byte per 16-byte access (groups * rows * c / 4 bytes
// in fp32, / 8 in bf16) and the backward reduce pass reads those bytes instead of the activation
#define MVG_BN_BITS_FACES(SUFFIX, T)                                                                                          \
  int mvg_bn_apply_bits##SUFFIX(const T *y, const float *scale, const float *shift, const T *residual,                      \
                                const float *res_scale, const float *res_shift, T *out, uint8_t *relu_bits, int groups,      \
                                int64_t rows_per_group, int c, void *stream) {                                               \
    MVG_REQUIRE(relu_bits != nullptr, "bn_apply_bits: relu_bits is required");                                               \
    return bn_apply_impl<T>(y, scale, shift, residual, res_scale, res_shift, 1, out, groups, rows_per_group, c, stream,      \
                            relu_bits);                                                                                      \
  }                                                                                                                          \
  int mvg_bn_bwd_reduce_bits##SUFFIX(const T *g, const uint8_t *relu_bits, const T *y, const float *mean,                   \
                                     const float *invstd, int groups, int64_t rows_per_group, int c, float *s1, float *s2,   \
                                     float *dgamma, float *dbeta, int accumulate, float *workspace, T *dz_out,               \
                                     void *stream) {                                                                         \
    MVG_REQUIRE(relu_bits != nullptr, "bn_bwd_reduce_bits: relu_bits is required");                                          \
    return bn_bwd_reduce_impl<T>(g, nullptr, y, mean, invstd, nullptr, nullptr, groups, rows_per_group, c, s1, s2, dgamma,   \
                                 dbeta, accumulate, workspace, dz_out, stream, relu_bits);                                   \
  }
// ---- split path (conv_split.hip): conv outputs and gradients fp32, conv INPUTS (activations, dy) in sp ----------
// the reduce pass of a unit whose dy goes out in sp: mvg_bn_bwd_reduce / _bits (mask from relu_bits, or from relu_scale /
// relu_shift, or none) that also leaves max |masked gradient| per (group, channel) in mx [groups][c]
int mvg_bn_bwd_reduce_split(const float *g, const uint8_t *relu_bits, const float *y, const float *mean, const float *invstd,
                            const float *relu_scale, const float *relu_shift, int groups, int64_t rows_per_group, int c, float *s1,
                            float *s2, float *dgamma, float *dbeta, int accumulate, float *workspace, float *dz_out,
                            float *mx, const float *gamma, float *dy_sinv, void *stream) {
  MVG_REQUIRE(mx != nullptr, "bn_bwd_reduce_split: mx is required");
  MVG_REQUIRE((gamma == nullptr) == (dy_sinv == nullptr), "bn_bwd_reduce_split: gamma and dy_sinv go together");
  return bn_bwd_reduce_impl<float>(g, nullptr, y, mean, invstd, relu_scale, relu_shift, groups, rows_per_group, c, s1, s2, dgamma,
                                   dbeta, accumulate, workspace, dz_out, stream, relu_bits, mx, gamma, dy_sinv);
}

int mvg_bn_apply_split(const float *y, const float *scale, const float *shift, const void *residual, int residual_sp,
                       const float *res_scale, const float *res_shift, int relu, void *out_sp, uint8_t *relu_bits, int groups,
                       int64_t rows_per_group, int c, void *stream) {
  MVG_REQUIRE(c % 8 == 0, "bn_apply_split: c %% 8 != 0");
  MVG_REQUIRE(!(residual_sp && res_scale), "bn_apply_split: an sp residual is already normalised (no res_scale / res_shift)");
  MVG_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (residual || !res_scale),
              "bn_apply_split: res_scale / res_shift go together and need a residual");
  hipStream_t st = (hipStream_t)stream;
  const long long n8 = rows_per_group * (c / 8);
  ProfScope ps(MVG_K_BN_APPLY, st, 0.0, 8.0 * groups * (double)n8 * (4.0 + SP_BYTES + (residual ? (residual_sp ? (double)SP_BYTES : 4.0) : 0.0)));
  const dim3 grid(grid_for(n8), groups), block(256);
  if (residual && residual_sp)
    hipLaunchKernelGGL(bn_apply_sp_kernel<true>, grid, block, 0, st, y, scale, shift, residual, res_scale, res_shift, relu,
                       (sp_t *)out_sp, n8, c / 8, c, (unsigned short *)relu_bits);
  else
    hipLaunchKernelGGL(bn_apply_sp_kernel<false>, grid, block, 0, st, y, scale, shift, residual, res_scale, res_shift, relu,
                       (sp_t *)out_sp, n8, c / 8, c, (unsigned short *)relu_bits);
  return check_launch("bn_apply_split");
}

int mvg_bn_bwd_apply_split(const float *g, const float *y, const float *mean, const float *invstd, const float *gamma,
                           const float *s1, const float *s2, const float *relu_scale, const float *relu_shift, int groups,
                           int64_t rows_per_group, int c, void *dy_sp, const float *mx, float *dy_sinv, int dy_sinv_ready, void *stream) {
  MVG_REQUIRE(c % 8 == 0, "bn_bwd_apply_split: c %% 8 != 0");
  MVG_REQUIRE(mx && dy_sinv, "bn_bwd_apply_split: mx (max |masked gradient| per (group, channel) from the reduce pass) and dy_sinv are required");
  MVG_REQUIRE((relu_scale == nullptr) == (relu_shift == nullptr), "bn_bwd_apply_split: relu_scale and relu_shift go together");
  hipStream_t st = (hipStream_t)stream;
  const long long n8 = rows_per_group * (c / 8);
  ProfScope ps(MVG_K_BN_BWD_APPLY, st, 0.0, 8.0 * groups * (double)n8 * (8.0 + SP_BYTES));
  if (!dy_sinv_ready) {          // (the reduce pass / the fused backward-data launch left *dy_sinv when it was given gamma)
    hipLaunchKernelGGL(bn_dy_scale_kernel, dim3(1), dim3(1024), 0, st, gamma, invstd, s1, s2, mx, groups, c, 1.0f / (float)rows_per_group,
                       sqrtf((float)rows_per_group), dy_sinv);
    if (check_launch("bn_dy_scale")) return 1;
  }
  hipLaunchKernelGGL(bn_bwd_apply_sp_kernel, dim3(grid_for(n8), groups), dim3(256), 0, st, g, y, mean, invstd, gamma, s1, s2,
                     relu_scale, relu_shift, n8, 1.0f / (float)rows_per_group, c / 8, c, (sp_t *)dy_sp, dy_sinv);
  return check_launch("bn_bwd_apply_split");
}

int mvg_bn_relu_maxpool_fwd_split(const float *y, const float *scale, const float *shift, void *pooled_sp, uint8_t *argmax,
                                  int groups, int n_per_group, int h, int w, int c, int ho, int wo, void *stream) {
  MVG_REQUIRE(c % 8 == 0, "bn_relu_maxpool_fwd_split: c %% 8 != 0");
  return bn_relu_maxpool_fwd_impl<float, sp_t>(y, scale, shift, (sp_t *)pooled_sp, argmax, groups, n_per_group, h, w, c, ho, wo,
                                               stream);
}

// the stem tail's backward on the split path: the pooled gradient and y are fp32, dy goes to the stem's wgrad in sp;
// mx [groups][c] (from the reduce pass) bounds the gradient of a pixel: 4 x the largest masked window gradient
int mvg_bn_relu_maxpool_bwd_reduce_split(const float *g_pooled, const uint8_t *argmax, const float *y, const float *mean,
                                         const float *invstd, const float *scale, const float *shift, int groups, int n_per_group,
                                         int h, int w, int c, int ho, int wo, float *s1, float *s2, float *dgamma, float *dbeta,
                                         int accumulate, float *workspace, float *mx, const float *gamma, float *dy_sinv, void *stream) {
  MVG_REQUIRE(mx != nullptr, "bn_relu_maxpool_bwd_reduce_split: mx is required");
  MVG_REQUIRE((gamma == nullptr) == (dy_sinv == nullptr), "bn_relu_maxpool_bwd_reduce_split: gamma and dy_sinv go together");
  return bn_relu_maxpool_bwd_reduce_impl<float>(g_pooled, argmax, y, mean, invstd, scale, shift, groups, n_per_group, h, w, c, ho, wo,
                                                s1, s2, dgamma, dbeta, accumulate, workspace, stream, mx, gamma, dy_sinv);
}

int mvg_bn_relu_maxpool_bwd_apply_split(const float *g_pooled, const uint8_t *argmax, const float *y, const float *mean,
                                        const float *invstd, const float *gamma, const float *scale, const float *shift,
                                        const float *s1, const float *s2, int groups, int n_per_group, int h, int w, int c, int ho,
                                        int wo, void *dy_sp, const float *mx, float *dy_sinv, int dy_sinv_ready, void *stream) {
  MVG_REQUIRE(c % 8 == 0 && mx && dy_sinv, "bn_relu_maxpool_bwd_apply_split: c %% 8 != 0, or mx / dy_sinv missing");
  const long long rows = (long long)n_per_group * h * w;
  if (!dy_sinv_ready) {
    hipLaunchKernelGGL(bn_dy_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, gamma, invstd, s1, s2, mx, groups, c,
                       1.0f / (float)rows, sqrtf((float)rows), dy_sinv);
    if (check_launch("bn_dy_scale")) return 1;
  }
  return bn_relu_maxpool_bwd_apply_impl<float, sp_t>(g_pooled, argmax, y, mean, invstd, gamma, scale, shift, s1, s2, groups,
                                                     n_per_group, h, w, c, ho, wo, (sp_t *)dy_sp, stream, dy_sinv);
}

MVG_BN_BITS_FACES(, float)
MVG_BN_BITS_FACES(_bf16, uint16_t)
#undef MVG_BN_BITS_FACES

}  // extern "C"
