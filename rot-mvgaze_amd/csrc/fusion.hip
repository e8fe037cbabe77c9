// Geometry, cross-view fusion operand builders, bias-gradient sums, the 2-wide gaze head and
// the angular loss.  All of these are tiny / HBM-bound next to the GEMMs.
#include <math.h>

#include "common.h"
#include "elem.h"

namespace mvg {

// R = Ry(yaw) @ Rx(-pitch)  (utils/math.py:188-219)
__global__ void rotation_matrix_kernel(const float *__restrict__ py, float *__restrict__ rot, int n, int inverse) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p = -py[2 * i], y = py[2 * i + 1];
  const float cp = cosf(p), sp = sinf(p), cy = cosf(y), sy = sinf(y);
  float m[9] = {cy, sy * sp, sy * cp, 0.f, cp, -sp, -sy, cy * sp, cy * cp};
  float *o = rot + 9 * (long long)i;
  if (inverse) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) o[a * 3 + b] = m[b * 3 + a];
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = m[k];
  }
}

// rel[d][b] = rot[b][vi[d]] @ rot[b][vj[d]]^T
__global__ void relative_rotation_kernel(const float *__restrict__ rot, const int *__restrict__ vi,
                                         const int *__restrict__ vj, float *__restrict__ rel, int batch, int views,
                                         int dirs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dirs * batch) return;
  const int d = i / batch, b = i - d * batch;
  const float *ra = rot + ((long long)b * views + vi[d]) * 9;
  const float *rb = rot + ((long long)b * views + vj[d]) * 9;
  float *o = rel + (long long)i * 9;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[r * 3 + c] = ra[r * 3 + 0] * rb[c * 3 + 0] + ra[r * 3 + 1] * rb[c * 3 + 1] + ra[r * 3 + 2] * rb[c * 3 + 2];
}

// x[(d,b)] = [ img_feat[view_of[d]][b] | rel[d][b] @ feat[src_of[d]][b] ]
__global__ __launch_bounds__(256) void rotcat_fwd_kernel(const float *__restrict__ img_feat,
                                                         const float *__restrict__ feat, const float *__restrict__ rel,
                                                         const int *__restrict__ view_of, const int *__restrict__ src_of,
                                                         float *__restrict__ x, int batch, int cf, int nvec) {
  const int row = blockIdx.x;            // d*batch + b
  const int d = row / batch, b = row - d * batch;
  const int kin = cf + 3 * nvec;
  float *xo = x + (long long)row * kin;
  const float4 *src = reinterpret_cast<const float4 *>(img_feat + ((long long)view_of[d] * batch + b) * cf);
  for (int i = threadIdx.x; i < cf / 4; i += 256) reinterpret_cast<float4 *>(xo)[i] = src[i];
  const float *f = feat + ((long long)src_of[d] * batch + b) * 3 * nvec;
  float r[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (rel) {
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = rel[(long long)row * 9 + k];
  }
  for (int k = threadIdx.x; k < nvec; k += 256) {
    const float f0 = f[k], f1 = f[nvec + k], f2 = f[2 * nvec + k];
    xo[cf + k] = r[0] * f0 + r[1] * f1 + r[2] * f2;
    xo[cf + nvec + k] = r[3] * f0 + r[4] * f1 + r[5] * f2;
    xo[cf + 2 * nvec + k] = r[6] * f0 + r[7] * f1 + r[8] * f2;
  }
}

// dfeat[src_of[d]][b] = rel^T @ dx_rot
__global__ __launch_bounds__(256) void rotcat_bwd_feat_kernel(const float *__restrict__ dx, const float *__restrict__ rel,
                                                              const int *__restrict__ src_of, float *__restrict__ dfeat,
                                                              int batch, int cf, int nvec) {
  const int row = blockIdx.x;
  const int d = row / batch, b = row - d * batch;
  const int kin = cf + 3 * nvec;
  const float *g = dx + (long long)row * kin + cf;
  float *o = dfeat + ((long long)src_of[d] * batch + b) * 3 * nvec;
  float r[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (rel) {
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = rel[(long long)row * 9 + k];
  }
  for (int k = threadIdx.x; k < nvec; k += 256) {
    const float g0 = g[k], g1 = g[nvec + k], g2 = g[2 * nvec + k];
    o[k] = r[0] * g0 + r[3] * g1 + r[6] * g2;
    o[nvec + k] = r[1] * g0 + r[4] * g1 + r[7] * g2;
    o[2 * nvec + k] = r[2] * g0 + r[5] * g1 + r[8] * g2;
  }
}

// ---- ablation variants (rot_mv.py:53-86,136-171) -------------------------------------------------
// encode_rotmat: x[(d,b)] = [ img_feat | feat (NOT rotated) | rel flattened (9) | zeros to ld ]
__global__ __launch_bounds__(256) void rotcat_ext_fwd_kernel(const float *__restrict__ img_feat,
                                                             const float *__restrict__ feat,
                                                             const float *__restrict__ rel_apply,
                                                             const float *__restrict__ rel_append,
                                                             const int *__restrict__ view_of, const int *__restrict__ src_of,
                                                             float *__restrict__ x, int ld, int batch, int cf, int nvec) {
  const int row = blockIdx.x;            // d*batch + b
  const int d = row / batch, b = row - d * batch;
  float *xo = x + (long long)row * ld;
  const float4 *src = reinterpret_cast<const float4 *>(img_feat + ((long long)view_of[d] * batch + b) * cf);
  for (int i = threadIdx.x; i < cf / 4; i += 256) reinterpret_cast<float4 *>(xo)[i] = src[i];
  const float *f = feat + ((long long)src_of[d] * batch + b) * 3 * nvec;
  float r[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (rel_apply) {
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = rel_apply[(long long)row * 9 + k];
  }
  for (int k = threadIdx.x; k < nvec; k += 256) {
    const float f0 = f[k], f1 = f[nvec + k], f2 = f[2 * nvec + k];
    xo[cf + k] = r[0] * f0 + r[1] * f1 + r[2] * f2;
    xo[cf + nvec + k] = r[3] * f0 + r[4] * f1 + r[5] * f2;
    xo[cf + 2 * nvec + k] = r[6] * f0 + r[7] * f1 + r[8] * f2;
  }
  const int tail = cf + 3 * nvec;
  for (int k = threadIdx.x; tail + k < ld; k += 256)
    xo[tail + k] = (rel_append && k < 9) ? rel_append[(long long)row * 9 + k] : 0.f;
}

__global__ __launch_bounds__(256) void rotcat_ext_bwd_kernel(const float *__restrict__ dx, int ld,
                                                             const float *__restrict__ rel, const int *__restrict__ src_of,
                                                             float *__restrict__ dfeat, int batch, int cf, int nvec) {
  const int row = blockIdx.x;
  const int d = row / batch, b = row - d * batch;
  const float *g = dx + (long long)row * ld + cf;
  float *o = dfeat + ((long long)src_of[d] * batch + b) * 3 * nvec;
  float r[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (rel) {
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = rel[(long long)row * 9 + k];
  }
  for (int k = threadIdx.x; k < nvec; k += 256) {
    const float g0 = g[k], g1 = g[nvec + k], g2 = g[2 * nvec + k];
    o[k] = r[0] * g0 + r[3] * g1 + r[6] * g2;
    o[nvec + k] = r[1] * g0 + r[4] * g1 + r[7] * g2;
    o[2 * nvec + k] = r[2] * g0 + r[5] * g1 + r[8] * g2;
  }
}

// share_feature: the IntensityBatchNorm scales of one iteration.  The reference calls the fuser's ONE
// normaliser 2*dirs times in sequence (direction 0: feat_0 then feat_1, direction 1: ...); in training
// every call first moves running_mean towards the batch std of the column norms, then divides by
// (running_mean + eps).  One thread per column k walks the calls in that order.  The norm of a
// rotated column equals the norm of the column (rel is orthonormal), so feat is read unrotated.
__global__ __launch_bounds__(256) void ibn_scales_kernel(const float *__restrict__ a, const float *__restrict__ feat,
                                                         const int *__restrict__ view_of, const int *__restrict__ src_of,
                                                         float *__restrict__ running_mean, int training, float momentum,
                                                         float eps, float *__restrict__ scales, int batch, int dirs,
                                                         int nvec) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nvec) return;
  float rm = running_mean[k];
  for (int call = 0; call < 2 * dirs; ++call) {
    const int d = call >> 1;
    const float *src = (call & 1) ? feat + (long long)src_of[d] * batch * 3 * nvec : a + (long long)view_of[d] * batch * 3 * nvec;
    if (training) {
      // biased variance over the batch of the column norm (two passes, like torch.var)
      double sum = 0.0;
      for (int b = 0; b < batch; ++b) {
        const float *x = src + (long long)b * 3 * nvec + k;
        const float x0 = x[0], x1 = x[nvec], x2 = x[2 * nvec];
        sum += (double)sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
      }
      const double mean = sum / batch;
      double m2 = 0.0;
      for (int b = 0; b < batch; ++b) {
        const float *x = src + (long long)b * 3 * nvec + k;
        const float x0 = x[0], x1 = x[nvec], x2 = x[2 * nvec];
        const double dlt = (double)sqrtf(x0 * x0 + x1 * x1 + x2 * x2) - mean;
        m2 += dlt * dlt;
      }
      const float var = (float)(m2 / batch);
      const float sd = sqrtf(fmaxf(var, eps));
      rm = rm * (1.f - momentum) + sd * momentum;
    }
    scales[(long long)call * nvec + k] = 1.f / (rm + eps);
  }
  if (training) running_mean[k] = rm;
}

// x[(d,b)][axis][0:nvec] = s0 * a[view_of[d]][b][axis][:],  [nvec:2nvec] = s1 * (rel @ feat[src_of[d]][b])[axis][:]
__global__ __launch_bounds__(256) void paircat_fwd_kernel(const float *__restrict__ a, const float *__restrict__ feat,
                                                          const float *__restrict__ rel, const float *__restrict__ scales,
                                                          const int *__restrict__ view_of, const int *__restrict__ src_of,
                                                          float *__restrict__ x, int batch, int nvec) {
  const int row = blockIdx.x;
  const int d = row / batch, b = row - d * batch;
  float *xo = x + (long long)row * 6 * nvec;
  const float *pa = a + ((long long)view_of[d] * batch + b) * 3 * nvec;
  const float *f = feat + ((long long)src_of[d] * batch + b) * 3 * nvec;
  float r[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (rel) {
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = rel[(long long)row * 9 + k];
  }
  for (int k = threadIdx.x; k < nvec; k += 256) {
    const float s0 = scales ? scales[(long long)(2 * d) * nvec + k] : 1.f;
    const float s1 = scales ? scales[(long long)(2 * d + 1) * nvec + k] : 1.f;
    const float f0 = f[k], f1 = f[nvec + k], f2 = f[2 * nvec + k];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      xo[ax * 2 * nvec + k] = pa[ax * nvec + k] * s0;
      xo[ax * 2 * nvec + nvec + k] = (r[3 * ax] * f0 + r[3 * ax + 1] * f1 + r[3 * ax + 2] * f2) * s1;
    }
  }
}

// da_dir[(d,b)][axis][k] = s0 * dx[..][axis][k];   dfeat[src_of[d]][b] = rel^T @ (s1 * dx[..][axis][nvec + k])
__global__ __launch_bounds__(256) void paircat_bwd_kernel(const float *__restrict__ dx, const float *__restrict__ rel,
                                                          const float *__restrict__ scales, const int *__restrict__ src_of,
                                                          float *__restrict__ da_dir, float *__restrict__ dfeat, int batch,
                                                          int nvec) {
  const int row = blockIdx.x;
  const int d = row / batch, b = row - d * batch;
  const float *g = dx + (long long)row * 6 * nvec;
  float *oa = da_dir + (long long)row * 3 * nvec;
  float *of = dfeat + ((long long)src_of[d] * batch + b) * 3 * nvec;
  float r[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (rel) {
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = rel[(long long)row * 9 + k];
  }
  for (int k = threadIdx.x; k < nvec; k += 256) {
    const float s0 = scales ? scales[(long long)(2 * d) * nvec + k] : 1.f;
    const float s1 = scales ? scales[(long long)(2 * d + 1) * nvec + k] : 1.f;
    float gr[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      oa[ax * nvec + k] = g[ax * 2 * nvec + k] * s0;
      gr[ax] = g[ax * 2 * nvec + nvec + k] * s1;
    }
    of[k] = r[0] * gr[0] + r[3] * gr[1] + r[6] * gr[2];
    of[nvec + k] = r[1] * gr[0] + r[4] * gr[1] + r[7] * gr[2];
    of[2 * nvec + k] = r[2] * gr[0] + r[5] * gr[1] + r[8] * gr[2];
  }
}

// ---- the fusion block on the split-operand kernels: operand builders that write sp directly ---------------------------
// max |x| of up to 8 small tensors in one launch (grid.y = tensor): out[i] receives the float's bits by atomicMax (the
// caller clears the slots); the bounds the builders below scale by (image features, lifted features, hidden-layer biases)
// max over the workgroup (256 threads) of a non-negative value, then ONE atomicMax of its bits (atomics on one address
// serialise at the memory side, ~12 ns each: one per wave of a 1500-workgroup launch costs more than the launch's work)
__device__ __forceinline__ void block_atomic_absmax(float m, unsigned *__restrict__ slot) {
  __shared__ float wred[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    if (t > 0.f) atomicMax(slot, __float_as_uint(t));
  }
}
struct AbsMaxItems {
  const float *p[8];
  long long n[8];
  unsigned *out[8];
};
__global__ __launch_bounds__(256) void absmax_multi_kernel(AbsMaxItems it) {
  const float *__restrict__ x = it.p[blockIdx.y];
  const long long n = it.n[blockIdx.y];
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  block_atomic_absmax(m, it.out[blockIdx.y]);
}

__device__ __forceinline__ void ld8f(const float *p, float (&v)[8]) {
  const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st_sp_chunk(uint4 *dst, const float (&v)[8], float scale) {
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = v[k] * scale;
  uint4 q1, q2;
  split2_chunk(w, q1, q2);
  dst[0] = q1;
  dst[1] = q2;
}

// One launch builds, in sp and scaled, the two operands that depend on a feature tensor F (rot_mv.py:44-50,234-239,249-254):
//   xf[m] = [ img[row_img[m]] | rel[m] @ F[row_src_f[m]] ]   the NEXT fusion iteration's fuser input (may be null)
//   xh[m] = [ img[row_img[m]] |          F[row_src_h[m]] ]   this iteration's gaze-head input       (may be null)
// Scales: |rel @ f| <= |f|_2 <= sqrt(3) max |f| (rel is orthonormal), so xf is stored times the 2^k that
// max(am_img, sqrt(3) am_feat) allows and xh times the one max(am_img, am_feat) allows (am_*: float bits left by
// absmax_multi_kernel / the producing GEMM's epilogue); *xf_sinv / *xh_sinv receive 2^-k.  One workgroup per row.
__global__ __launch_bounds__(256) void fuse_build_split_kernel(const float *__restrict__ img, const float *__restrict__ feat,
                                                               const float *__restrict__ rel, const int *__restrict__ row_img,
                                                               const int *__restrict__ row_src_f, const int *__restrict__ row_src_h,
                                                               uint4 *__restrict__ xf, uint4 *__restrict__ xh,
                                                               const unsigned *__restrict__ am_img, const unsigned *__restrict__ am_feat,
                                                               float *__restrict__ xf_sinv, float *__restrict__ xh_sinv, int cf, int nvec) {
  const int m = blockIdx.x;
  const float ai = __uint_as_float(*am_img), af = __uint_as_float(*am_feat);
  const float sf = sp_scale_for(fmaxf(ai, 1.7320509f * af)), sh = sp_scale_for(fmaxf(ai, af));
  if (m == 0 && threadIdx.x == 0) {
    if (xf) *xf_sinv = 1.f / sf;
    if (xh) *xh_sinv = 1.f / sh;
  }
  const int c8i = cf >> 3, c8v = nvec >> 3, c8row = c8i + 3 * c8v;
  const float *src = img + (long long)row_img[m] * cf;
  for (int c8 = threadIdx.x; c8 < c8i; c8 += 256) {
    float v[8];
    ld8f(src + 8 * c8, v);
    if (xf) st_sp_chunk(xf + SP_NP * ((long long)m * c8row + c8), v, sf);
    if (xh) st_sp_chunk(xh + SP_NP * ((long long)m * c8row + c8), v, sh);
  }
  float r[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (rel) {
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = rel[(long long)m * 9 + k];
  }
  for (int t = threadIdx.x; t < 3 * c8v; t += 256) {
    const int a = t / c8v, k8 = t - a * c8v;
    if (xf) {
      const float *f = feat + (long long)row_src_f[m] * 3 * nvec + 8 * k8;
      float f0[8], f1[8], f2[8], o[8];
      ld8f(f, f0);
      ld8f(f + nvec, f1);
      ld8f(f + 2 * nvec, f2);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = r[3 * a] * f0[k] + r[3 * a + 1] * f1[k] + r[3 * a + 2] * f2[k];     // rotcat_fwd_kernel's expression
      st_sp_chunk(xf + SP_NP * ((long long)m * c8row + c8i + t), o, sf);
    }
    if (xh) {
      float o[8];
      ld8f(feat + (long long)row_src_h[m] * 3 * nvec + a * nvec + 8 * k8, o);
      st_sp_chunk(xh + SP_NP * ((long long)m * c8row + c8i + t), o, sh);
    }
  }
}

// The backward of the builders, ONE launch per iteration: from the gradients of the operands that were built from F
// (dxh: the head input's, rows = directions; dxn: the next iteration's fuser input's) to
//   dF[(s,b)] = dxh[(s,b)][cf:] + sum over d with seg[d] == s (ascending) of rel[(d,b)]^T @ dxn[(d,b)][cf:]        blocks [0, S B)
//   da[(v,b)] (+)= sum over d with vi[d] == v of (dxh[(d,b)][:cf] + dxn[(d,b)][:cf])                                blocks [S B, S B + V B)
// (segments: the partner direction for F_it, the source VIEW for the lifted features of iteration 0) and max |dF| into
// *absmax (float bits, atomicMax) for the split of dF that follows.
__global__ __launch_bounds__(256) void fuse_unbuild_kernel(const float *__restrict__ dxh, const float *__restrict__ dxn,
                                                           const float *__restrict__ rel, const int *__restrict__ seg,
                                                           const int *__restrict__ vi, float *__restrict__ dF, float *__restrict__ da,
                                                           int da_accumulate, int S, int V, int D, int B, int cf, int nvec,
                                                           unsigned *__restrict__ absmax, int rows_per_block) {
  const int kin = cf + 3 * nvec;
  const int fblocks = (S * B + rows_per_block - 1) / rows_per_block;
  if ((int)blockIdx.x < fblocks) {
    float mx = 0.f;
    for (int rr = 0; rr < rows_per_block; ++rr) {
      const int row_o = blockIdx.x * rows_per_block + rr;
      if (row_o >= S * B) break;
      const int s = row_o / B, b = row_o - s * B;
      for (int k = threadIdx.x; k < nvec; k += 256) {
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        if (dxh) {
          const float *g = dxh + ((long long)s * B + b) * kin + cf;
          o0 = g[k];
          o1 = g[nvec + k];
          o2 = g[2 * nvec + k];
        }
        if (dxn)
          for (int d = 0; d < D; ++d) {
            if (seg[d] != s) continue;
            const long long row = (long long)d * B + b;
            const float *g = dxn + row * kin + cf;
            const float g0 = g[k], g1 = g[nvec + k], g2 = g[2 * nvec + k];
            if (rel) {
              const float *r = rel + row * 9;
              o0 += r[0] * g0 + r[3] * g1 + r[6] * g2;             // rotcat_bwd_feat_kernel's expressions
              o1 += r[1] * g0 + r[4] * g1 + r[7] * g2;
              o2 += r[2] * g0 + r[5] * g1 + r[8] * g2;
            } else {
              o0 += g0;
              o1 += g1;
              o2 += g2;
            }
          }
        float *o = dF + ((long long)s * B + b) * 3 * nvec;
        o[k] = o0;
        o[nvec + k] = o1;
        o[2 * nvec + k] = o2;
        mx = fmaxf(mx, fmaxf(fabsf(o0), fmaxf(fabsf(o1), fabsf(o2))));
      }
    }
    if (absmax) block_atomic_absmax(mx, absmax);
    return;
  }
  const int t = blockIdx.x - fblocks;
  const int v = t / B, b = t - v * B;
  for (int cq = threadIdx.x; cq < cf / 4; cq += 256) {
    float4 *o = reinterpret_cast<float4 *>(da + ((long long)v * B + b) * cf) + cq;
    float4 acc = da_accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int d = 0; d < D; ++d) {
      if (vi[d] != v) continue;
      const long long row = (long long)d * B + b;
      if (dxh) {
        const float4 g = reinterpret_cast<const float4 *>(dxh + row * kin)[cq];
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      }
      if (dxn) {
        const float4 g = reinterpret_cast<const float4 *>(dxn + row * kin)[cq];
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      }
    }
    *o = acc;
  }
}

// out[v][b][0:width] (+)= sum over d with seg_of[d] == v (ascending d: reproducible) of x rows
__global__ __launch_bounds__(256) void segment_sum_kernel(const float *__restrict__ x, long long row_stride, int w4n,
                                                          const int *__restrict__ seg_of, float *__restrict__ out,
                                                          int batch, int dirs, int segments, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over segments*batch*width/4
  if (i >= (long long)segments * batch * w4n) return;
  const int cq = (int)(i % w4n);
  const long long t = i / w4n;
  const int b = (int)(t % batch), v = (int)(t / batch);
  float4 s = accumulate ? reinterpret_cast<float4 *>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int d = 0; d < dirs; ++d) {
    if (seg_of[d] != v) continue;
    const float4 g = reinterpret_cast<const float4 *>(x + ((long long)d * batch + b) * row_stride)[cq];
    s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
  }
  reinterpret_cast<float4 *>(out)[i] = s;
}

__global__ __launch_bounds__(256) void scale_by_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                       float *__restrict__ out, long long n) {
  const float s = scale[0];
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = x[i] * s;
}

// out[c] (+)= sum_rows x[rows][c] : block = 64 channels x 4 row lanes
__global__ __launch_bounds__(256) void axpby_kernel(const float *__restrict__ x, float *__restrict__ y, float a, float b,
                                                    long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}

// ---- Adam (torch.optim.Adam semantics: coupled L2 weight decay, bias correction) over flat arenas --
// trainer.py:54 optim.Adam(params, lr, weight_decay=1e-6); one launch updates every parameter.
__global__ __launch_bounds__(256) void adam_kernel(float4 *__restrict__ p, const float4 *__restrict__ g,
                                                   float4 *__restrict__ m, float4 *__restrict__ v, long long n4, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  const long long stride = (long long)gridDim.x * 256;
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
#define MVG_ADAM1(c)                                             \
    {                                                            \
      const float gr = gg.c + wd * pp.c;                         \
      mm.c = b1 * mm.c + (1.f - b1) * gr;                        \
      vv.c = b2 * vv.c + (1.f - b2) * gr * gr;                   \
      const float denom = sqrtf(vv.c) / bc2_sqrt + eps;          \
      pp.c = pp.c - step_size * (mm.c / denom);                  \
    }
    MVG_ADAM1(x) MVG_ADAM1(y) MVG_ADAM1(z) MVG_ADAM1(w)
#undef MVG_ADAM1
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}

// ---- skinny linear (out_features <= 4): the Linear(512 -> 2) of the gaze head ------------------
// fwd: one wave per row.
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, float *__restrict__ y,
                                                         int rows, int k, int nout) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < k; i += 64) {
    const float v = x[(long long)row * k + i];
    for (int j = 0; j < nout; ++j) acc[j] += v * w[(long long)j * k + i];
  }
  for (int j = 0; j < nout; ++j) {
    const float s = wave_sum(acc[j]);
    if (lane == 0) y[(long long)row * nout + j] = s + (bias ? bias[j] : 0.f);
  }
}
// bwd: dx[m][i] = mask * sum_j dy[m][j] w[j][i]
__global__ __launch_bounds__(256) void skinny_bwd_dx_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                            const float *__restrict__ mask, float *__restrict__ dx,
                                                            long long total, int k, int nout, unsigned *__restrict__ absmax) {
  float mx = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / k;
    const int col = (int)(i - m * k);
    float s = 0.f;
    for (int j = 0; j < nout; ++j) s += dy[m * nout + j] * w[(long long)j * k + col];
    if (mask && !(mask[i] > 0.f)) s = 0.f;
    dx[i] = s;
    mx = fmaxf(mx, fabsf(s));
  }
  if (absmax) block_atomic_absmax(mx, absmax);     // max |dx| for the split of dx that follows (float bits, order-independent)
}
// dw[j][i] = sum_m dy[m][j] x[m][i];  db[j] = sum_m dy[m][j]
// 4 columns x 64 row lanes per workgroup (k / 4 workgroups: 128 for the gaze head's 512 inputs); the row lanes are summed
// through LDS in lane order (deterministic).  The bias gradient is summed by workgroup 0 with all of its threads - one
// thread per output walking all rows serially was 290 us of this kernel's 296 at C5's 3584 rows.
__global__ __launch_bounds__(256) void skinny_bwd_dw_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            float *__restrict__ dw, float *__restrict__ db, int rows,
                                                            int k, int nout, int accumulate) {
  __shared__ float sh[4][64][5];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int i = blockIdx.x * 4 + cl;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < k)
    for (int m = rl; m < rows; m += 64) {
      const float v = x[(long long)m * k + i];
      for (int j = 0; j < nout; ++j) acc[j] += dy[(long long)m * nout + j] * v;
    }
  for (int j = 0; j < 4; ++j) sh[j][rl][cl] = acc[j];
  __syncthreads();
  if (rl < nout && i < k) {
    const int j = rl;
    float t = 0.f;
    for (int r = 0; r < 64; ++r) t += sh[j][r][cl];
    dw[(long long)j * k + i] = (accumulate ? dw[(long long)j * k + i] : 0.f) + t;
  }
  if (db && blockIdx.x == 0) {
    __shared__ float sb[4][256];
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = threadIdx.x; m < rows; m += 256)
      for (int j = 0; j < nout; ++j) part[j] += dy[(long long)m * nout + j];
    for (int j = 0; j < 4; ++j) sb[j][threadIdx.x] = part[j];
    __syncthreads();
    if (threadIdx.x < nout) {
      float t = 0.f;
      for (int r = 0; r < 256; ++r) t += sb[threadIdx.x][r];
      db[threadIdx.x] = (accumulate ? db[threadIdx.x] : 0.f) + t;
    }
  }
}

// ---- L1 / L2 loss of GazeLoss (losses/gaze_loss.py:56-64): mean |a - b|^p over all elements ----------
__global__ __launch_bounds__(256) void gaze_lp_loss_kernel(const float *__restrict__ pred, const float *__restrict__ label,
                                                           int n, int p, float *__restrict__ loss,
                                                           float *__restrict__ dpred) {
  __shared__ double sh[4];
  double local = 0.0;
  const float inv_n = 1.f / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = pred[i] - label[i];
    const float a = fabsf(d);
    local += (double)(p == 1 ? a : a * a);
    if (dpred) {
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);      // torch.abs: zero gradient at 0
      dpred[i] = (p == 1 ? sgn : 2.f * a * sgn) * inv_n;
    }
  }
  local = wave_sum_d(local);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (float)((sh[0] + sh[1] + sh[2] + sh[3]) / (double)n);
}

// ---- angular loss ------------------------------------------------------------------------------
// v(p,y) = (cos p sin y, sin p, cos p cos y); sim = <u/|u|, v/|v|> with the norms clamped at 1e-6
// (ATen cosine_similarity); clamp to [-1,1] (hardtanh: zero gradient AT and beyond the bounds);
// theta = acos(sim) * 180/pi.
__global__ __launch_bounds__(256) void gaze_loss_kernel(const float *__restrict__ pred, const float *__restrict__ gt,
                                                        int n, float row_weight, float *__restrict__ loss,
                                                        int accumulate, float *__restrict__ dpred,
                                                        float *__restrict__ theta_out) {
  __shared__ double sh[4];
  double local = 0.0;
  const float k180 = 57.29577951308232f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float pp = pred[2 * i], py = pred[2 * i + 1], gp = gt[2 * i], gy = gt[2 * i + 1];
    const float cpp = cosf(pp), spp = sinf(pp), cpy = cosf(py), spy = sinf(py);
    const float cgp = cosf(gp), sgp = sinf(gp), cgy = cosf(gy), sgy = sinf(gy);
    const float v0 = cpp * spy, v1 = spp, v2 = cpp * cpy;
    const float u0 = cgp * sgy, u1 = sgp, u2 = cgp * cgy;
    const float nv = fmaxf(sqrtf(v0 * v0 + v1 * v1 + v2 * v2), 1e-6f);
    const float nu = fmaxf(sqrtf(u0 * u0 + u1 * u1 + u2 * u2), 1e-6f);
    const float a0 = u0 / nu, a1 = u1 / nu, a2 = u2 / nu;
    const float b0 = v0 / nv, b1 = v1 / nv, b2 = v2 / nv;
    const float sim = a0 * b0 + a1 * b1 + a2 * b2;
    const float sc = fminf(fmaxf(sim, -1.f), 1.f);
    const float theta = acosf(sc) * k180;
    local += (double)theta;
    if (theta_out) theta_out[i] = theta;
    if (dpred) {
      float gpitch = 0.f, gyaw = 0.f;
      if (sim > -1.f && sim < 1.f) {
        const float dth = -k180 / sqrtf(1.f - sc * sc);          // dtheta/dsim
        // dsim/dv = (a - sim*b)/|v|   (gradient through the normalisation)
        const float d0 = (a0 - sim * b0) / nv, d1 = (a1 - sim * b1) / nv, d2 = (a2 - sim * b2) / nv;
        // dv/dpitch = (-sin p sin y, cos p, -sin p cos y); dv/dyaw = (cos p cos y, 0, -cos p sin y)
        gpitch = dth * (d0 * (-spp * spy) + d1 * cpp + d2 * (-spp * cpy));
        gyaw = dth * (d0 * (cpp * cpy) + d2 * (-cpp * spy));
      }
      dpred[2 * i] = row_weight * gpitch;
      dpred[2 * i + 1] = row_weight * gyaw;
    }
  }
  local = wave_sum_d(local);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = (sh[0] + sh[1] + sh[2] + sh[3]) * (double)row_weight;
    loss[0] = (accumulate ? loss[0] : 0.f) + (float)tot;
  }
}


// All iterations' angular losses in ONE launch (stereo_loss.py:65-84 over the head's stacked predictions): pred [iters][n][2],
// gt [n][2], row r of iteration i weighs w[i * dirs + r / batch] / batch; loss = the weighted sum (fp64 accumulation, one
// workgroup: fixed order), dpred = the weighted analytic gradient.
struct LossWeights {
  float w[512];
};
__global__ __launch_bounds__(256) void gaze_loss_multi_kernel(const float *__restrict__ pred, const float *__restrict__ gt, int iters,
                                                              int dirs, int batch, LossWeights lw, float *__restrict__ loss,
                                                              float *__restrict__ dpred) {
  __shared__ double sh[4];
  double local = 0.0;
  const float k180 = 57.29577951308232f;
  const int n = dirs * batch;
  const float inv_b = 1.f / (float)batch;
  for (int i = threadIdx.x; i < iters * n; i += 256) {
    const int it = i / n, r = i - it * n;
    const float row_weight = lw.w[it * dirs + r / batch] * inv_b;
    const float pp = pred[2 * i], py = pred[2 * i + 1], gp = gt[2 * r], gy = gt[2 * r + 1];
    const float cpp = cosf(pp), spp = sinf(pp), cpy = cosf(py), spy = sinf(py);
    const float cgp = cosf(gp), sgp = sinf(gp), cgy = cosf(gy), sgy = sinf(gy);
    const float v0 = cpp * spy, v1 = spp, v2 = cpp * cpy;
    const float u0 = cgp * sgy, u1 = sgp, u2 = cgp * cgy;
    const float nv = fmaxf(sqrtf(v0 * v0 + v1 * v1 + v2 * v2), 1e-6f);
    const float nu = fmaxf(sqrtf(u0 * u0 + u1 * u1 + u2 * u2), 1e-6f);
    const float a0 = u0 / nu, a1 = u1 / nu, a2 = u2 / nu;
    const float b0 = v0 / nv, b1 = v1 / nv, b2 = v2 / nv;
    const float sim = a0 * b0 + a1 * b1 + a2 * b2;
    const float sc = fminf(fmaxf(sim, -1.f), 1.f);
    const float theta = acosf(sc) * k180;
    local += (double)theta * (double)row_weight;
    if (dpred) {
      float gpitch = 0.f, gyaw = 0.f;
      if (sim > -1.f && sim < 1.f) {
        const float dth = -k180 / sqrtf(1.f - sc * sc);
        const float d0 = (a0 - sim * b0) / nv, d1 = (a1 - sim * b1) / nv, d2 = (a2 - sim * b2) / nv;
        gpitch = dth * (d0 * (-spp * spy) + d1 * cpp + d2 * (-spp * cpy));
        gyaw = dth * (d0 * (cpp * cpy) + d2 * (-cpp * spy));
      }
      dpred[2 * i] = row_weight * gpitch;
      dpred[2 * i + 1] = row_weight * gyaw;
    }
  }
  local = wave_sum_d(local);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// Adam with its step counter and learning rate on the DEVICE (a captured step replays with nothing from the host):
// adam_tick_kernel advances state = {step, bias correction 1, sqrt(bias correction 2)} (one thread), adam_dev_kernel is
// adam_kernel reading lr from hyper[0] and the corrections from state.
__global__ void adam_tick_kernel(float *__restrict__ state, float b1, float b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float step = state[0] + 1.f;
  state[0] = step;
  state[1] = (float)(1.0 - pow((double)b1, (double)step));
  state[2] = (float)sqrt(1.0 - pow((double)b2, (double)step));
}
__global__ __launch_bounds__(256) void adam_dev_kernel(float4 *__restrict__ p, const float4 *__restrict__ g, float4 *__restrict__ m,
                                                       float4 *__restrict__ v, long long n4, const float *__restrict__ lr_dev,
                                                       const float *__restrict__ state, float b1, float b2, float eps, float wd) {
  const long long stride = (long long)gridDim.x * 256;
  const float step_size = lr_dev[0] / state[1], bc2_sqrt = state[2];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
#define MVG_ADAM1(c)                                             \
    {                                                            \
      const float gr = gg.c + wd * pp.c;                         \
      mm.c = b1 * mm.c + (1.f - b1) * gr;                        \
      vv.c = b2 * vv.c + (1.f - b2) * gr * gr;                   \
      const float denom = sqrtf(vv.c) / bc2_sqrt + eps;          \
      pp.c = pp.c - step_size * (mm.c / denom);                  \
    }
    MVG_ADAM1(x) MVG_ADAM1(y) MVG_ADAM1(z) MVG_ADAM1(w)
#undef MVG_ADAM1
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}

}  // namespace mvg

using namespace mvg;

extern "C" {

int mvg_rotation_matrix_2d(const float *pitch_yaw, float *rot, int n, int inverse, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_GEOMETRY, st, 0.0, 44.0 * n);
  hipLaunchKernelGGL(rotation_matrix_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, pitch_yaw, rot, n, inverse);
  return check_launch("rotation_matrix_2d");
}

int mvg_relative_rotation(const float *rot, const int32_t *vi, const int32_t *vj, float *rel, int batch, int views,
                          int dirs, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_GEOMETRY, st, 0.0, 108.0 * dirs * batch);
  hipLaunchKernelGGL(relative_rotation_kernel, dim3(ceil_div((long long)dirs * batch, 256)), dim3(256), 0, st, rot, vi, vj,
                     rel, batch, views, dirs);
  return check_launch("relative_rotation");
}

int mvg_rotcat_fwd(const float *img_feat, const float *feat, const float *rel, const int32_t *view_of,
                   const int32_t *src_of, float *x, int batch, int dirs, int cf, int nvec, void *stream) {
  MVG_REQUIRE(cf % 4 == 0, "rotcat: cf %% 4 != 0");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ROTCAT, st, 18.0 * dirs * batch * nvec, 8.0 * dirs * (double)batch * (cf + 3 * nvec));
  hipLaunchKernelGGL(rotcat_fwd_kernel, dim3(dirs * batch), dim3(256), 0, st, img_feat, feat, rel, view_of, src_of, x,
                     batch, cf, nvec);
  return check_launch("rotcat_fwd");
}

int mvg_rotcat_bwd(const float *dx, const float *rel, const int32_t *src_of, float *dfeat, int batch, int dirs, int cf,
                   int nvec, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ROTCAT, st, 18.0 * dirs * batch * nvec, 24.0 * dirs * (double)batch * nvec);
  hipLaunchKernelGGL(rotcat_bwd_feat_kernel, dim3(dirs * batch), dim3(256), 0, st, dx, rel, src_of, dfeat, batch, cf, nvec);
  return check_launch("rotcat_bwd");
}

int mvg_rotcat_ext_fwd(const float *img_feat, const float *feat, const float *rel_apply, const float *rel_append,
                       const int32_t *view_of, const int32_t *src_of, float *x, int ld, int batch, int dirs, int cf, int nvec,
                       void *stream) {
  MVG_REQUIRE(cf % 4 == 0 && ld % 4 == 0, "rotcat_ext: cf / ld %% 4 != 0");
  MVG_REQUIRE(ld >= cf + 3 * nvec + (rel_append ? 9 : 0), "rotcat_ext: ld too small");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ROTCAT, st, 18.0 * dirs * batch * nvec, 8.0 * dirs * (double)batch * ld);
  hipLaunchKernelGGL(rotcat_ext_fwd_kernel, dim3(dirs * batch), dim3(256), 0, st, img_feat, feat, rel_apply, rel_append, view_of,
                     src_of, x, ld, batch, cf, nvec);
  return check_launch("rotcat_ext_fwd");
}

int mvg_rotcat_ext_bwd(const float *dx, int ld, const float *rel, const int32_t *src_of, float *dfeat, int batch, int dirs,
                       int cf, int nvec, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ROTCAT, st, 18.0 * dirs * batch * nvec, 24.0 * dirs * (double)batch * nvec);
  hipLaunchKernelGGL(rotcat_ext_bwd_kernel, dim3(dirs * batch), dim3(256), 0, st, dx, ld, rel, src_of, dfeat, batch, cf, nvec);
  return check_launch("rotcat_ext_bwd");
}

int mvg_ibn_scales(const float *a, const float *feat, const int32_t *view_of, const int32_t *src_of, float *running_mean,
                   int training, float momentum, float eps, float *scales, int batch, int dirs, int nvec, void *stream) {
  MVG_REQUIRE(batch > 0 && dirs > 0 && nvec > 0, "ibn_scales: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ELEMENTWISE, st, 0.0, 24.0 * dirs * (double)batch * nvec * 2);
  hipLaunchKernelGGL(ibn_scales_kernel, dim3(ceil_div(nvec, 256)), dim3(256), 0, st, a, feat, view_of, src_of, running_mean,
                     training, momentum, eps, scales, batch, dirs, nvec);
  return check_launch("ibn_scales");
}

int mvg_paircat_fwd(const float *a, const float *feat, const float *rel, const float *scales, const int32_t *view_of,
                    const int32_t *src_of, float *x, int batch, int dirs, int nvec, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ROTCAT, st, 18.0 * dirs * batch * nvec, 48.0 * dirs * (double)batch * nvec);
  hipLaunchKernelGGL(paircat_fwd_kernel, dim3(dirs * batch), dim3(256), 0, st, a, feat, rel, scales, view_of, src_of, x, batch,
                     nvec);
  return check_launch("paircat_fwd");
}

int mvg_paircat_bwd(const float *dx, const float *rel, const float *scales, const int32_t *src_of, float *da_dir, float *dfeat,
                    int batch, int dirs, int nvec, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ROTCAT, st, 18.0 * dirs * batch * nvec, 48.0 * dirs * (double)batch * nvec);
  hipLaunchKernelGGL(paircat_bwd_kernel, dim3(dirs * batch), dim3(256), 0, st, dx, rel, scales, src_of, da_dir, dfeat, batch,
                     nvec);
  return check_launch("paircat_bwd");
}

int mvg_segment_sum(const float *x, int64_t row_stride, int width, const int32_t *seg_of, float *out, int batch, int dirs,
                    int segments, int accumulate, void *stream) {
  MVG_REQUIRE(width % 4 == 0 && row_stride % 4 == 0, "segment_sum: width/row_stride %% 4 != 0");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)segments * batch * (width / 4);
  ProfScope ps(MVG_K_ELEMENTWISE, st, 0.0, 4.0 * (double)batch * width * (dirs + segments));
  hipLaunchKernelGGL(segment_sum_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, x, (long long)row_stride, width / 4,
                     seg_of, out, batch, dirs, segments, accumulate);
  return check_launch("segment_sum");
}

int mvg_scale_by(const float *x, const float *scale, float *out, int64_t n, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ELEMENTWISE, st, 0.0, 8.0 * (double)n);
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(scale_by_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, scale, out, (long long)n);
  return check_launch("scale_by");
}

int mvg_axpby(const float *x, float *y, float a, float b, int64_t n, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ELEMENTWISE, st, 0.0, 12.0 * (double)n);
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, a, b, (long long)n);
  return check_launch("axpby");
}

int mvg_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void *stream) {
  MVG_REQUIRE(n % 4 == 0 && step >= 1, "adam: n %% 4 != 0 or step < 1");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ELEMENTWISE, st, 0.0, 28.0 * (double)n);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (float4 *)param, (const float4 *)grad,
                     (float4 *)exp_avg, (float4 *)exp_avg_sq, (long long)(n / 4), lr, beta1, beta2, eps, weight_decay,
                     (float)bc1, (float)sqrt(bc2));
  return check_launch("adam_step");
}

int mvg_absmax_multi(const float *const *host_ptrs, const int64_t *host_counts, float *const *host_out_slots, int n, void *stream) {
  MVG_REQUIRE(host_ptrs && host_counts && host_out_slots && n >= 1 && n <= 8, "absmax_multi: 1..8 tensors");
  AbsMaxItems it;
  memset(&it, 0, sizeof(it));
  long long most = 0;
  for (int i = 0; i < n; ++i) {
    MVG_REQUIRE(host_ptrs[i] && host_out_slots[i] && host_counts[i] >= 0, "absmax_multi: null tensor / slot");
    it.p[i] = host_ptrs[i];
    it.n[i] = host_counts[i];
    it.out[i] = (unsigned *)host_out_slots[i];
    if (host_counts[i] > most) most = host_counts[i];
  }
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ELEMENTWISE, st, 0.0, 0.0);
  long long blocks = (most + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(absmax_multi_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, st, it);
  return check_launch("absmax_multi");
}

int mvg_fuse_build_split(const float *img_feat, const float *feat, const float *rel, const int32_t *row_img, const int32_t *row_src_f,
                         const int32_t *row_src_h, void *xf_sp, void *xh_sp, const float *am_img, const float *am_feat,
                         float *xf_sinv, float *xh_sinv, int rows, int cf, int nvec, void *stream) {
  MVG_REQUIRE(img_feat && feat && row_img && am_img && am_feat && rows > 0, "fuse_build_split: null argument");
  MVG_REQUIRE(cf % 8 == 0 && nvec % 8 == 0, "fuse_build_split: cf and nvec must be multiples of 8");
  MVG_REQUIRE((!xf_sp || (row_src_f && xf_sinv)) && (!xh_sp || (row_src_h && xh_sinv)) && (xf_sp || xh_sp),
              "fuse_build_split: each operand needs its row table and its sinv slot");
  hipStream_t st = (hipStream_t)stream;
  const int nout = (xf_sp ? 1 : 0) + (xh_sp ? 1 : 0);
  ProfScope ps(MVG_K_ROTCAT, st, 18.0 * rows * nvec, (4.0 + 4.0 * nout) * (double)rows * (cf + 3 * nvec));
  hipLaunchKernelGGL(fuse_build_split_kernel, dim3(rows), dim3(256), 0, st, img_feat, feat, rel, row_img, row_src_f, row_src_h, (uint4 *)xf_sp,
                     (uint4 *)xh_sp, (const unsigned *)am_img, (const unsigned *)am_feat, xf_sinv, xh_sinv, cf, nvec);
  return check_launch("fuse_build_split");
}

int mvg_fuse_unbuild(const float *dxh, const float *dxn, const float *rel, const int32_t *seg, const int32_t *vi, float *dfeat, float *da,
                     int da_accumulate, int segments, int views, int dirs, int batch, int cf, int nvec, float *absmax, void *stream) {
  MVG_REQUIRE((dxh || dxn) && seg && vi && dfeat && da, "fuse_unbuild: null argument");
  MVG_REQUIRE(cf % 4 == 0 && segments > 0 && views > 0 && dirs > 0 && batch > 0, "fuse_unbuild: bad sizes");
  MVG_REQUIRE(!dxh || segments == dirs, "fuse_unbuild: the head-input gradient has one row per direction (segments == dirs)");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ROTCAT, st, 18.0 * dirs * batch * nvec, 4.0 * (double)batch * ((dxh ? 1 : 0) + (dxn ? 1 : 0)) * dirs * (cf + 3 * nvec));
  const int rpb = 4;                      // feature rows per workgroup: a quarter of the atomics, still >= 1 workgroup per CU at C3
  hipLaunchKernelGGL(fuse_unbuild_kernel, dim3(ceil_div(segments * batch, rpb) + views * batch), dim3(256), 0, st, dxh, dxn, rel, seg, vi, dfeat,
                     da, da_accumulate, segments, views, dirs, batch, cf, nvec, (unsigned *)absmax, rpb);
  return check_launch("fuse_unbuild");
}

int mvg_gaze_angular_loss_multi(const float *pred, const float *gt, int iters, int dirs, int batch, const float *host_weights, float *loss,
                                float *dpred, void *stream) {
  MVG_REQUIRE(pred && gt && host_weights && loss && iters > 0 && dirs > 0 && batch > 0 && iters * dirs <= 512,
              "gaze_angular_loss_multi: bad arguments (iters * dirs <= 512)");
  LossWeights lw;
  memset(&lw, 0, sizeof(lw));
  for (int i = 0; i < iters * dirs; ++i) lw.w[i] = host_weights[i];
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_LOSS, st, 0.0, 24.0 * iters * dirs * batch);
  hipLaunchKernelGGL(gaze_loss_multi_kernel, dim3(1), dim3(256), 0, st, pred, gt, iters, dirs, batch, lw, loss, dpred);
  return check_launch("gaze_angular_loss_multi");
}

int mvg_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const float *lr_dev, float *state3,
                      float beta1, float beta2, float eps, float weight_decay, void *stream) {
  MVG_REQUIRE(param && grad && exp_avg && exp_avg_sq && lr_dev && state3 && n % 4 == 0, "adam_step_dev: null argument or n %% 4 != 0");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_ELEMENTWISE, st, 0.0, 28.0 * (double)n);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st, state3, beta1, beta2);
  if (check_launch("adam_tick")) return 1;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (float4 *)param, (const float4 *)grad, (float4 *)exp_avg,
                     (float4 *)exp_avg_sq, (long long)(n / 4), lr_dev, state3, beta1, beta2, eps, weight_decay);
  return check_launch("adam_step_dev");
}

int mvg_linear_skinny_fwd(const float *x, const float *w, const float *bias, float *y, int rows, int k, int nout,
                          void *stream) {
  MVG_REQUIRE(nout >= 1 && nout <= 4, "skinny linear: out_features must be 1..4");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_LINEAR_FPROP, st, 2.0 * rows * (double)k * nout, 4.0 * ((double)rows * k + (double)k * nout));
  hipLaunchKernelGGL(skinny_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, x, w, bias, y, rows, k, nout);
  return check_launch("skinny_fwd");
}

int mvg_linear_skinny_bwd(const float *dy, const float *x, const float *w, const float *mask, float *dx, float *dw,
                          float *db, int rows, int k, int nout, int accumulate, float *dx_absmax, void *stream) {
  MVG_REQUIRE(nout >= 1 && nout <= 4, "skinny linear: out_features must be 1..4");
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    ProfScope ps(MVG_K_LINEAR_DGRAD, st, 2.0 * rows * (double)k * nout, 8.0 * (double)rows * k);
    const long long total = (long long)rows * k;
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(skinny_bwd_dx_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, w, mask, dx, total, k, nout, (unsigned *)dx_absmax);
    if (check_launch("skinny_bwd_dx")) return 1;
  }
  if (dw) {
    ProfScope ps(MVG_K_LINEAR_WGRAD, st, 2.0 * rows * (double)k * nout, 4.0 * (double)rows * k);
    hipLaunchKernelGGL(skinny_bwd_dw_kernel, dim3(ceil_div(k, 4)), dim3(256), 0, st, dy, x, dw, db, rows, k, nout,
                       accumulate);
    if (check_launch("skinny_bwd_dw")) return 1;
  }
  return 0;
}

int mvg_gaze_lp_loss(const float *pred, const float *label, int n, int p, float *loss, float *dpred, void *stream) {
  MVG_REQUIRE(n > 0 && (p == 1 || p == 2), "gaze_lp_loss: n > 0 and p in {1, 2}");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_LOSS, st, 0.0, 12.0 * n);
  hipLaunchKernelGGL(gaze_lp_loss_kernel, dim3(1), dim3(256), 0, st, pred, label, n, p, loss, dpred);
  return check_launch("gaze_lp_loss");
}

int mvg_gaze_angular_loss(const float *pred, const float *gt, int n, float row_weight, float *loss, int accumulate,
                          float *dpred, float *theta_out, void *stream) {
  MVG_REQUIRE(n > 0, "gaze loss: n <= 0");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_LOSS, st, 0.0, 24.0 * n);
  hipLaunchKernelGGL(gaze_loss_kernel, dim3(1), dim3(256), 0, st, pred, gt, n, row_weight, loss, accumulate, dpred,
                     theta_out);
  return check_launch("gaze_angular_loss");
}

}  // extern "C"
