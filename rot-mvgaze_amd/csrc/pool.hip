// MaxPool2d(3,2,1), global average pool, and the NCHW <-> NHWC4 boundary repack.  HBM-bound.
#include "common.h"
#include "elem.h"

namespace mvg {

// one thread = one (n, ho, wo, 4 channels).  argmax = index (0..8) of the FIRST maximum in
// (kh, kw) scan order over the in-bounds window (ATen's max_pool2d tie rule: strict >).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float4 *__restrict__ x, float4 *__restrict__ y,
                                                          uchar4 *__restrict__ argmax, long long total, int h, int w,
                                                          int c4n, int ho, int wo) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % c4n);
  long long t = i / c4n;
  const int ox = (int)(t % wo);
  t /= wo;
  const int oy = (int)(t % ho);
  const long long n = t / ho;
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
      const float4 v = x[((n * h + iy) * w + ix) * c4n + cq];
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
  y[i] = best;
  argmax[i] = idx;
}

// gather form (no atomics): an input pixel collects the gradient of every window it won.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float4 *__restrict__ dy,
                                                          const uchar4 *__restrict__ argmax, float4 *__restrict__ dx,
                                                          long long total, int h, int w, int c4n, int ho, int wo) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % c4n);
  long long t = i / c4n;
  const int ix = (int)(t % w);
  t /= w;
  const int iy = (int)(t % h);
  const long long n = t / h;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // windows (oy, ox) with oy*2-1 <= iy <= oy*2+1
  const int oy0 = iy >> 1, ox0 = ix >> 1;   // candidates: floor(iy/2) and floor((iy+1)/2)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int oy = oy0 + a;
    const int kh = iy - (oy * 2 - 1);
    if (kh < 0 || kh > 2 || oy >= ho) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ox = ox0 + b;
      const int kw = ix - (ox * 2 - 1);
      if (kw < 0 || kw > 2 || ox >= wo) continue;
      const long long o = ((n * ho + oy) * wo + ox) * c4n + cq;
      const uchar4 am = argmax[o];
      const float4 g = dy[o];
      const unsigned char k = (unsigned char)(kh * 3 + kw);
      if (am.x == k) acc.x += g.x;
      if (am.y == k) acc.y += g.y;
      if (am.z == k) acc.z += g.z;
      if (am.w == k) acc.w += g.w;
    }
  }
  dx[i] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T *__restrict__ x, float4 *__restrict__ y, int n,
                                                          int hw, int c4n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * c4n) return;
  const int cq = (int)(i % c4n);
  const long long img = i / c4n;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = 0; p < hw; ++p) {
    const float4 v = Elem<T>::ld4(x, (img * hw + p) * c4n + cq);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float inv = 1.f / (float)hw;
  y[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float4 *__restrict__ dy, T *__restrict__ dx,
                                                          long long total, int hw, int c4n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % c4n);
  const long long img = i / ((long long)hw * c4n);
  const float inv = 1.f / (float)hw;
  const float4 g = dy[img * c4n + cq];
  Elem<T>::st4(dx, i, make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv));
}

// fp32 NCHW images -> bf16 NHWC with the channels zero-padded to 8 (one 16-byte vector per pixel): the
// bf16 stem reads 8-channel pixels (K = 7*7*8)
__global__ __launch_bounds__(256) void nchw_to_nhwc8_bf16_kernel(const float *__restrict__ src, uint4 *__restrict__ dst,
                                                                 long long pixels, int c, long long hw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels) return;
  const long long n = i / hw, p = i - n * hw;
  const float *s = src + n * c * hw + p;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < c; ++k) v[k] = s[k * hw];
  dst[i] = make_uint4(bf16_pack2(v[0], v[1]), bf16_pack2(v[2], v[3]), bf16_pack2(v[4], v[5]), bf16_pack2(v[6], v[7]));
}

__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float *__restrict__ src, float4 *__restrict__ dst,
                                                            long long pixels, int c, long long hw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels) return;
  const long long n = i / hw, p = i - n * hw;
  const float *s = src + n * c * hw + p;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < c; ++k) v[k] = s[k * hw];
  dst[i] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void nhwc4_to_nchw_kernel(const float4 *__restrict__ src, float *__restrict__ dst,
                                                            long long pixels, int c, long long hw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels) return;
  const long long n = i / hw, p = i - n * hw;
  const float4 v4 = src[i];
  const float v[4] = {v4.x, v4.y, v4.z, v4.w};
  float *d = dst + n * c * hw + p;
  for (int k = 0; k < c; ++k) d[k * hw] = v[k];
}

// uint8 HWC face patch -> normalised fp32 NHWC4: the deterministic part of the reference's input
// pipeline (dataset/gaze.py:106-111 BGR->RGB; main.py:50-55 ToTensor = /255, Normalize(mean, std)).
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char *__restrict__ src,
                                                            float4 *__restrict__ dst, long long pixels, float m0,
                                                            float m1, float m2, float s0, float s1, float s2,
                                                            int swap_rb) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels) return;
  const unsigned char *p = src + 3 * i;
  float c0 = (float)p[swap_rb ? 2 : 0], c1 = (float)p[1], c2 = (float)p[swap_rb ? 0 : 2];
  c0 = (c0 / 255.0f - m0) / s0;
  c1 = (c1 / 255.0f - m1) / s1;
  c2 = (c2 / 255.0f - m2) / s2;
  dst[i] = make_float4(c0, c1, c2, 0.f);
}

// test_transform of main.py:50-55 for patches that are not already S x S: ToTensor (/255) ->
// Resize((S, S), antialias=True) -> Normalize, fused, NHWC4 out.  torchvision's tensor Resize is
// torch.nn.functional.interpolate(mode="bilinear", antialias=True, align_corners=False) = ATen
// _upsample_bilinear2d_aa: per axis a triangle filter of half-width max(scale, 1) input pixels around
// scale * (i + 0.5), weights normalised by their float32 sum (UpSampleKernel.cpp,
// _compute_indices_min_size_weights_aa); width pass first, then height.  One thread per output pixel;
// the weights are recomputed per thread (<= (2*ceil(scale)+1)^2 filter evaluations).
struct AaAxis {
  float scale, support, invscale;
  int in_size, max_k;
};
__device__ __forceinline__ void aa_window(const AaAxis &a, int i, float &center, int &x0, int &xs, float &total) {
  // The product must be rounded before it is used: the window bounds are truncations of centre -+ support
  // + 0.5 and ATen's host code rounds each step.  hipcc contracts a*b+c into an fma by default (its
  // __fmul_rn / __fadd_rn are plain operators and `#pragma clang fp contract(off)` did not survive the
  // inlining here): fma(scale, i + 0.5, -support) moved the first tap of every third column at scale 5/3.
  // The empty asm makes the rounded product an opaque value.
  center = a.scale * ((float)i + 0.5f);
  asm volatile("" : "+v"(center));
  x0 = (int)__fadd_rn(__fsub_rn(center, a.support), 0.5f);
  if (x0 < 0) x0 = 0;
  int hi = (int)__fadd_rn(__fadd_rn(center, a.support), 0.5f);
  if (hi > a.in_size) hi = a.in_size;
  xs = hi - x0;
  xs = xs < 0 ? 0 : (xs > a.max_k ? a.max_k : xs);
  total = 0.f;
  for (int j = 0; j < xs; ++j) {
    const float t = __fmul_rn(__fadd_rn(__fsub_rn((float)(j + x0), center), 0.5f), a.invscale);
    total = __fadd_rn(total, fmaxf(0.f, 1.f - fabsf(t)));
  }
}
__device__ __forceinline__ float aa_weight(const AaAxis &a, int j, int x0, float center, float total) {
  const float t = __fmul_rn(__fadd_rn(__fsub_rn((float)(j + x0), center), 0.5f), a.invscale);
  const float w = fmaxf(0.f, 1.f - fabsf(t));
  return total != 0.f ? w / total : w;
}
__global__ __launch_bounds__(256) void preprocess_u8_resize_kernel(const unsigned char *__restrict__ src,
                                                                   float4 *__restrict__ dst, int n, int h, int w, int oh,
                                                                   int ow, AaAxis ay, AaAxis ax, float m0, float m1,
                                                                   float m2, float s0, float s1, float s2, int swap_rb) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * oh * ow) return;
  const int ox = (int)(i % ow);
  const long long r = i / ow;
  const int oy = (int)(r % oh), img = (int)(r / oh);
  float cx, cy, tx, ty;
  int x0, xs, y0, ys;
  aa_window(ax, ox, cx, x0, xs, tx);
  aa_window(ay, oy, cy, y0, ys, ty);
  const unsigned char *base = src + (long long)img * h * w * 3;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  for (int jy = 0; jy < ys; ++jy) {
    const unsigned char *row = base + ((long long)(y0 + jy) * w + x0) * 3;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int jx = 0; jx < xs; ++jx) {
      const float wx = aa_weight(ax, jx, x0, cx, tx);
      t0 = __fadd_rn(t0, __fmul_rn(wx, (float)row[3 * jx] / 255.0f));
      t1 = __fadd_rn(t1, __fmul_rn(wx, (float)row[3 * jx + 1] / 255.0f));
      t2 = __fadd_rn(t2, __fmul_rn(wx, (float)row[3 * jx + 2] / 255.0f));
    }
    const float wy = aa_weight(ay, jy, y0, cy, ty);
    o0 = __fadd_rn(o0, __fmul_rn(wy, t0));
    o1 = __fadd_rn(o1, __fmul_rn(wy, t1));
    o2 = __fadd_rn(o2, __fmul_rn(wy, t2));
  }
  if (swap_rb) {
    const float t = o0;
    o0 = o2;
    o2 = t;
  }
  dst[i] = make_float4((o0 - m0) / s0, (o1 - m1) / s1, (o2 - m2) / s2, 0.f);
}

// RandomMultiErasing (utils/augment.py:10-47): img *= nearest-neighbour upsampling of a per-image
// g x g keep-mask (F.interpolate default mode: src = min(int(floorf(dst * (float)g / size)), g - 1)).
// grid[n] == 0: this image is not erased.  NCHW in place.
__global__ __launch_bounds__(256) void multi_erase_kernel(float *__restrict__ img, const float *__restrict__ masks,
                                                          const int *__restrict__ grid, int gmax, int c, int h, int w) {
  const int n = blockIdx.y;
  const int g = grid[n];
  if (g <= 0) return;
  const long long hw = (long long)h * w;
  const float sy = (float)g / (float)h, sx = (float)g / (float)w;
  const float *m = masks + (long long)n * gmax * gmax;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / w), x = (int)(i - (long long)y * w);
    int my = (int)floorf((float)y * sy), mx = (int)floorf((float)x * sx);
    my = my < g - 1 ? my : g - 1;
    mx = mx < g - 1 ? mx : g - 1;
    const float keep = m[my * g + mx];
    for (int ch = 0; ch < c; ++ch) img[((long long)n * c + ch) * hw + i] *= keep;
  }
}

}  // namespace mvg

using namespace mvg;

extern "C" {

int mvg_maxpool3x3s2_fwd(const float *x, float *y, uint8_t *argmax, int n, int h, int w, int c, int ho, int wo,
                         void *stream) {
  MVG_REQUIRE(c % 4 == 0, "maxpool: c %% 4 != 0");
  MVG_REQUIRE(ho == (h + 2 - 3) / 2 + 1 && wo == (w + 2 - 3) / 2 + 1, "maxpool: bad output size");
  const long long total = (long long)n * ho * wo * (c / 4);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_POOL, st, 0.0, 4.0 * ((double)n * h * w * c + (double)total * 5));
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const float4 *)x, (float4 *)y,
                     (uchar4 *)argmax, total, h, w, c / 4, ho, wo);
  return check_launch("maxpool_fwd");
}

int mvg_maxpool3x3s2_bwd(const float *dy, const uint8_t *argmax, float *dx, int n, int h, int w, int c, int ho, int wo,
                         void *stream) {
  MVG_REQUIRE(c % 4 == 0, "maxpool: c %% 4 != 0");
  const long long total = (long long)n * h * w * (c / 4);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_POOL, st, 0.0, 4.0 * ((double)n * h * w * c + (double)n * ho * wo * c * 1.25));
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const float4 *)dy,
                     (const uchar4 *)argmax, (float4 *)dx, total, h, w, c / 4, ho, wo);
  return check_launch("maxpool_bwd");
}

#define MVG_AVGPOOL_FACES(SUFFIX, T)                                                                                  \
  int mvg_avgpool_fwd##SUFFIX(const T *x, float *y, int n, int hw, int c, void *stream) {                             \
    MVG_REQUIRE(c % 4 == 0, "avgpool: c %% 4 != 0");                                                                   \
    hipStream_t st = (hipStream_t)stream;                                                                             \
    ProfScope ps(MVG_K_POOL, st, 0.0, (double)n * (Elem<T>::kBytes * hw + 4.0) * c);                                  \
    hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(ceil_div((long long)n * (c / 4), 256)), dim3(256), 0, st, x,       \
                       (float4 *)y, n, hw, c / 4);                                                                    \
    return check_launch("avgpool_fwd");                                                                               \
  }                                                                                                                    \
  int mvg_avgpool_bwd##SUFFIX(const float *dy, T *dx, int n, int hw, int c, void *stream) {                           \
    MVG_REQUIRE(c % 4 == 0, "avgpool: c %% 4 != 0");                                                                   \
    hipStream_t st = (hipStream_t)stream;                                                                             \
    const long long total = (long long)n * hw * (c / 4);                                                              \
    ProfScope ps(MVG_K_POOL, st, 0.0, (double)n * (Elem<T>::kBytes * hw + 4.0) * c);                                  \
    hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const float4 *)dy, dx,    \
                       total, hw, c / 4);                                                                             \
    return check_launch("avgpool_bwd");                                                                               \
  }
MVG_AVGPOOL_FACES(, float)
MVG_AVGPOOL_FACES(_bf16, uint16_t)
#undef MVG_AVGPOOL_FACES

int mvg_avgpool_fwd_split(const void *x_sp, float *y, int n, int hw, int c, void *stream) {
  MVG_REQUIRE(c % 8 == 0, "avgpool_split: c %% 8 != 0");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_POOL, st, 0.0, (double)n * (6.0 * hw + 4.0) * c);
  hipLaunchKernelGGL(avgpool_fwd_kernel<sp_t>, dim3(ceil_div((long long)n * (c / 4), 256)), dim3(256), 0, st, (const sp_t *)x_sp,
                     (float4 *)y, n, hw, c / 4);
  return check_launch("avgpool_fwd_split");
}

int mvg_nchw_to_nhwc8_bf16(const float *src, uint16_t *dst, int n, int c, int h, int w, void *stream) {
  MVG_REQUIRE(c >= 1 && c <= 8, "nchw_to_nhwc8_bf16: c must be 1..8");
  hipStream_t st = (hipStream_t)stream;
  const long long pixels = (long long)n * h * w;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, (double)pixels * (4.0 * c + 16.0));
  hipLaunchKernelGGL(nchw_to_nhwc8_bf16_kernel, dim3(ceil_div(pixels, 256)), dim3(256), 0, st, src, (uint4 *)dst, pixels, c,
                     (long long)h * w);
  return check_launch("nchw_to_nhwc8_bf16");
}

int mvg_nchw_to_nhwc4(const float *src, float *dst, int n, int c, int h, int w, void *stream) {
  MVG_REQUIRE(c >= 1 && c <= 4, "nchw_to_nhwc4: c must be 1..4");
  hipStream_t st = (hipStream_t)stream;
  const long long pixels = (long long)n * h * w;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 4.0 * (double)pixels * (c + 4));
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(ceil_div(pixels, 256)), dim3(256), 0, st, src, (float4 *)dst, pixels, c,
                     (long long)h * w);
  return check_launch("nchw_to_nhwc4");
}

int mvg_preprocess_u8hwc(const uint8_t *src, float *dst, int n, int h, int w, float mean0, float mean1, float mean2,
                         float std0, float std1, float std2, int swap_rb, void *stream) {
  MVG_REQUIRE(std0 > 0.f && std1 > 0.f && std2 > 0.f, "preprocess: std must be positive");
  hipStream_t st = (hipStream_t)stream;
  const long long pixels = (long long)n * h * w;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 19.0 * (double)pixels);
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(ceil_div(pixels, 256)), dim3(256), 0, st, src, (float4 *)dst, pixels, mean0,
                     mean1, mean2, std0, std1, std2, swap_rb);
  return check_launch("preprocess_u8hwc");
}

static AaAxis aa_axis(int in_size, int out_size) {
  AaAxis a;
  a.scale = (float)in_size / (float)out_size;
  a.support = a.scale >= 1.f ? a.scale : 1.f;
  a.invscale = a.scale >= 1.f ? 1.f / a.scale : 1.f;
  a.in_size = in_size;
  a.max_k = (int)ceilf(a.support) * 2 + 1;
  return a;
}

int mvg_preprocess_u8hwc_resize(const uint8_t *src, float *dst, int n, int h, int w, int oh, int ow, float mean0,
                                float mean1, float mean2, float std0, float std1, float std2, int swap_rb, void *stream) {
  MVG_REQUIRE(std0 > 0.f && std1 > 0.f && std2 > 0.f, "preprocess: std must be positive");
  MVG_REQUIRE(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "preprocess_resize: bad sizes");
  if (h == oh && w == ow)      // torchvision's resize returns the input unchanged
    return mvg_preprocess_u8hwc(src, dst, n, h, w, mean0, mean1, mean2, std0, std1, std2, swap_rb, stream);
  hipStream_t st = (hipStream_t)stream;
  const long long pixels = (long long)n * oh * ow;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 3.0 * (double)n * h * w + 16.0 * (double)pixels);
  hipLaunchKernelGGL(preprocess_u8_resize_kernel, dim3(ceil_div(pixels, 256)), dim3(256), 0, st, src, (float4 *)dst, n, h, w,
                     oh, ow, aa_axis(h, oh), aa_axis(w, ow), mean0, mean1, mean2, std0, std1, std2, swap_rb);
  return check_launch("preprocess_u8hwc_resize");
}

int mvg_multi_erase_nchw(float *img, const float *masks, const int32_t *grid, int gmax, int n, int c, int h, int w,
                         void *stream) {
  MVG_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && gmax > 0, "multi_erase: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 8.0 * (double)n * c * h * w);
  long long bx = ((long long)h * w + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(multi_erase_kernel, dim3((unsigned)bx, n), dim3(256), 0, st, img, masks, grid, gmax, c, h, w);
  return check_launch("multi_erase");
}

int mvg_nhwc4_to_nchw(const float *src, float *dst, int n, int c, int h, int w, void *stream) {
  MVG_REQUIRE(c >= 1 && c <= 4, "nhwc4_to_nchw: c must be 1..4");
  hipStream_t st = (hipStream_t)stream;
  const long long pixels = (long long)n * h * w;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 4.0 * (double)pixels * (c + 4));
  hipLaunchKernelGGL(nhwc4_to_nchw_kernel, dim3(ceil_div(pixels, 256)), dim3(256), 0, st, (const float4 *)src, dst, pixels,
                     c, (long long)h * w);
  return check_launch("nhwc4_to_nchw");
}

}  // extern "C"
