// Types and helpers shared by the implicit-GEMM translation units (conv_igemm.hip: fp32 MFMA,
// conv_bf16.hip: bf16 MFMA).  Everything here is static / inline: the library is built without
// relocatable device code, so a kernel must be defined in the translation unit that launches it.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace mvg {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte load through a buffer descriptor: an offset >= num_records returns zeros in hardware, so
// predication is a v_cndmask on the offset - no branch, no select on the data.
// pred_off(off, ok): sets bit 31 when !ok -> beyond any descriptor here (tensors/groups < 2 GiB);
// written arithmetically so that `off` is computed unconditionally (a select with an "expensive"
// arm is turned back into a branch by the compiler, splitting the K-step's basic block).
__device__ __forceinline__ unsigned pred_off(unsigned off, bool ok) { return off | ((unsigned)(!ok) << 31); }
__device__ __forceinline__ float4 buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// base / bytes must be wave-uniform; readfirstlane makes that provable to the compiler (otherwise it
// wraps every buffer op in a waterfall loop - cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, long long bytes) {
  unsigned n = bytes > 0x7FFFFFF0ll ? 0x7FFFFFF0u : (bytes < 0 ? 0u : (unsigned)bytes);
  const unsigned long long b = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  n = __builtin_amdgcn_readfirstlane(n);
  void *ub = (void *)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)n, 0x00020000);
}

// exact unsigned 32-bit division by a runtime constant (Granlund-Montgomery round-up form)
struct FastDiv {
  unsigned mul, sh1, sh2, d;
};
static FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.sh1 = l < 1 ? l : 1;
  f.sh2 = l > 0 ? l - 1 : 0;
  f.d = d;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv &f) {
  const unsigned t = __umulhi(f.mul, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// Per-class view of a launch.  fprop and stride-1 dgrad have one class; a stride-2 dgrad has one per
// output-pixel parity (py, px) - each a dense sub-convolution over its own taps - and all of them
// run in ONE launch, so that the short classes (one tap: 8 K-steps per tile) share the device with
// the long ones instead of each paying a partly filled round of workgroups.
struct IgemmClass {
  int out_h, out_w;          // spatial extent of the GEMM's row space
  int ntaps, tap_ns, tap_r0, tap_s0;
  int ktotal;                // ntaps * src_c
  int cls_py, cls_px, cls_cy, cls_cx;
  int mtiles_per_group;
  long long rows_per_group;
  FastDiv tap_ns_div, ohw_div, ow_div;
  int korder;                // 1: K-steps run (32-channel block, tap, half) - see igemm_kernel
  FastDiv per_div;           // K-steps per 32-channel block: ntaps * 32 / BK
  long long unit0;           // first (tile, K-step) unit of this class in the launch's unit space
  int tile0;                 // first tile of this class
  int KT;                    // K-steps per tile (>= 1: a class without taps runs one all-zero step; split / bf16 kernels
                             // with a fused BatchNorm reduce: 0 - such a class's tiles are epilogue only)
  int part0;                 // fused BatchNorm-backward reduce: this class's first partial row inside a group (bn_part)
};

struct IgemmParams {
  const float *a;       // gathered operand (fprop: x, dgrad: dy)
  const float *b;       // weights, KRSC
  float *out;
  const float *bias;    // [ncols] or null (fprop)
  const float *scale;   // [ncols] or null (fprop, inference: y = acc*scale + bias (+addend) [relu])
  const float *mask;    // like out or null (dgrad)
  const float *addend;  // like out or null (dgrad)
  float *stats;         // [groups][P][2][ncols] or null (fprop)
  int relu;
  int groups;
  int out_h, out_w;     // spatial extent of the GEMM's row space
  int src_h, src_w;     // spatial extent of the gathered tensor
  int src_c;            // channels of the gathered tensor (K per tap)
  int src_c_shift;      // log2(src_c) when r*s > 1
  int ncols;            // GEMM N
  int r, s, stride_shift, stride, pad;
  int ktotal;           // r*s*src_c
  int cin;              // dgrad: weight inner dim
  int rs;
  long long rows_per_group;
  long long src_img_stride;  // src_h*src_w*src_c
  int imgs_per_group;
  int mtiles_per_group, ntiles;
  // tap sub-lattice: GEMM k = (ti*tap_ns + tj)*src_c + c with filter tap (r, s) =
  // (tap_r0 + tap_step*ti, tap_s0 + tap_step*tj).  fprop / stride-1 dgrad: the whole filter.
  // Stride-2 dgrad runs once per output-pixel parity class (py, px): only the taps with
  // (y + pad - r) even contribute, so each class is a dense sub-convolution over its own taps
  // (no multiply-by-zero work); rows are the class's pixels (y, x) = (2*y2 + py, 2*x2 + px).
  int ntaps, tap_ns, tap_r0, tap_s0, tap_step;
  int cls_step, cls_py, cls_px, cls_cy, cls_cx;
  int full_h, full_w;        // dgrad: spatial extent of dx (row decode when cls_step == 2)
  long long a_group_bytes;   // bytes of one group of the gathered tensor
  long long b_bytes;         // bytes of the weight tensor
  FastDiv tap_ns_div;
  FastDiv ohw_div, ow_div;   // row -> (image, y, x) of the GEMM's row space (out_h*out_w, out_w)
  // split-K (small-M GEMMs: the Linear layers of the fusion block stream 100-240 MB of weights
  // over <= a few hundred rows; splitting K spreads that stream over every CU).  Partial tiles go
  // to `slab` [splits][groups*rows][ncols] and splitk_reduce_kernel applies the epilogue.
  int splits, ktiles_per_split;
  float *slab;
  // stream-K (fp32 conv kernels): sk_tiles > 0 -> the grid is P persistent workgroups that each take
  // an equal share of the sk_tiles x ceil(ktotal/BK) K-steps, in tile-major order.  A workgroup
  // whose share starts or ends inside a tile writes that piece's raw accumulators to `slab`
  // (slot 0: piece that does not start the tile, slot 1: piece that starts it) and
  // igemm_fixup_kernel sums the pieces in workgroup order and runs the epilogue.
  int sk_tiles;
  int no_remap;              // several classes, one tile per workgroup: keep the dispatch order (longest class first)
  int nt_out;                // split kernels, fp32 result: non-temporal stores
  int ncls;
  IgemmClass cls[4];         // per-class view (one class unless this is a stride-2 dgrad)
  int b_row_len;             // bf16 kernels: elements per row of the (k-contiguous) weight operand
  int stats_partials;        // bf16_epilogue: partials per group of `stats` as the caller allocated it (0: mtiles * wave rows)
  // Backward-data fused with the BatchNorm-backward REDUCE pass of the unit whose output gradient this
  // launch produces (split and bf16 kernels): the epilogue masks the gradient by that unit's ReLU
  // (bn_bits, or fma(bn_y, bn_rscale, bn_rshift) > 0, or no mask), stores the masked gradient and adds up,
  // per workgroup and column, s1 = sum(dz) and s2 = sum(dz * xhat), xhat = (bn_y - bn_mean) * bn_invstd, into
  // bn_part [groups][bn_parts][bn_part_rows][ncols] (bn_parts = row tiles per group over all classes: a stride-2
  // launch's parity classes each own the range starting at IgemmClass::part0; a class without taps - a 1x1 stride-2
  // filter touches one pixel in four - still runs its tiles, epilogue only, for the mask and the sums).
  // bn_y has the storage type of `out` (fp32, or bf16 in the bf16 kernels).
  const float *bn_y, *bn_mean, *bn_invstd, *bn_rscale, *bn_rshift;
  float *bn_part;
  int bn_parts;
  // split kernels, inference forward (BatchNorm folded: y = acc * scale + bias (+ residual) [relu]): the residual
  // and / or the result in sp (the next conv's operand format) instead of fp32
  int addend_sp, out_sp, mask_sp;
  // split kernels: the operands hold (value * 2^k) for a per-tensor k (elem.h: sp_t); the epilogue multiplies the
  // accumulators by *a_sinv * *b_sinv (device scalars, each 2^-k of its operand; null = 1)
  const float *a_sinv, *b_sinv;
  // split kernels running a Linear of the fusion block (igemm_split16_kernel<.., LIN = true>): out_absmax receives max |result| of
  // an fp32 result (atomicMax on the bits; the caller clears it); an sp result (out_sp) is stored times 2^k from the bound
  // ktotal * 2^30 * a_sinv * b_sinv + *bias_absmax and *out_sinv receives 2^-k
  unsigned *out_absmax;
  float *out_sinv;
  const float *bias_absmax;
  int stride_w, pad_w;            // split / bf16 kernels, forward: horizontal stride / padding (= stride / pad except for the stem's
                                  // row-window form: a 7 x 1 filter, vertical stride 2, over windows that already step by 2)
  int stats_fold;                 // bf16 stem (two output columns per window as 2 x cout GEMM columns): the BatchNorm partials of
                                  // columns c and c + ncols/2 are written as partials 2 pi and 2 pi + 1 of channel c
  int bn_part_rows;               // fused BatchNorm-backward reduce: rows per partial in bn_part - 2 (s1, s2) or 3 (+ max |dz| per channel)
  const unsigned char *bn_bits;   // the unit's ReLU mask as bits: one byte per 4 channels (fp32: mvg_bn_apply_split) or per 8 (bf16: mvg_bn_apply_bits_bf16)
  // Cross-view fusion (igemm_kernel AMODE = 1, Linear forward): the A operand is never materialised - row m of
  //   X = [ img_feat[rc_row_img[m]] (rc_cf floats) | rc_rel[m] (3x3) @ feat[rc_row_src[m]] (3 x rc_nvec, axis-major) ]
  // (rot_mv.py:44-50,234-239) is generated by the loader: the image part is a plain row load (p.a = img_feat),
  // the rotated part three loads and three fmas per float4.  rc_rel == null: no rotation (head input, ignore_rotmat).
  const float *rc_feat, *rc_rel;
  const int *rc_row_img, *rc_row_src;
  int rc_cf, rc_nvec, rc_nvec_shift;
  long long rc_img_bytes, rc_feat_bytes;
};

// bijective XCD-aware remap of a 1-D grid (cdna_hip_programming.md §5 "XCD swizzle must be
// bijective"): blocks that are adjacent after the remap share an XCD L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const int base = (xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  return base + (orig >> 3);
}


struct WgradParams {
  const float *x;   // [imgs][h][w][cin]
  const float *dy;  // [imgs][ho][wo][cout]
  float *out;       // [splits][cout][rs*cin] (or dw directly when splits == 1)
  int h, w, cin, cout, r, s, stride, pad, ho, wo;
  int ncols;        // rs*cin
  long long pixels; // imgs*ho*wo
  long long pixels_per_split;
  long long x_bytes;
  int mtiles, ntiles;
  int accumulate;   // only meaningful when splits == 1
  float *db;        // Linear layers: column sums of dy (the bias gradient) ride along, [splits][cout] slabs like `out`
                    // (or the gradient itself when splits == 1); null = not wanted
  // wgrad_kernel XMODE = 1: the x operand is the generated cross-view input X (see IgemmParams::rc_*); x = img_feat
  const float *rc_feat, *rc_rel;
  const int *rc_row_img, *rc_row_src;
  int rc_cf, rc_nvec;
  long long rc_img_bytes, rc_feat_bytes;
  FastDiv ohw_div, wo_div, cin_div, s_div;
  const float *dy_sinv;           // split kernels: 2^-k of the dy operand's per-tensor scale (device scalar, null = 1)
  const float *x_sinv;            // ... and of the x operand's (a Linear's scaled input / hidden activation; null = 1: BatchNorm outputs)
  int stride_w, pad_w;            // split kernels: horizontal stride / padding (see IgemmParams)
};

// Sum of the per-split slabs.  A workgroup covers 256/lanes float4 columns; `lanes` threads per column
// each add every lanes-th slab, then the lanes are summed through LDS in a fixed order
// (deterministic).  With one thread per column (the obvious form) a 64-channel layer has 144
// workgroups each issuing 150 dependent loads: latency-bound at ~1.2 TB/s.
static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ dw,
                                                           long long n4, int splits, int accumulate, int lanes) {
  __shared__ float4 sh[256];
  const int cols = 256 / lanes;
  const int c = threadIdx.x % cols, l = threadIdx.x / cols;
  const long long i = (long long)blockIdx.x * cols + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4)
    for (int k = l; k < splits; k += lanes) {
      const float4 v = reinterpret_cast<const float4 *>(slabs)[(long long)k * n4 + i];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (l == 0 && i < n4) {
    float4 t = accumulate ? reinterpret_cast<const float4 *>(dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < lanes; ++k) {
      const float4 v = sh[k * cols + c];
      t.x += v.x;
      t.y += v.y;
      t.z += v.z;
      t.w += v.w;
    }
    reinterpret_cast<float4 *>(dw)[i] = t;
  }
}

// Algorithmic input channels of a conv for the profiler's FLOP / byte accounting: the 7x7 stem is the
// only conv with cin == 4 and its fourth channel is zero padding (SURVEY 8(d) counts Cin = 3).
static double alg_cin(const mvg_conv_desc *d) { return d->cin == 4 ? 3.0 : (double)d->cin; }

static int ilog2_exact(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

static int validate(const mvg_conv_desc *d) {
  MVG_REQUIRE(d != nullptr, "conv: null descriptor");
  MVG_REQUIRE(d->groups > 0 && d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "conv: bad sizes");
  MVG_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride must be 1 or 2 (got %d)", d->stride);
  MVG_REQUIRE(d->ho == (d->h + 2 * d->pad - d->r) / d->stride + 1 && d->wo == (d->w + 2 * d->pad - d->s) / d->stride + 1,
              "conv: ho/wo inconsistent with h/w/pad/stride");
  MVG_REQUIRE(d->cin % 4 == 0, "conv: cin %% 4 != 0 (%d)", d->cin);
  if (d->r * d->s > 1) {
    MVG_REQUIRE(ilog2_exact(d->cin) >= 0 && ilog2_exact(d->cout) >= 0, "conv: r*s>1 needs power-of-two channels");
  }
  MVG_REQUIRE((long long)d->n * d->ho * d->wo < (1LL << 31) && (long long)d->n * d->h * d->w < (1LL << 31),
              "conv: rows per group overflow int32");
  return 0;
}

// row tiles per group of the backward-data launch = partials per group of the fused reduce: a stride-2 launch has its
// four parity classes' tiles (classes without taps included: their pixels are masked and summed too)
[[maybe_unused]] static int dgrad_bn_partials(const mvg_conv_desc *d, int bm) {
  int parts = 0;
  for (int py = 0; py < d->stride; ++py)
    for (int px = 0; px < d->stride; ++px) {
      const int sub_h = (d->h - py + d->stride - 1) / d->stride, sub_w = (d->w - px + d->stride - 1) / d->stride;
      if (sub_h > 0 && sub_w > 0) parts += ceil_div((long long)d->n * sub_h * sub_w, bm);
    }
  return parts;
}

// the class view of the top-level fields (single-class launches)
static void class_from_params(IgemmClass &c, const IgemmParams &p) {
  memset(&c, 0, sizeof(c));
  c.out_h = p.out_h;
  c.out_w = p.out_w;
  c.ntaps = p.ntaps;
  c.tap_ns = p.tap_ns;
  c.tap_r0 = p.tap_r0;
  c.tap_s0 = p.tap_s0;
  c.ktotal = p.ktotal;
  c.cls_py = p.cls_py;
  c.cls_px = p.cls_px;
  c.cls_cy = p.cls_cy;
  c.cls_cx = p.cls_cx;
  c.mtiles_per_group = p.mtiles_per_group;
  c.rows_per_group = p.rows_per_group;
  c.tap_ns_div = p.tap_ns_div;
  c.ohw_div = p.ohw_div;
  c.ow_div = p.ow_div;
  c.KT = p.ktotal > 0 ? ceil_div(p.ktotal, 16) : 1;
  c.korder = (p.ntaps > 1 && p.src_c % 32 == 0) ? 1 : 0;
  c.per_div = make_fastdiv((unsigned)(p.ntaps > 0 ? 2 * p.ntaps : 1));      // for BK = 16; launch_igemm resets it
}

}  // namespace mvg
