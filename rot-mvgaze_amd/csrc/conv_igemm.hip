// Implicit-GEMM convolution / linear kernels on the gfx950 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, needed for the 1e-4 parity bar).
//
//   fprop : y[m][o]        = sum_{tap,c} x[pix(m,tap)][c]  * w[o][tap][c]      (NT GEMM, gather on A)
//   dgrad : dx[m][c]       = sum_{tap,o} dy[pix'(m,tap)][o] * w[o][tap][c]     (NN GEMM, gather on A)
//   wgrad : dw[o][tap][c]  = sum_m       dy[m][o]           * x[pix(m,tap)][c] (TN GEMM, split over m)
//
// Tensors are NHWC fp32, weights KRSC.  One workgroup = 256 threads = 4 waves (one per SIMD);
// each wave owns a (TM x TN) grid of 32x32 accumulator tiles.  Operands are staged
// global -> registers -> LDS with the next tile's global loads in flight under the current
// tile's MFMAs (one barrier per K-tile), LDS double-buffered.
//
// MFMA operand maps (cdna_hip_programming.md §3): for 32x32x2 lane l supplies A[i=l&31][k=l>>5]
// and B[k=l>>5][j=l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Because both
// operands' k index comes from the same lane half, any bijection between (half, step) and k is
// legal: the k-contiguous LDS images are read with one ds_read_b128 per 4 steps (k = 8g+4h+t).
#include "conv_shared.h"

namespace mvg {
// Shared epilogue of the implicit-GEMM kernels (C/D layout of the 32x32 MFMA is dtype-independent):
// split-K slab store, or bias/ReLU + BN partial statistics (fprop), or ReLU mask + addend (dgrad).
template <int BM, int BN, int WGM, int WGN, bool DGRAD>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams &p, const IgemmClass &c, f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], int g,
                                               int mtile, int ntile, int split, int wm, int wn, int li, int lh,
                                               int ohw, int *row_lds = nullptr) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  const long long row_base = (long long)mtile * BM + wm * WTM;
  const long long grow0 = (long long)g * c.rows_per_group;
  // Stride-2 parity class: a row of the class is the dx pixel (2*y2 + py, 2*x2 + px).  The tile's BM row
  // offsets are decoded ONCE (one thread per row, two exact divisions) into LDS; the stores then cost a
  // table read and an add instead of ~40 VALU instructions of division and 64-bit index math per element
  // (the one-tap classes have 8 K-steps per tile: the old epilogue was longer than their main loop).
  if (DGRAD && p.cls_step == 2 && p.splits <= 1 && row_lds != nullptr &&
      (long long)p.groups * p.imgs_per_group * p.full_h * p.full_w * p.ncols < (1ll << 31)) {
    __syncthreads();                                   // every wave is done with the operand images
    for (int r = threadIdx.x; r < BM; r += blockDim.x) {
      const long long m = (long long)mtile * BM + r;
      int off = -1;
      if (m < c.rows_per_group) {
        const int rr = (int)m;
        const int img = (int)fdiv((unsigned)rr, c.ohw_div), rem = rr - img * ohw;
        const int y2 = (int)fdiv((unsigned)rem, c.ow_div), x2 = rem - y2 * c.out_w;
        off = (((g * p.imgs_per_group + img) * p.full_h + 2 * y2 + c.cls_py) * p.full_w + 2 * x2 + c.cls_px) * p.ncols;
      }
      row_lds[r] = off;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = ntile * BN + wn * WTN + j * 32 + li;
      const bool cok = col < p.ncols;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int4 ro = *reinterpret_cast<const int4 *>(row_lds + wm * WTM + i * 32 + 8 * e4 + 4 * lh);
          const int rov[4] = {ro.x, ro.y, ro.z, ro.w};
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            if (cok && rov[d] >= 0) {
              const int off = rov[d] + col;
              float v = acc[i][j][4 * e4 + d];
              if (p.mask) v = (p.mask[off] > 0.f) ? v : 0.f;
              if (p.addend) v += p.addend[off];
              p.out[off] = v;
            }
          }
        }
      }
    }
    return;
  }
  if (p.splits > 1) {       // raw partial tile -> slab; the epilogue runs in splitk_reduce_kernel
    float *slab = p.slab + (long long)split * p.groups * c.rows_per_group * p.ncols;
    if (row_base + WTM <= c.rows_per_group && ntile * BN + wn * WTN + WTN <= p.ncols &&
        c.rows_per_group * (long long)p.ncols < (1ll << 30)) {           // interior block: no predicates
      float *slab_t = slab + (grow0 + row_base) * p.ncols;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned lane_off = (unsigned)(4 * lh) * (unsigned)p.ncols + (unsigned)(ntile * BN + wn * WTN + j * 32 + li);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            slab_t[(unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * (unsigned)p.ncols + lane_off] = acc[i][j][e];
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = ntile * BN + wn * WTN + j * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long row = row_base + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (col < p.ncols && row < c.rows_per_group) slab[(grow0 + row) * p.ncols + col] = acc[i][j][e];
        }
    }
    return;
  }
  // Fast path (wave-uniform test): the wave's whole WTM x WTN block lies inside the tensor and the rows
  // map linearly to memory - no per-element predicates or 64-bit index arithmetic: a uniform base per
  // (i, e) row plus one 32-bit lane offset.  The general path below costs ~30 VALU instructions per
  // stored element, which displaced a fifth of the MFMA time of the short-K (64-channel) layers.
  if (!(DGRAD && p.cls_step == 2) && row_base + WTM <= c.rows_per_group && ntile * BN + wn * WTN + WTN <= p.ncols &&
      c.rows_per_group * (long long)p.ncols < (1ll << 30)) {
    const long long tile_off = (grow0 + row_base) * p.ncols;             // uniform
    float *out_t = p.out + tile_off;
    const float *add_t = p.addend ? p.addend + tile_off : nullptr;
    const float *mask_t = p.mask ? p.mask + tile_off : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = ntile * BN + wn * WTN + j * 32 + li;
      const unsigned lane_off = (unsigned)(4 * lh) * (unsigned)p.ncols + (unsigned)col;
      float bias = 0.f, scl = 1.f;
      if (!DGRAD && p.bias) bias = p.bias[col];
      if (!DGRAD && p.scale) scl = p.scale[col];
      float csum = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned off = (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * (unsigned)p.ncols + lane_off;
          float v = acc[i][j][e];
          if (DGRAD) {
            if (mask_t) v = (mask_t[off] > 0.f) ? v : 0.f;
            if (add_t) v += add_t[off];
          } else {
            v = v * scl + bias;
            if (add_t) v += add_t[off];
            if (p.relu) v = fmaxf(v, 0.f);
            csum += v;
          }
          out_t[off] = v;
        }
      }
      if (!DGRAD && p.stats) {
        csum += __shfl_xor(csum, 32, 64);
        const float mean = csum / (float)WTM;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float dlt = acc[i][j][e] - mean;
            q += dlt * dlt;
          }
        q += __shfl_xor(q, 32, 64);
        if (lh == 0) {
          const long long P = (long long)c.mtiles_per_group * WGM;
          const long long pi = (long long)mtile * WGM + wm;
          float *st = p.stats + (((long long)g * P + pi) * 2) * p.ncols;
          st[col] = csum;
          st[p.ncols + col] = q;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = ntile * BN + wn * WTN + j * 32 + li;
    const bool cok = col < p.ncols;
    float bias = 0.f, scl = 1.f;
    if (!DGRAD && p.bias && cok) bias = p.bias[col];
    if (!DGRAD && p.scale && cok) scl = p.scale[col];        // inference: BatchNorm folded into the conv
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = row_base + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const bool ok = cok && row < c.rows_per_group;
        float v = acc[i][j][e];
        if (ok) {
          long long off;
          if (DGRAD && p.cls_step == 2) {
            const int rr = (int)row;
            const int img = (int)fdiv((unsigned)rr, c.ohw_div), rem = rr - img * ohw;
            const int y2 = (int)fdiv((unsigned)rem, c.ow_div), x2 = rem - y2 * c.out_w;
            off = ((((long long)g * p.imgs_per_group + img) * p.full_h + 2 * y2 + c.cls_py) * p.full_w + 2 * x2 +
                   c.cls_px) * p.ncols + col;
          } else {
            off = (grow0 + row) * p.ncols + col;
          }
          if (DGRAD) {
            if (p.mask) v = (p.mask[off] > 0.f) ? v : 0.f;
            if (p.addend) v += p.addend[off];
          } else {
            v = v * scl + bias;
            if (p.addend) v += p.addend[off];                  // inference: residual branch
            if (p.relu) v = fmaxf(v, 0.f);
            csum += v;
          }
          p.out[off] = v;
        }
      }
    }
    if (!DGRAD && p.stats) {
      // per-wave partial: column sum and sum of squares centred on the partial's own mean
      long long cnt_ll = c.rows_per_group - row_base;
      const int cnt = cnt_ll <= 0 ? 0 : (cnt_ll > WTM ? WTM : (int)cnt_ll);
      csum += __shfl_xor(csum, 32, 64);
      const float mean = cnt > 0 ? csum / (float)cnt : 0.f;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long row = row_base + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row < c.rows_per_group) {
            const float dlt = acc[i][j][e] - mean;
            q += dlt * dlt;
          }
        }
      q += __shfl_xor(q, 32, 64);
      if (lh == 0 && cok) {
        const long long P = (long long)c.mtiles_per_group * WGM;
        const long long pi = (long long)mtile * WGM + wm;
        float *st = p.stats + (((long long)g * P + pi) * 2) * p.ncols;
        st[col] = csum;
        st[p.ncols + col] = q;
      }
    }
  }
}

// resident workgroups per CU the register budget is held to (LDS allows 4 / 5 / 8 / 5)
constexpr int igemm_min_blocks(int bm, int bn, int bk, bool dgrad) { return (bm == 128 && bn == 128 && bk == 16 && dgrad) ? 3 : 2; }

// FASTA (host: every class has >= 16 channels per tap and <= 32 taps): the 16 k of a K-step lie inside
// ONE tap, so the tap decode, the tap's pixel displacement and the channel base are wave-uniform
// (scalar unit) and each row's bounds checks collapse to one bit of a per-segment tap mask.  PMC on the
// 64-column kernel: VALU active 22 % + MFMA busy 71 % of the cycles - the vector ALU work of the loader
// does not hide under the MFMAs, it displaces them.
// AMODE = 1 (Linear forward with FASTA): the cross-view fusion's rotate + concat runs inside the A loader
// (IgemmParams::rc_*): X = [img_feat | R @ F] never exists in memory.
template <int BM, int BN, int BK, int WGM, int WGN, bool DGRAD, bool FASTA = false, int AMODE = 0>
__global__ __launch_bounds__(256, igemm_min_blocks(BM, BN, BK, DGRAD)) void igemm_kernel(IgemmParams p) {
  static_assert(AMODE == 0 || (!DGRAD && FASTA), "AMODE 1: forward GEMM with the uniform-tap loader");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KV = BK / 4;                 // float4 per k-contiguous row
  constexpr int RPP = 256 / KV;              // rows per loader pass (k-contiguous images)
  constexpr int A_PASSES = BM / RPP;
  constexpr int LDA = BK + 4;
  // B image: fprop [BN][LDA] (k-contiguous), dgrad [BK][LDB] (n-contiguous)
  constexpr int LDB = BN + 4;
  constexpr int B_PASSES_F = (BN + RPP - 1) / RPP;
  constexpr int NV = BN / 4;
  constexpr int KRPP = 256 / NV;             // k-rows per pass (dgrad B)
  constexpr int B_PASSES_D = (BK + KRPP - 1) / KRPP;
  constexpr int A_ELEMS = BM * LDA;
  constexpr int B_ELEMS = DGRAD ? (B_PASSES_D * KRPP) * LDB : (B_PASSES_F * RPP) * LDA;
  static_assert(BM % RPP == 0, "tile");
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];

  const int nwg = gridDim.x;
  const int wg_all = p.no_remap ? (int)blockIdx.x : xcd_remap(blockIdx.x, nwg);
  // this workgroup's range of (tile, K-step) units; the unit space runs class by class, tile by tile
  long long u0, u1;
  int split = 0;
  if (p.sk_tiles > 0) {
    const long long U = p.cls[p.ncls - 1].unit0 + (long long)(p.sk_tiles - p.cls[p.ncls - 1].tile0) * p.cls[p.ncls - 1].KT;
    u0 = (long long)wg_all * U / nwg;
    u1 = (long long)(wg_all + 1) * U / nwg;
  } else {
    const int tiles_total = nwg / p.splits;
    split = wg_all / tiles_total;
    const int wg0 = wg_all - split * tiles_total;
    int ci = 0;
    for (int i = 1; i < p.ncls; ++i) ci += wg0 >= p.cls[i].tile0;
    const int KTc = p.cls[ci].KT;
    const int kb = split * p.ktiles_per_split;
    const int ke = (kb + p.ktiles_per_split < KTc) ? kb + p.ktiles_per_split : KTc;
    u0 = p.cls[ci].unit0 + (long long)(wg0 - p.cls[ci].tile0) * KTc + kb;
    u1 = u0 + (ke - kb);                   // plan_splitk never makes an empty split
  }
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.b, p.b_bytes);

  while (u0 < u1) {
  // Everything below is derived from `tid` through an opaque copy, so that the compiler treats it
  // as segment-local: hoisting the per-thread loader state out of this loop costs ~50 VGPRs (one
  // resident workgroup per CU less) for a once-per-segment saving.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int a_kv = tid % KV;
  const int a_r0 = tid / KV;
  int ci = 0;
  for (int i = 1; i < p.ncls; ++i) ci += u0 >= p.cls[i].unit0;
  const IgemmClass &c = p.cls[ci];
  const int KT_all = c.KT;
  const int ohw = c.out_h * c.out_w;
  const long long ur = u0 - c.unit0;
  const int wg = (int)(ur / KT_all);              // tile within the class
  const int kt_begin = (int)(ur - (long long)wg * KT_all);
  int KT = kt_begin + (int)(u1 - u0);
  if (KT > KT_all) KT = KT_all;
  u0 += KT - kt_begin;
  const int ntile = wg % p.ntiles;
  const int mt_all = wg / p.ntiles;
  const int g = mt_all / c.mtiles_per_group;
  const int mtile = mt_all - g * c.mtiles_per_group;

  // ---- A loader state -------------------------------------------------------------------
  unsigned a_img[A_PASSES];            // byte offset of the row's image inside this group
  int a_y0[A_PASSES], a_x0[A_PASSES];
  bool a_ok[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const long long m = (long long)mtile * BM + a_r0 + i * RPP;
    a_ok[i] = m < c.rows_per_group;
    const int mm = a_ok[i] ? (int)m : 0;
    const int img = (int)fdiv((unsigned)mm, c.ohw_div);
    const int rem = mm - img * ohw;
    const int oy = (int)fdiv((unsigned)rem, c.ow_div), ox = rem - oy * c.out_w;
    if (DGRAD) {
      a_y0[i] = oy + c.cls_cy;
      a_x0[i] = ox + c.cls_cx;
    } else {
      a_y0[i] = oy * p.stride - p.pad;
      a_x0[i] = ox * p.stride - p.pad;
    }
    a_img[i] = (unsigned)(img * p.src_img_stride * 4);
  }
  // FASTA: per row, byte offset of (image, y0, x0, channel 0) and the mask of in-bounds taps
  unsigned a_base[A_PASSES], a_vmask[A_PASSES];
  unsigned b_base[DGRAD ? B_PASSES_D : B_PASSES_F];
  bool b_ok[DGRAD ? B_PASSES_D : B_PASSES_F];
  if (FASTA) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      a_base[i] = a_img[i] + (unsigned)((a_y0[i] * p.src_w + a_x0[i]) * p.src_c) * 4u + (unsigned)a_kv * 16u;
      unsigned m = 0;
      for (int t = 0; t < c.ntaps; ++t) {
        const int fr = (int)fdiv((unsigned)t, c.tap_ns_div), fs = t - fr * c.tap_ns;
        const int iy = DGRAD ? a_y0[i] - fr : a_y0[i] + fr;
        const int ix = DGRAD ? a_x0[i] - fs : a_x0[i] + fs;
        m |= (unsigned)(((unsigned)iy < (unsigned)p.src_h) & ((unsigned)ix < (unsigned)p.src_w)) << t;
      }
      a_vmask[i] = a_ok[i] ? m : 0u;
    }
    if (!DGRAD) {
#pragma unroll
      for (int i = 0; i < B_PASSES_F; ++i) {
        const int n = ntile * BN + a_r0 + i * RPP;
        b_ok[i] = n < p.ncols;
        b_base[i] = ((unsigned)n * (unsigned)c.ktotal + (unsigned)a_kv * 4u) * 4u;
      }
    } else {
      const int ncol = ntile * BN + (tid % NV) * 4;
#pragma unroll
      for (int i = 0; i < B_PASSES_D; ++i) {
        const int krow = tid / NV + i * KRPP;
        b_ok[i] = ncol < p.ncols;
        b_base[i] = ((unsigned)krow * (unsigned)p.rs * (unsigned)p.cin + (unsigned)ncol) * 4u;
      }
    }
  }
  const __amdgpu_buffer_rsrc_t rs_a = AMODE == 1 ? make_rsrc(p.a, p.rc_img_bytes)
      : make_rsrc(p.a + (long long)g * p.imgs_per_group * p.src_img_stride, p.a_group_bytes);
  // AMODE 1: per loader row, byte offsets of its image-feature row and its source-feature row, and its 3x3
  __amdgpu_buffer_rsrc_t rs_f = rs_a;
  unsigned rc_io[AMODE == 1 ? A_PASSES : 1], rc_fo[AMODE == 1 ? A_PASSES : 1];
  float rc_r[AMODE == 1 ? A_PASSES : 1][9];
  if constexpr (AMODE == 1) {
    rs_f = make_rsrc(p.rc_feat, p.rc_feat_bytes);
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const long long m = (long long)mtile * BM + a_r0 + i * RPP;
      const int mm = a_ok[i] ? (int)m : 0;
      rc_io[i] = (unsigned)p.rc_row_img[mm] * (unsigned)p.rc_cf * 4u + (unsigned)a_kv * 16u;
      rc_fo[i] = (unsigned)p.rc_row_src[mm] * (unsigned)(3 * p.rc_nvec) * 4u + (unsigned)a_kv * 16u;
#pragma unroll
      for (int k = 0; k < 9; ++k) rc_r[i][k] = p.rc_rel ? p.rc_rel[(long long)mm * 9 + k] : ((k & 3) == 0 ? 1.f : 0.f);
    }
  }

  float4 a_reg[A_PASSES];
  float4 b_reg[DGRAD ? B_PASSES_D : B_PASSES_F];

  // Predicated loads are branch-free (buffer loads: an out-of-range offset reads zeros) so that the
  // whole K-step stays ONE basic block and the scheduler can interleave loader VALU / VMEM / LDS
  // traffic with the MFMAs.
  auto load_tiles = [&](int kt) {
    // ---- A: one float4 (4 channels of one tap) per row pass
    // first k of this K-step.  korder: k runs (32-channel block, tap, half) instead of (tap, channel),
    // so that the 9 taps of a 3x3 filter revisit a pixel's 128-byte line within 18 consecutive
    // K-steps; tap-major order comes back to it C/16 K-steps later, by which time the other
    // workgroups of the XCD have pushed it out of the 4 MB L2 (9x the algorithmic reads measured).
    int kstart = kt * BK;
    if (c.korder) {
      const int cblk = (int)fdiv((unsigned)kt, c.per_div), rem = kt - cblk * (c.ntaps * (32 / BK));
      kstart = BK == 32 ? (rem << p.src_c_shift) + cblk * 32 : ((rem >> 1) << p.src_c_shift) + cblk * 32 + (rem & 1) * 16;
    }
    if constexpr (FASTA) {
      const int ks = __builtin_amdgcn_readfirstlane(kstart);
      const int tap_u = c.ntaps > 1 ? (ks >> p.src_c_shift) : 0;
      const int chb = ks - (tap_u << p.src_c_shift);                                 // first channel of the step
      const int fru = (int)fdiv((unsigned)tap_u, c.tap_ns_div), fsu = tap_u - fru * c.tap_ns;
      const int disp = (fru * p.src_w + fsu) * p.src_c;                              // the tap's pixel displacement
      const unsigned sdelta = (unsigned)(((DGRAD ? -disp : disp) + chb) * 4);
      const bool kok_u = (kt < KT) & (ks < c.ktotal);
      if constexpr (AMODE == 1) {
        if (ks < p.rc_cf) {                                   // image-feature part of the row
#pragma unroll
          for (int i = 0; i < A_PASSES; ++i) a_reg[i] = buf_ld16(rs_a, pred_off(rc_io[i] + (unsigned)ks * 4u, kok_u & a_ok[i]));
        } else {                                              // rotated part: axis and column are uniform over the wave
          const int kk = ks - p.rc_cf;
          const int axis = kk >> p.rc_nvec_shift;
          const unsigned n0 = (unsigned)(kk - (axis << p.rc_nvec_shift)) * 4u;
          const unsigned st1 = (unsigned)p.rc_nvec * 4u;
#pragma unroll
          for (int i = 0; i < A_PASSES; ++i) {
            const bool ok = kok_u & a_ok[i];
            const float4 f0 = buf_ld16(rs_f, pred_off(rc_fo[i] + n0, ok));
            const float4 f1 = buf_ld16(rs_f, pred_off(rc_fo[i] + n0 + st1, ok));
            const float4 f2 = buf_ld16(rs_f, pred_off(rc_fo[i] + n0 + 2u * st1, ok));
            const float r0 = axis == 0 ? rc_r[i][0] : (axis == 1 ? rc_r[i][3] : rc_r[i][6]);
            const float r1 = axis == 0 ? rc_r[i][1] : (axis == 1 ? rc_r[i][4] : rc_r[i][7]);
            const float r2 = axis == 0 ? rc_r[i][2] : (axis == 1 ? rc_r[i][5] : rc_r[i][8]);
            a_reg[i] = make_float4(r0 * f0.x + r1 * f1.x + r2 * f2.x, r0 * f0.y + r1 * f1.y + r2 * f2.y,
                                   r0 * f0.z + r1 * f1.z + r2 * f2.z, r0 * f0.w + r1 * f1.w + r2 * f2.w);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
          const bool ok = kok_u & (((a_vmask[i] >> tap_u) & 1u) != 0u);
          a_reg[i] = buf_ld16(rs_a, pred_off(a_base[i] + sdelta, ok));
        }
      }
      if (!DGRAD) {
        const unsigned kb = (unsigned)ks * 4u;
#pragma unroll
        for (int i = 0; i < B_PASSES_F; ++i) b_reg[i] = buf_ld16(rs_b, pred_off(b_base[i] + kb, b_ok[i] & kok_u));
      } else {
        const int bi = fru, bj = fsu;
        const int btap = (c.tap_r0 + p.tap_step * bi) * p.s + c.tap_s0 + p.tap_step * bj;
        const unsigned kb = (unsigned)((chb * p.rs + btap) * p.cin) * 4u;
#pragma unroll
        for (int i = 0; i < B_PASSES_D; ++i) b_reg[i] = buf_ld16(rs_b, pred_off(b_base[i] + kb, b_ok[i] & kok_u));
      }
      return;
    }
    const int k0 = kstart + a_kv * 4;
    int tap = 0, ch = k0;
    if (c.ntaps > 1) {
      tap = k0 >> p.src_c_shift;
      ch = k0 - (tap << p.src_c_shift);
    }
    const int fr = (int)fdiv((unsigned)tap, c.tap_ns_div), fs = tap - fr * c.tap_ns;   // lattice coordinates (ti, tj)
    const bool kok = (kt < KT) & (k0 < c.ktotal);
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int iy = DGRAD ? a_y0[i] - fr : a_y0[i] + fr;
      const int ix = DGRAD ? a_x0[i] - fs : a_x0[i] + fs;
      const bool ok = a_ok[i] & kok & ((unsigned)iy < (unsigned)p.src_h) & ((unsigned)ix < (unsigned)p.src_w);
      a_reg[i] = buf_ld16(rs_a, pred_off(a_img[i] + (unsigned)((iy * p.src_w + ix) * p.src_c + ch) * 4u, ok));
    }
    // ---- B
    if (!DGRAD) {
#pragma unroll
      for (int i = 0; i < B_PASSES_F; ++i) {
        const int n = ntile * BN + a_r0 + i * RPP;
        b_reg[i] = buf_ld16(rs_b, pred_off((unsigned)(n * c.ktotal + k0) * 4u, (n < p.ncols) & kok));
      }
    } else {
      const int nv = tid % NV;
      const int ncol = ntile * BN + nv * 4;
#pragma unroll
      for (int i = 0; i < B_PASSES_D; ++i) {
        const int k = kstart + tid / NV + i * KRPP;
        int bt = 0, o = k;
        if (c.ntaps > 1) {
          bt = k >> p.src_c_shift;
          o = k - (bt << p.src_c_shift);
        }
        const int bi = (int)fdiv((unsigned)bt, c.tap_ns_div), bj = bt - bi * c.tap_ns;
        const int btap = (c.tap_r0 + p.tap_step * bi) * p.s + c.tap_s0 + p.tap_step * bj;
        b_reg[i] = buf_ld16(rs_b, pred_off((unsigned)((o * p.rs + btap) * p.cin + ncol) * 4u, (kt < KT) & (k < c.ktotal) & (ncol < p.ncols)));
      }
    }
  };

  // LDS images are allocated for every loader pass (rows beyond the tile land in slack), so the
  // stores need no predicate either.
  auto store_tiles = [&](int buf) {
    float *As = smem + buf * (A_ELEMS + B_ELEMS);
    float *Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<float4 *>(As + (a_r0 + i * RPP) * LDA + a_kv * 4) = a_reg[i];
    if (!DGRAD) {
#pragma unroll
      for (int i = 0; i < B_PASSES_F; ++i)
        *reinterpret_cast<float4 *>(Bs + (a_r0 + i * RPP) * LDA + a_kv * 4) = b_reg[i];
    } else {
      const int nv = tid % NV;
#pragma unroll
      for (int i = 0; i < B_PASSES_D; ++i)
        *reinterpret_cast<float4 *>(Bs + (tid / NV + i * KRPP) * LDB + nv * 4) = b_reg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // 3-stage pipeline: while tile kt multiplies out of LDS[kt&1], tile kt+1 (fetched during the
  // previous step) is written to LDS[(kt+1)&1] and the global loads of tile kt+2 are issued.  Tiles
  // past the end are predicated off (zeros), so the loop body has no branches.
  load_tiles(kt_begin);
  store_tiles(kt_begin & 1);
  load_tiles(kt_begin + 1);
  __syncthreads();

  for (int kt = kt_begin; kt < KT; ++kt) {
    const int cur = kt & 1;
    store_tiles(cur ^ 1);
    load_tiles(kt + 2);
    const float *As = smem + cur * (A_ELEMS + B_ELEMS);
    const float *Bs = As + A_ELEMS;
    // fragments of k-group kg+1 are fetched from LDS while the MFMAs of k-group kg run
    float4 av[2][TM];
    float4 bv[2][TN];
    auto load_frags = [&](int kg, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[slot][i] = *reinterpret_cast<const float4 *>(As + (wm * WTM + i * 32 + li) * LDA + kg * 8 + lh * 4);
      if (!DGRAD) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[slot][j] = *reinterpret_cast<const float4 *>(Bs + (wn * WTN + j * 32 + li) * LDA + kg * 8 + lh * 4);
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float *bp = Bs + (kg * 8 + lh * 4) * LDB + wn * WTN + j * 32 + li;
          bv[slot][j] = make_float4(bp[0], bp[LDB], bp[2 * LDB], bp[3 * LDB]);
        }
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      if (kg + 1 < BK / 8) load_frags(kg + 1, (kg + 1) & 1);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(((const float *)&av[kg & 1][i])[t],
                                                             ((const float *)&bv[kg & 1][j])[t], acc[i][j], 0, 0, 0);
    }
    // schedule hint: issue the next tile's global loads early in the MFMA stream (one load every
    // two MFMAs) so their latency is covered by the remaining MFMAs of this step.
    {
      constexpr int NLOADS = A_PASSES + (DGRAD ? B_PASSES_D : B_PASSES_F);
#pragma unroll
      for (int l = 0; l < NLOADS; ++l) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
      }
    }
    __syncthreads();
  }

  if (p.sk_tiles > 0 && (kt_begin > 0 || KT < KT_all)) {
    // a piece of a tile shared with neighbouring workgroups: raw accumulators, fragment order
    float4 *dst = reinterpret_cast<float4 *>(p.slab) + ((long long)wg_all * 2 + (kt_begin > 0 ? 0 : 1)) * (BM * BN / 4);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          dst[((i * TN + j) * 4 + q) * 256 + tid] =
              make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
  } else {
    igemm_epilogue<BM, BN, WGM, WGN, DGRAD>(p, c, acc, g, mtile, ntile, split, wm, wn, li, lh, ohw,
                                            reinterpret_cast<int *>(smem));
    if (DGRAD && p.cls_step == 2) __syncthreads();      // the row table lives in the operand buffers of the next segment
  }
  }  // while (u0 < u1)
}

// stream-K fix-up: one workgroup per cut between logical workgroups c and c+1; the first cut inside a
// tile sums that tile's pieces (in workgroup order, so the result is deterministic) and runs the epilogue.
template <int BM, int BN, int BK, int WGM, int WGN, bool DGRAD>
__global__ __launch_bounds__(256) void igemm_fixup_kernel(IgemmParams p, int P) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const long long U = p.cls[p.ncls - 1].unit0 + (long long)(p.sk_tiles - p.cls[p.ncls - 1].tile0) * p.cls[p.ncls - 1].KT;
  const int cut = blockIdx.x;
  const long long ucut = (long long)(cut + 1) * U / P;
  int ci = 0;
  for (int i = 1; i < p.ncls; ++i) ci += ucut >= p.cls[i].unit0;
  const IgemmClass &c = p.cls[ci];
  const long long tile = (ucut - c.unit0) / c.KT;
  const long long t0 = c.unit0 + tile * c.KT, t1 = t0 + c.KT;
  if (ucut == t0) return;                          // the cut falls on a tile boundary
  if ((long long)cut * U / P > t0) return;         // an earlier cut lies inside the same tile and owns it
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (int b = cut; b < P; ++b) {
    const long long ub = (long long)b * U / P;
    if (ub >= t1) break;
    if ((long long)(b + 1) * U / P == ub) continue;       // an empty share stored nothing (P > U; not planned, but safe)
    const float4 *src = reinterpret_cast<const float4 *>(p.slab) + ((long long)b * 2 + (ub > t0 ? 0 : 1)) * (BM * BN / 4);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = src[((i * TN + j) * 4 + q) * 256 + tid];
          acc[i][j][4 * q] += v.x;
          acc[i][j][4 * q + 1] += v.y;
          acc[i][j][4 * q + 2] += v.z;
          acc[i][j][4 * q + 3] += v.w;
        }
  }
  const int wg = (int)tile;
  const int ntile = wg % p.ntiles;
  const int mt_all = wg / p.ntiles;
  const int g = mt_all / c.mtiles_per_group;
  const int mtile = mt_all - g * c.mtiles_per_group;
  __shared__ int row_lds[DGRAD ? BM : 1];
  igemm_epilogue<BM, BN, WGM, WGN, DGRAD>(p, c, acc, g, mtile, ntile, 0, wm, wn, li, lh, c.out_h * c.out_w,
                                          DGRAD ? row_lds : nullptr);
}

// out = epilogue(sum_s slab[s]) : fprop (bias, relu) / dgrad (mask, addend); fixed summation order.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float4 *__restrict__ slab, float4 *out, long long n4,
                                                            int splits, int c4n, const float4 *__restrict__ bias, int relu,
                                                            const float4 *__restrict__ mask, const float4 *addend) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 s = slab[i];
    for (int k = 1; k < splits; ++k) {
      const float4 v = slab[(long long)k * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
      const float4 b = bias[i % c4n];
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (relu) {
      s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
    }
    if (mask) {
      const float4 m = mask[i];
      s.x = m.x > 0.f ? s.x : 0.f; s.y = m.y > 0.f ? s.y : 0.f; s.z = m.z > 0.f ? s.z : 0.f; s.w = m.w > 0.f ? s.w : 0.f;
    }
    if (addend) {
      const float4 a = addend[i];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    out[i] = s;
  }
}

// ------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------

// INCR (host: ho*wo >= 32, every product below fits 24 x 24 bits): the pixel coordinates of a loader
// row advance by exactly BK pixels per K-step, so (image offset, oy, ox) are carried in registers and
// stepped with one multiply-shift division instead of being decoded from the pixel index every time
// (99 VALU instructions per K-step, 26 of them quarter-rate multiplies, against 32 MFMAs - and VALU
// work displaces MFMAs on this machine; rows beyond the split fall outside the dy descriptor = zeros).
// XMODE = 1 (Linear layers of the cross-view fusion, non-incremental loader): the x operand is the generated
// row [img_feat | R @ F] of WgradParams::rc_* - the wgrad twin of igemm_kernel's AMODE 1.
template <int BM, int BN, int BK, int WGM, int WGN, bool INCR = false, int XMODE = 0>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  static_assert(XMODE == 0 || !INCR, "XMODE 1 uses the per-row loader");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int MV = BM / 4, NVB = BN / 4;
  constexpr int A_KRPP = 256 / MV, B_KRPP = 256 / NVB;
  constexpr int A_PASSES = (BK + A_KRPP - 1) / A_KRPP, B_PASSES = (BK + B_KRPP - 1) / B_KRPP;
  constexpr int A_ELEMS = (A_PASSES * A_KRPP) * LDA, B_ELEMS = (B_PASSES * B_KRPP) * LDB;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  // 1-D grid, XCD-aware: workgroups round-robin over the 8 XCDs, so the bijective remap hands every
  // XCD a contiguous run of (split, tile) ids.  All tiles of a split (the same pixel range of dy and
  // x) then run on ONE XCD at about the same time and share its L2 - spread over the XCDs, every
  // tile re-read its operands from memory (16x the algorithmic bytes on the 256-channel layers).
  const int tiles = p.mtiles * p.ntiles;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int split = logical / tiles;
  const int tile = logical - split * tiles;
  const int ntile = tile % p.ntiles, mtile = tile / p.ntiles;
  const long long m_begin = (long long)split * p.pixels_per_split;
  long long m_end = m_begin + p.pixels_per_split;
  if (m_end > p.pixels) m_end = p.pixels;
  const int m_count = m_end > m_begin ? (int)(m_end - m_begin) : 0;       // pixels of this split
  const int ohw = p.ho * p.wo;
  const long long img0 = m_begin / ohw;                                    // scalar, once
  const unsigned rem0 = (unsigned)(m_begin - img0 * ohw);

  // A: dy rows of this split, contiguous along cout.  B: x gathered per (pixel, tap); this thread's
  // column (tap, c) is fixed for the whole K loop.
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.dy + m_begin * p.cout, 4ll * m_count * p.cout);
  const long long x_img_elems = (long long)p.h * p.w * p.cin;
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.x + img0 * x_img_elems, p.x_bytes - 4ll * img0 * x_img_elems);
  const __amdgpu_buffer_rsrc_t rs_xi = XMODE == 1 ? make_rsrc(p.x, p.rc_img_bytes) : rs_b;
  const __amdgpu_buffer_rsrc_t rs_xf = XMODE == 1 ? make_rsrc(p.rc_feat, p.rc_feat_bytes) : rs_b;
  const int a_mv = tid % MV, a_k0 = tid / MV;
  const int a_col = mtile * BM + a_mv * 4;
  const bool a_cok = a_col < p.cout;
  const int b_nv = tid % NVB, b_k0 = tid / NVB;
  const int b_col = ntile * BN + b_nv * 4;
  const bool b_cok = b_col < p.ncols;
  const int b_tap = (int)fdiv((unsigned)(b_cok ? b_col : 0), p.cin_div);
  const int b_c = (b_cok ? b_col : 0) - b_tap * p.cin;
  const int b_fr = (int)fdiv((unsigned)b_tap, p.s_div), b_fs = b_tap - b_fr * p.s;
  const int b_dy = b_fr - p.pad, b_dx = b_fs - p.pad;

  // INCR: a thread's B loads all come from ONE pixel row of the K-step (row = tid / (NVB / B_PASSES)) and
  // differ in the column group, so the pixel state (oy, ox, image offset) is carried and stepped once per
  // K-step; per column group only the tap's displacement and bounds differ.
  constexpr int B_TPR = NVB / B_PASSES;                 // threads per k-row
  static_assert(!INCR || NVB % B_PASSES == 0, "wgrad loader mapping");
  const int i_row = tid / B_TPR, i_nv0 = tid % B_TPR;
  int s_oy = 0, s_ox = 0;
  unsigned s_imgoff = 0, a_off[A_PASSES];
  int i_dy[B_PASSES], i_dx[B_PASSES];
  unsigned i_tconst[B_PASSES];
  bool i_cok[B_PASSES];
  const unsigned row_bytes = (unsigned)(p.stride * p.w * p.cin * 4), col_bytes = (unsigned)(p.stride * p.cin * 4);
  const unsigned img_bytes = (unsigned)(x_img_elems * 4);
  if (INCR) {
    const unsigned pix = rem0 + (unsigned)i_row;
    const unsigned img = fdiv(pix, p.ohw_div);
    const unsigned rem = pix - img * (unsigned)ohw;
    const unsigned oy = fdiv(rem, p.wo_div);
    s_oy = (int)oy;
    s_ox = (int)(rem - oy * (unsigned)p.wo);
    s_imgoff = img * img_bytes;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const int col = ntile * BN + (i_nv0 + j * B_TPR) * 4;
      i_cok[j] = col < p.ncols;
      const int tap = (int)fdiv((unsigned)(i_cok[j] ? col : 0), p.cin_div);
      const int cc = (i_cok[j] ? col : 0) - tap * p.cin;
      const int fr = (int)fdiv((unsigned)tap, p.s_div), fs = tap - fr * p.s;
      i_dy[j] = fr - p.pad;
      i_dx[j] = fs - p.pad;
      i_tconst[j] = (unsigned)(((i_dy[j] * p.w + i_dx[j]) * p.cin + cc) * 4);
    }
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      a_off[i] = a_cok ? (unsigned)((a_k0 + i * A_KRPP) * p.cout + a_col) * 4u : 0x80000000u;
  }
  float4 a_reg[A_PASSES], b_reg[B_PASSES];
  auto load_tiles = [&](int kt) {
    if constexpr (INCR) {
      // called with kt = 0, 1, 2, ... in order: the state is that of K-step kt on entry
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        a_reg[i] = buf_ld16(rs_a, a_off[i]);                 // rows >= m_count are beyond the descriptor: zeros
        a_off[i] += (unsigned)(BK * p.cout * 4);
      }
      const bool mok = (kt * BK + i_row < m_count) & (i_row < BK);
      const int iy0 = s_oy * p.stride, ix0 = s_ox * p.stride;
      const unsigned pixoff = s_imgoff + __umul24((unsigned)s_oy, row_bytes) + __umul24((unsigned)s_ox, col_bytes);
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j) {
        const bool ok = mok & i_cok[j] & ((unsigned)(iy0 + i_dy[j]) < (unsigned)p.h) & ((unsigned)(ix0 + i_dx[j]) < (unsigned)p.w);
        b_reg[j] = buf_ld16(rs_b, pred_off(pixoff + i_tconst[j], ok));
      }
      // advance BK pixels: columns wrap into rows, rows into the next image (at most once: ho*wo >= 2*BK)
      unsigned nx = (unsigned)s_ox + BK;
      unsigned q;
      if (p.wo >= BK) {                                      // uniform: at most one row wrap
        q = nx >= (unsigned)p.wo ? 1u : 0u;
        nx -= q ? (unsigned)p.wo : 0u;
      } else {
        q = fdiv(nx, p.wo_div);
        nx -= q * (unsigned)p.wo;
      }
      s_ox = (int)nx;
      const int noy = s_oy + (int)q;
      const bool wrap = noy >= p.ho;
      s_oy = wrap ? noy - p.ho : noy;
      s_imgoff += wrap ? img_bytes : 0u;
      return;
    }
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int m = kt * BK + a_k0 + i * A_KRPP;
      a_reg[i] = buf_ld16(rs_a, pred_off((unsigned)(m * p.cout + a_col) * 4u, (m < m_count) & a_cok));
    }
    if constexpr (XMODE == 1) {
      const bool is_img = b_col < p.rc_cf;                       // uniform over the workgroup: cf is a multiple of BN
      const int kk = b_col - p.rc_cf;
      const int axis = is_img ? 0 : kk / p.rc_nvec;
      const int n0 = kk - axis * p.rc_nvec;
#pragma unroll
      for (int i = 0; i < B_PASSES; ++i) {
        const int m = kt * BK + b_k0 + i * B_KRPP;
        const bool ok = (m < m_count) & b_cok;
        const long long row = ok ? m_begin + m : 0;
        if (is_img) {
          b_reg[i] = buf_ld16(rs_xi, pred_off(((unsigned)p.rc_row_img[row] * (unsigned)p.rc_cf + (unsigned)b_col) * 4u, ok));
        } else {
          const unsigned fo = ((unsigned)p.rc_row_src[row] * (unsigned)(3 * p.rc_nvec) + (unsigned)n0) * 4u;
          const float4 f0 = buf_ld16(rs_xf, pred_off(fo, ok));
          const float4 f1 = buf_ld16(rs_xf, pred_off(fo + (unsigned)p.rc_nvec * 4u, ok));
          const float4 f2 = buf_ld16(rs_xf, pred_off(fo + (unsigned)p.rc_nvec * 8u, ok));
          float r0 = axis == 0 ? 1.f : 0.f, r1 = axis == 1 ? 1.f : 0.f, r2 = axis == 2 ? 1.f : 0.f;
          if (p.rc_rel) {
            const float *rr = p.rc_rel + row * 9 + axis * 3;
            r0 = rr[0];
            r1 = rr[1];
            r2 = rr[2];
          }
          b_reg[i] = make_float4(r0 * f0.x + r1 * f1.x + r2 * f2.x, r0 * f0.y + r1 * f1.y + r2 * f2.y,
                                 r0 * f0.z + r1 * f1.z + r2 * f2.z, r0 * f0.w + r1 * f1.w + r2 * f2.w);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int m = kt * BK + b_k0 + i * B_KRPP;
      const unsigned pix = rem0 + (unsigned)m;
      const unsigned img = fdiv(pix, p.ohw_div);
      const unsigned rem = pix - img * (unsigned)ohw;
      const unsigned oy = fdiv(rem, p.wo_div);
      const unsigned ox = rem - oy * (unsigned)p.wo;
      const int iy = (int)oy * p.stride + b_dy, ix = (int)ox * p.stride + b_dx;
      const bool ok = (m < m_count) & b_cok & ((unsigned)iy < (unsigned)p.h) & ((unsigned)ix < (unsigned)p.w);
      b_reg[i] = buf_ld16(rs_b, pred_off((unsigned)(((img * p.h + iy) * p.w + ix) * p.cin + b_c) * 4u, ok));
    }
  };
  // Linear layers: the bias gradient is the column sum of dy, and this kernel streams dy anyway: the
  // workgroups of the first column tile add up what they load (each K-step's rows exactly once, in order)
  const bool do_bias = (p.db != nullptr) & (ntile == 0);
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto store_tiles = [&](int buf) {
    float *As = smem + buf * (A_ELEMS + B_ELEMS);
    float *Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<float4 *>(As + (a_k0 + i * A_KRPP) * LDA + a_mv * 4) = a_reg[i];
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        if (a_k0 + i * A_KRPP < BK) {
          bsum.x += a_reg[i].x;
          bsum.y += a_reg[i].y;
          bsum.z += a_reg[i].z;
          bsum.w += a_reg[i].w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      if (INCR)
        *reinterpret_cast<float4 *>(Bs + i_row * LDB + (i_nv0 + i * B_TPR) * 4) = b_reg[i];
      else
        *reinterpret_cast<float4 *>(Bs + (b_k0 + i * B_KRPP) * LDB + b_nv * 4) = b_reg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int KT = (m_count + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  load_tiles(1);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    store_tiles(cur ^ 1);
    load_tiles(kt + 2);
    const float *As = smem + cur * (A_ELEMS + B_ELEMS);
    const float *Bs = As + A_ELEMS;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = As[(ks * 2 + lh) * LDA + wm * WTM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bs[(ks * 2 + lh) * LDB + wn * WTN + j * 32 + li];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    {
      constexpr int NLOADS = A_PASSES + B_PASSES;
#pragma unroll
      for (int l = 0; l < NLOADS; ++l) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __syncthreads();
  }

  if (do_bias) {
    // rows of the loader (a_k0) summed through LDS in a fixed order; the K loop ended with a barrier
    float4 *sh = reinterpret_cast<float4 *>(smem);
    sh[tid] = bsum;
    __syncthreads();
    if (a_k0 == 0 && a_cok) {
      float4 t = sh[a_mv];
      for (int k = 1; k < A_KRPP; ++k) {
        const float4 v = sh[k * MV + a_mv];
        t.x += v.x;
        t.y += v.y;
        t.z += v.z;
        t.w += v.w;
      }
      float4 *dst = reinterpret_cast<float4 *>(p.db + (long long)split * p.cout + a_col);
      if (p.accumulate) {
        const float4 o = *dst;
        t.x += o.x;
        t.y += o.y;
        t.z += o.z;
        t.w += o.w;
      }
      *dst = t;
    }
  }
  float *out = p.out + (long long)split * p.cout * p.ncols;
  // fast path (wave-uniform): the wave's block lies inside dw - no predicates, 32-bit offsets
  if (mtile * BM + wm * WTM + WTM <= p.cout && ntile * BN + wn * WTN + WTN <= p.ncols &&
      (long long)p.cout * p.ncols < (1ll << 30)) {
    float *out_t = out + (long long)(mtile * BM + wm * WTM) * p.ncols;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned lane_off = (unsigned)(i * 32 + 4 * lh) * (unsigned)p.ncols + (unsigned)(ntile * BN + wn * WTN + j * 32 + li);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned off = (unsigned)((e & 3) + 8 * (e >> 2)) * (unsigned)p.ncols + lane_off;
          float v = acc[i][j][e];
          if (p.accumulate) v += out_t[off];
          out_t[off] = v;
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = ntile * BN + wn * WTN + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mtile * BM + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < p.cout && col < p.ncols) {
          const long long off = (long long)row * p.ncols + col;
          float v = acc[i][j][e];
          if (p.accumulate) v += out[off];
          out[off] = v;
        }
      }
    }
}

// A stride-2 parity class without taps (e.g. three of the four classes of a 1x1 stride-2 conv):
// dx = 0 there, plus the addend.
__global__ __launch_bounds__(256) void dgrad_empty_class_kernel(float4 *__restrict__ dx, const float4 *__restrict__ addend,
                                                                long long n, int sub_h, int sub_w, int full_h, int full_w,
                                                                int c4, int py, int px) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int cc = (int)(i % c4);
    long long t = i / c4;
    const int x2 = (int)(t % sub_w);
    t /= sub_w;
    const int y2 = (int)(t % sub_h);
    const long long img = t / sub_h;
    const long long off = ((img * full_h + 2 * y2 + py) * full_w + 2 * x2 + px) * c4 + cc;
    dx[off] = addend ? addend[off] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct TileChoice {
  int bm, bn;
};

// K depth of one pipeline step: 32 for the 128x128 fprop tile (half the barriers per MFMA, twice the
// LDS; +3-4 % on the 128..512-channel layers), 16 elsewhere (measured: 32 for dgrad -2.7 %).
static int tile_bk(int bm, int bn, bool dgrad) { return (bm == 128 && bn == 128 && !dgrad) ? 32 : 16; }

// ---- stream-K planning ---------------------------------------------------------------------
template <int BM, int BN, int BK, int WGM, int WGN, bool DGRAD, bool FASTA = false, int AMODE = 0>
static int igemm_occupancy() {
  static int occ = 0;
  if (occ <= 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<BM, BN, BK, WGM, WGN, DGRAD, FASTA, AMODE>, 256, 0) != hipSuccess) {
      (void)hipGetLastError();
      n = 1;
    }
    occ = n > 0 ? n : 1;
  }
  return occ;
}
static int tile_occupancy(int bm, int bn, bool dgrad) {
  if (bm == 128 && bn == 128) {
    if (dgrad) return tile_bk(128, 128, true) == 32 ? igemm_occupancy<128, 128, 32, 2, 2, true>() : igemm_occupancy<128, 128, 16, 2, 2, true>();
    return tile_bk(128, 128, false) == 32 ? igemm_occupancy<128, 128, 32, 2, 2, false>() : igemm_occupancy<128, 128, 16, 2, 2, false>();
  }
  if (bm == 128 && bn == 64) return dgrad ? igemm_occupancy<128, 64, 16, 2, 2, true>() : igemm_occupancy<128, 64, 16, 2, 2, false>();
  if (bm == 64 && bn == 64) return dgrad ? igemm_occupancy<64, 64, 16, 2, 2, true>() : igemm_occupancy<64, 64, 16, 2, 2, false>();
  return dgrad ? igemm_occupancy<128, 32, 16, 4, 1, true>() : igemm_occupancy<128, 32, 16, 4, 1, false>();
}

// Decide between one-tile-per-workgroup ("data parallel") and stream-K for `tiles` tiles of BM x BN
// with KT K-steps each; returns the persistent grid size P (0 = data parallel).  The ResNet layer
// shapes give tile counts like 49 * 2^k, which never divide the 256 CUs x occupancy slots: the last,
// partly filled round of workgroups costs a full round.  Model (fitted to tile_probe.py runs): a CU
// with n resident workgroups runs at min(1, n * solo) of its matrix rate; the fix-up moves each
// piece twice at ~4 TB/s (128-wide tiles) / ~2.5 TB/s (64-wide: more, smaller pieces).
static int plan_streamk(long long tiles, int KT, int bm, int bn, int occ, int bk = 16) {
  if (tiles <= 0 || bm != 128 || bn < 64) return 0;
  const int cus = compute_cus();
  const long long S = (long long)occ * cus;
  if (tiles * KT < 8 * S) return 0;                       // < 8 K-steps per workgroup: overheads dominate
  const double solo = bn >= 128 ? 0.62 : 0.40;
  const double tile_us = 2.0 * bm * bn * (double)bk * KT / (115e6 / cus);   // one tile on a whole CU at 115 TF/s
  const long long full_rounds = tiles / S, rem = tiles - full_rounds * S;
  double t_dp = full_rounds * occ * tile_us;
  if (rem > 0) {
    const long long n = (rem + cus - 1) / cus;
    const double r = n * solo < 1.0 ? n * solo : 1.0;
    t_dp += n * tile_us / r;
  }
  const long long pieces = tiles + S < 2 * S ? tiles + S : 2 * S;
  const double fix_bw = bn >= 128 ? 4.0e6 : 2.5e6;        // bytes / us
  const double t_fix = pieces * (double)bm * bn * 4.0 * 2.0 / fix_bw + 4.0;
  const double t_sk = (double)tiles / cus * tile_us * (bn >= 128 ? 1.0 : 1.04) + t_fix;
  return t_sk < 0.95 * t_dp ? (int)S : 0;
}

// Tile choice over the classes of one launch (rows_c, ktotal_c).  The largest tile the column count
// allows when stream-K will balance it over the CUs; otherwise the largest tile that still gives
// >= 2 workgroups per CU (else the smallest).
static void count_tiles(const long long *rows, const int *ktotal, int ncls, int groups, int ncols, TileChoice t, int bk,
                        long long &tiles, long long &units) {
  tiles = units = 0;
  for (int i = 0; i < ncls; ++i) {
    const long long ti = (long long)groups * ceil_div(rows[i], t.bm) * ceil_div(ncols, t.bn);
    tiles += ti;
    units += ti * (ktotal[i] > 0 ? ceil_div(ktotal[i], bk) : 1);
  }
}
static TileChoice choose_tile_multi(const long long *rows, const int *ktotal, int ncls, int groups, int ncols, bool dgrad) {
  const int cus = compute_cus();
  if (ncols <= 32) return {128, 32};
  long long tiles, units;
  {
    const TileChoice big = ncols >= 128 ? TileChoice{128, 128} : TileChoice{128, 64};
    const int bk = tile_bk(big.bm, big.bn, dgrad);
    count_tiles(rows, ktotal, ncls, groups, ncols, big, bk, tiles, units);
    if (tiles > 0 && plan_streamk(tiles, (int)(units / tiles), big.bm, big.bn, tile_occupancy(big.bm, big.bn, dgrad), bk) > 0)
      return big;
  }
  const TileChoice cand[3] = {{128, 128}, {128, 64}, {64, 64}};
  for (int i = 0; i < 3; ++i) {
    if (cand[i].bn > 64 && ncols < 128) continue;
    count_tiles(rows, ktotal, ncls, groups, ncols, cand[i], 16, tiles, units);
    if (tiles >= 2LL * cus) return cand[i];
  }
  return (ncols >= 64) ? TileChoice{64, 64} : TileChoice{128, 32};
}
static TileChoice choose_tile(long long rows_per_group, int groups, int ncols, int ktotal, bool dgrad) {
  return choose_tile_multi(&rows_per_group, &ktotal, 1, groups, ncols, dgrad);
}

template <int BM, int BN, int BK, int WGM, int WGN, bool DGRAD, bool FASTA = false, int AMODE = 0>
static int launch_igemm_tile(IgemmParams &p, long long tiles, long long units, hipStream_t st) {
  int P = 0;
  if (p.splits == 1 && units > tiles)
    P = plan_streamk(tiles, (int)(units / tiles), BM, BN, igemm_occupancy<BM, BN, BK, WGM, WGN, DGRAD, FASTA, AMODE>(), BK);
  if (P > 0) {
    float *scratch = stream_scratch(st, (size_t)P * 2 * BM * BN);
    if (!scratch) P = 0;                                   // no scratch: plain launch
    else p.slab = scratch;
  }
  if (P > 0) {
    p.sk_tiles = (int)tiles;
    hipLaunchKernelGGL((igemm_kernel<BM, BN, BK, WGM, WGN, DGRAD, FASTA, AMODE>), dim3((unsigned)P), dim3(256), 0, st, p);
    if (check_launch(DGRAD ? "conv_dgrad(stream-K)" : "conv_fprop(stream-K)")) return 1;
    hipLaunchKernelGGL((igemm_fixup_kernel<BM, BN, BK, WGM, WGN, DGRAD>), dim3((unsigned)(P - 1)), dim3(256), 0, st, p, P);
    return check_launch("conv stream-K fix-up");
  }
  p.sk_tiles = 0;
  hipLaunchKernelGGL((igemm_kernel<BM, BN, BK, WGM, WGN, DGRAD, FASTA, AMODE>), dim3((unsigned)(tiles * p.splits)), dim3(256), 0, st, p);
  return check_launch(DGRAD ? "conv_dgrad" : "conv_fprop");
}

// p.cls[0 .. ncls) hold the classes (out_h .. ow_div filled in); tile and unit offsets are set here
template <bool DGRAD>
static int launch_igemm(IgemmParams &p, TileChoice t, hipStream_t st) {
  p.ntiles = ceil_div(p.ncols, t.bn);
  if (p.splits < 1) p.splits = 1;
  if (p.splits == 1) p.ktiles_per_split = 1 << 30;
  const int bk = p.rc_feat ? 16 : tile_bk(t.bm, t.bn, DGRAD);     // the rotate + concat loader has a 16-deep instantiation only
  long long tiles = 0, units = 0;
  for (int i = 0; i < p.ncls; ++i) {
    IgemmClass &c = p.cls[i];
    c.mtiles_per_group = ceil_div(c.rows_per_group, t.bm);
    c.KT = c.ktotal > 0 ? ceil_div(c.ktotal, bk) : 1;
    c.per_div = make_fastdiv((unsigned)(c.ntaps > 0 ? c.ntaps * (32 / bk) : 1));
    c.tile0 = (int)tiles;
    c.unit0 = units;
    const long long ti = (long long)p.groups * c.mtiles_per_group * p.ntiles;
    tiles += ti;
    units += ti * c.KT;
  }
  MVG_REQUIRE(tiles * p.splits < (1LL << 31), "conv: grid too large");
  MVG_REQUIRE(p.splits == 1 || p.ncls == 1, "conv: split-K with several classes");
  if (tiles <= 0) return 0;
  // uniform-tap loader: every class has whole K-steps inside one tap and at most 32 taps
  bool fasta = true;
  for (int i = 0; i < p.ncls; ++i) {
    const IgemmClass &c = p.cls[i];
    fasta = fasta && c.ntaps >= 1 && c.ntaps <= 32 && c.ktotal % bk == 0 && p.src_c % bk == 0;
  }
  if constexpr (!DGRAD) {
    if (p.rc_feat != nullptr) {
      MVG_REQUIRE(fasta && t.bn >= 64, "fuser GEMM: shape not covered by the rotate + concat loader");
      if (t.bm == 128 && t.bn == 128) return launch_igemm_tile<128, 128, 16, 2, 2, false, true, 1>(p, tiles, units, st);
      if (t.bm == 128 && t.bn == 64) return launch_igemm_tile<128, 64, 16, 2, 2, false, true, 1>(p, tiles, units, st);
      return launch_igemm_tile<64, 64, 16, 2, 2, false, true, 1>(p, tiles, units, st);
    }
  }
  if (t.bm == 128 && t.bn == 128) {
    if (bk == 32) {
      if (fasta) return launch_igemm_tile<128, 128, 32, 2, 2, DGRAD, true>(p, tiles, units, st);
      return launch_igemm_tile<128, 128, 32, 2, 2, DGRAD>(p, tiles, units, st);
    }
    if (fasta) return launch_igemm_tile<128, 128, 16, 2, 2, DGRAD, true>(p, tiles, units, st);
    return launch_igemm_tile<128, 128, 16, 2, 2, DGRAD>(p, tiles, units, st);
  }
  if (t.bm == 128 && t.bn == 64) {
    if (fasta) return launch_igemm_tile<128, 64, 16, 2, 2, DGRAD, true>(p, tiles, units, st);
    return launch_igemm_tile<128, 64, 16, 2, 2, DGRAD>(p, tiles, units, st);
  }
  if (t.bm == 64 && t.bn == 64) {
    if (fasta) return launch_igemm_tile<64, 64, 16, 2, 2, DGRAD, true>(p, tiles, units, st);     // the fusion block's Linears
    return launch_igemm_tile<64, 64, 16, 2, 2, DGRAD>(p, tiles, units, st);
  }
  return launch_igemm_tile<128, 32, 16, 4, 1, DGRAD>(p, tiles, units, st);
}

// split-K plan for a GEMM whose tile grid cannot fill the device: returns splits (>= 1) and sets
// ktiles_per_split; bounded by the caller's workspace.
static int plan_splitk(IgemmParams &p, TileChoice t, size_t ws_floats, bool dgrad) {
  p.splits = 1;
  if (ws_floats == 0 || p.ncols % 4 != 0) return 1;
  const int cus = compute_cus();
  const long long tiles = (long long)p.groups * ceil_div(p.rows_per_group, t.bm) * ceil_div(p.ncols, t.bn);
  if (tiles >= cus) return 1;
  // K-steps in the units of the kernel that will run (the 128x128 fprop tile steps by 32: counting in
  // 16s here made the trailing splits start past the end of K and left their slabs unwritten)
  const int bk = p.rc_feat ? 16 : tile_bk(t.bm, t.bn, dgrad);
  const int KT = ceil_div(p.ktotal, bk);
  long long s = (2LL * cus + tiles - 1) / tiles;
  if (s > (long long)KT * bk / 128) s = (long long)KT * bk / 128;      // >= 128 k per split
  const long long slab = (long long)p.groups * p.rows_per_group * p.ncols;
  if (s > (long long)(ws_floats / (size_t)slab)) s = (long long)(ws_floats / (size_t)slab);
  if (s < 2) return 1;
  p.ktiles_per_split = ceil_div(KT, s);
  p.splits = ceil_div(KT, p.ktiles_per_split);
  return p.splits;
}

static int launch_splitk_reduce(const IgemmParams &p, hipStream_t st) {
  const long long n = (long long)p.groups * p.rows_per_group * p.ncols;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float4 *)p.slab, (float4 *)p.out,
                     n / 4, p.splits, p.ncols / 4, (const float4 *)p.bias, p.relu, (const float4 *)p.mask,
                     (const float4 *)p.addend);
  return check_launch("splitk_reduce");
}

static int wave_rows(TileChoice t) { return (t.bn == 32) ? 32 : t.bm / 2; }

}  // namespace mvg

using namespace mvg;

extern "C" {

int mvg_conv_stats_partials(const mvg_conv_desc *d, int32_t *rows_per_partial) {
  if (validate(d)) return -1;
  const long long rows = (long long)d->n * d->ho * d->wo;
  const TileChoice t = choose_tile(rows, d->groups, d->cout, d->r * d->s * d->cin, false);
  const int wr = wave_rows(t);
  if (rows_per_partial) *rows_per_partial = wr;
  return ceil_div(rows, t.bm) * (t.bm / wr);
}

struct RotCat {            // generated cross-view input, see IgemmParams::rc_*
  const float *feat, *rel;
  const int *row_img, *row_src;
  int cf, nvec;
  long long img_rows, feat_rows;
};

static int fprop_impl(const mvg_conv_desc *d, const float *x, const float *wgt, float *y, const float *bias, int relu,
                      float *stats, float *ws, size_t ws_floats, void *stream, const float *scale = nullptr,
                      const float *residual = nullptr, const RotCat *rc = nullptr) {
  if (validate(d)) return 2;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  if (rc) {
    p.rc_feat = rc->feat;
    p.rc_rel = rc->rel;
    p.rc_row_img = rc->row_img;
    p.rc_row_src = rc->row_src;
    p.rc_cf = rc->cf;
    p.rc_nvec = rc->nvec;
    p.rc_nvec_shift = ilog2_exact(rc->nvec);
    p.rc_img_bytes = 4ll * rc->img_rows * rc->cf;
    p.rc_feat_bytes = 4ll * rc->feat_rows * 3 * rc->nvec;
  }
  p.a = x;
  p.b = wgt;
  p.out = y;
  p.bias = bias;
  p.scale = scale;
  p.addend = residual;
  p.relu = relu;
  p.stats = stats;
  p.groups = d->groups;
  p.out_h = d->ho;
  p.out_w = d->wo;
  p.src_h = d->h;
  p.src_w = d->w;
  p.src_c = d->cin;
  p.src_c_shift = (d->r * d->s > 1) ? ilog2_exact(d->cin) : 0;
  p.ncols = d->cout;
  p.r = d->r;
  p.s = d->s;
  p.rs = d->r * d->s;
  p.stride = d->stride;
  p.stride_shift = d->stride == 2 ? 1 : 0;
  p.pad = d->pad;
  p.ktotal = d->r * d->s * d->cin;
  p.cin = d->cin;
  p.rows_per_group = (long long)d->n * d->ho * d->wo;
  p.src_img_stride = (long long)d->h * d->w * d->cin;
  p.imgs_per_group = d->n;
  p.ntaps = d->r * d->s;
  p.tap_ns = d->s;
  p.tap_step = 1;
  p.cls_step = 1;
  p.a_group_bytes = 4ll * d->n * p.src_img_stride;
  p.b_bytes = 4ll * d->cout * p.ktotal;
  MVG_REQUIRE(p.a_group_bytes < 0x7FFFFFF0ll && p.b_bytes < 0x7FFFFFF0ll, "conv: a group / the weights exceed 2 GiB");
  p.tap_ns_div = make_fastdiv((unsigned)p.tap_ns);
  p.ohw_div = make_fastdiv((unsigned)(p.out_h * p.out_w));
  p.ow_div = make_fastdiv((unsigned)p.out_w);
  const TileChoice t = choose_tile(p.rows_per_group, d->groups, d->cout, p.ktotal, false);
  const double flops = 2.0 * d->groups * (double)p.rows_per_group * d->cout * d->r * d->s * alg_cin(d);
  const double bytes = 4.0 * (d->groups * (double)d->n * d->h * d->w * alg_cin(d) + (double)d->cout * d->r * d->s * alg_cin(d) +
                              d->groups * (double)p.rows_per_group * d->cout);
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  ProfScope ps(lin ? MVG_K_LINEAR_FPROP : MVG_K_CONV_FPROP, (hipStream_t)stream, flops, bytes);
  p.ncls = 1;
  class_from_params(p.cls[0], p);
  if (!stats && plan_splitk(p, t, ws ? ws_floats : 0, false) > 1) {
    p.slab = ws;
    if (launch_igemm<false>(p, t, (hipStream_t)stream)) return 1;
    return launch_splitk_reduce(p, (hipStream_t)stream);
  }
  return launch_igemm<false>(p, t, (hipStream_t)stream);
}

static int dgrad_impl(const mvg_conv_desc *d, const float *dy, const float *wgt, float *dx, const float *mask,
                      const float *addend, float *ws, size_t ws_floats, void *stream) {
  if (validate(d)) return 2;
  MVG_REQUIRE(d->cout % 4 == 0, "dgrad: cout %% 4 != 0 (%d)", d->cout);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.a = dy;
  p.b = wgt;
  p.out = dx;
  p.mask = mask;
  p.addend = addend;
  p.groups = d->groups;
  p.out_h = d->h;
  p.out_w = d->w;
  p.src_h = d->ho;
  p.src_w = d->wo;
  p.src_c = d->cout;
  p.src_c_shift = (d->r * d->s > 1) ? ilog2_exact(d->cout) : 0;
  p.ncols = d->cin;
  p.r = d->r;
  p.s = d->s;
  p.rs = d->r * d->s;
  p.stride = d->stride;
  p.stride_shift = d->stride == 2 ? 1 : 0;
  p.pad = d->pad;
  p.ktotal = d->r * d->s * d->cout;
  p.cin = d->cin;
  p.src_img_stride = (long long)d->ho * d->wo * d->cout;
  p.imgs_per_group = d->n;
  p.full_h = d->h;
  p.full_w = d->w;
  p.a_group_bytes = 4ll * d->n * p.src_img_stride;
  p.b_bytes = 4ll * d->cout * d->r * d->s * d->cin;
  MVG_REQUIRE(p.a_group_bytes < 0x7FFFFFF0ll && p.b_bytes < 0x7FFFFFF0ll, "conv: a group / the weights exceed 2 GiB");
  // algorithmic flops: the transposed conv touches each (output pixel, tap) pair of the fprop once
  const double flops = 2.0 * d->groups * (double)d->n * d->ho * d->wo * d->cout * d->r * d->s * alg_cin(d);
  const double bytes = 4.0 * (d->groups * (double)d->n * d->ho * d->wo * d->cout + (double)d->cout * d->r * d->s * alg_cin(d) +
                              d->groups * (double)d->n * d->h * d->w * alg_cin(d));
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  ProfScope ps(lin ? MVG_K_LINEAR_DGRAD : MVG_K_CONV_DGRAD, (hipStream_t)stream, flops, bytes);
  const int step = d->stride;
  IgemmParams m = p;                   // the merged fp32 launch: every parity class in one grid
  m.ncls = 0;
  long long cls_rows[4];
  int cls_k[4];
  for (int py = 0; py < step; ++py)
    for (int px = 0; px < step; ++px) {
      const int sub_h = (d->h - py + step - 1) / step, sub_w = (d->w - px + step - 1) / step;
      if (sub_h <= 0 || sub_w <= 0) continue;
      const int r0 = (py + d->pad) % step, s0 = (px + d->pad) % step;
      const int nr = r0 < d->r ? (d->r - r0 + step - 1) / step : 0;
      const int ns = s0 < d->s ? (d->s - s0 + step - 1) / step : 0;
      IgemmParams q = p;
      q.out_h = sub_h;
      q.out_w = sub_w;
      q.rows_per_group = (long long)d->n * sub_h * sub_w;
      q.ntaps = nr * ns;
      q.tap_ns = ns > 0 ? ns : 1;
      q.tap_ns_div = make_fastdiv((unsigned)q.tap_ns);
      q.ohw_div = make_fastdiv((unsigned)(sub_h * sub_w));
      q.ow_div = make_fastdiv((unsigned)sub_w);
      q.tap_r0 = r0;
      q.tap_s0 = s0;
      q.tap_step = step;
      q.ktotal = nr * ns * d->cout;
      q.cls_step = step;
      q.cls_py = py;
      q.cls_px = px;
      q.cls_cy = (py + d->pad - r0) / step;
      q.cls_cx = (px + d->pad - s0) / step;
      if (q.ntaps == 0) {
        // no tap reaches this class: dx = addend (nothing to do when the caller accumulates in place)
        if (addend != dx || !addend) {
          MVG_REQUIRE(d->cin % 4 == 0, "dgrad: cin %% 4 != 0");
          const long long n = (long long)d->groups * d->n * sub_h * sub_w * (d->cin / 4);
          long long blocks = (n + 255) / 256;
          if (blocks > 4096) blocks = 4096;
          hipLaunchKernelGGL(dgrad_empty_class_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)dx,
                             (const float4 *)addend, n, sub_h, sub_w, d->h, d->w, d->cin / 4, py, px);
          if (check_launch("dgrad(empty class)")) return 1;
        }
        continue;
      }
      if (m.ncls == 0) {               // launch-wide fields that depend on the stride
        m.tap_step = step;
        m.cls_step = step;
        m.rows_per_group = q.rows_per_group;
        m.ktotal = q.ktotal;
        m.out_h = q.out_h;
        m.out_w = q.out_w;
      }
      cls_rows[m.ncls] = q.rows_per_group;
      cls_k[m.ncls] = q.ktotal;
      class_from_params(m.cls[m.ncls++], q);
    }
  if (m.ncls == 0) return 0;
  // longest class first: with one tile per workgroup the short tiles then fill the tail
  for (int i = 1; i < m.ncls; ++i)
    for (int j = i; j > 0 && cls_k[j] > cls_k[j - 1]; --j) {
      const IgemmClass tc = m.cls[j];
      m.cls[j] = m.cls[j - 1];
      m.cls[j - 1] = tc;
      const long long tr = cls_rows[j];
      cls_rows[j] = cls_rows[j - 1];
      cls_rows[j - 1] = tr;
      const int tk = cls_k[j];
      cls_k[j] = cls_k[j - 1];
      cls_k[j - 1] = tk;
    }
  m.no_remap = m.ncls > 1;
  const TileChoice t = choose_tile_multi(cls_rows, cls_k, m.ncls, d->groups, d->cin, true);
  if (m.ncls == 1 && step == 1 && plan_splitk(m, t, ws ? ws_floats : 0, true) > 1) {
    m.slab = ws;
    if (launch_igemm<true>(m, t, (hipStream_t)stream)) return 1;
    return launch_splitk_reduce(m, (hipStream_t)stream);
  }
  return launch_igemm<true>(m, t, (hipStream_t)stream);
}

static mvg_conv_desc linear_desc(int rows, int fin, int fout) {
  mvg_conv_desc d = {1, rows, 1, 1, fin, fout, 1, 1, 1, 0, 1, 1};
  return d;
}

}  // extern "C" (helpers above are static)

extern "C" {

int mvg_conv_fprop(const mvg_conv_desc *d, const float *x, const float *wgt, float *y, const float *bias, int relu,
                   float *stats, void *stream) {
  return fprop_impl(d, x, wgt, y, bias, relu, stats, nullptr, 0, stream);
}

int mvg_conv_dgrad(const mvg_conv_desc *d, const float *dy, const float *wgt, float *dx, const float *mask,
                   const float *addend, void *stream) {
  return dgrad_impl(d, dy, wgt, dx, mask, addend, nullptr, 0, stream);
}

int mvg_conv_fprop_affine(const mvg_conv_desc *d, const float *x, const float *wgt, float *out, const float *scale,
                          const float *shift, const float *residual, int relu, void *stream) {
  MVG_REQUIRE(scale && shift, "fprop_affine: scale and shift are required");
  return fprop_impl(d, x, wgt, out, shift, relu, nullptr, nullptr, 0, stream, scale, residual);
}

size_t mvg_linear_workspace_floats(int rows, int fin, int fout) {
  // up to 16 K-slices of the larger of the two GEMM outputs (fprop: rows x fout, dgrad: rows x fin)
  const size_t m = (size_t)rows * (size_t)(fin > fout ? fin : fout);
  return 16 * m;
}

int mvg_linear_fprop(const float *x, const float *w, const float *bias, int relu, float *y, int rows, int fin, int fout,
                     float *workspace, size_t ws_floats, void *stream) {
  const mvg_conv_desc d = linear_desc(rows, fin, fout);
  return fprop_impl(&d, x, w, y, bias, relu, nullptr, workspace, ws_floats, stream);
}

int mvg_linear_dgrad(const float *dy, const float *w, const float *mask, const float *addend, float *dx, int rows,
                     int fin, int fout, float *workspace, size_t ws_floats, void *stream) {
  const mvg_conv_desc d = linear_desc(rows, fin, fout);
  return dgrad_impl(&d, dy, w, dx, mask, addend, workspace, ws_floats, stream);
}

static TileChoice wgrad_tile(const mvg_conv_desc *d) {
  const int ncols = d->r * d->s * d->cin;
  if (d->cout >= 128 && ncols >= 128) return {128, 128};
  if (d->cout >= 64 && ncols >= 128) return {64, 128};
  if (ncols >= 64 && d->cout >= 64) return {64, 64};
  if (d->cout < 64) return {32, 128};   // skinny cout (e.g. 2-wide head is handled elsewhere)
  return {128, 32};
}

}  // extern "C"

template <int BM, int BN, int WGM, int WGN>
static int wgrad_occupancy_t() {
  static int occ = 0;
  if (occ <= 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_kernel<BM, BN, 16, WGM, WGN, true>, 256, 0) != hipSuccess) {
      (void)hipGetLastError();
      n = 2;
    }
    occ = n > 0 ? n : 1;
  }
  return occ;
}
static int wgrad_occupancy(TileChoice t) {
  if (t.bm == 128 && t.bn == 128) return wgrad_occupancy_t<128, 128, 2, 2>();
  if (t.bm == 64 && t.bn == 128) return wgrad_occupancy_t<64, 128, 2, 2>();
  if (t.bm == 64 && t.bn == 64) return wgrad_occupancy_t<64, 64, 2, 2>();
  if (t.bm == 32 && t.bn == 128) return wgrad_occupancy_t<32, 128, 1, 4>();
  return wgrad_occupancy_t<128, 32, 4, 1>();
}

extern "C" {

int mvg_conv_wgrad_splits(const mvg_conv_desc *d) {
  if (validate(d)) return -1;
  const TileChoice t = wgrad_tile(d);
  const int ncols = d->r * d->s * d->cin;
  const long long tiles = (long long)ceil_div(d->cout, t.bm) * ceil_div(ncols, t.bn);
  const long long pixels = (long long)d->groups * d->n * d->ho * d->wo;
  const int cus = compute_cus();
  // one resident round: tiles x splits <= CUs x workgroups-per-CU (rounding the split count UP puts a
  // handful of workgroups into a second round that costs as much as the first)
  int wpc = wgrad_occupancy(t);
  if (wpc > 4) wpc = 4;
  long long want = ((long long)wpc * cus) / tiles;
  long long maxs = pixels / 256;                           // at least 256 pixels (16 K-steps) per split
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

static int wgrad_impl(const mvg_conv_desc *d, const float *x, const float *dy, float *dw, float *db, float *workspace,
                      int splits, int accumulate, void *stream, const RotCat *rc = nullptr) {
  if (validate(d)) return 2;
  MVG_REQUIRE(d->cout % 4 == 0, "wgrad: cout %% 4 != 0 (%d)", d->cout);
  MVG_REQUIRE(splits >= 1, "wgrad: splits < 1");
  MVG_REQUIRE(splits == 1 || workspace != nullptr, "wgrad: workspace required for splits > 1");
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.dy = dy;
  p.h = d->h;
  p.w = d->w;
  p.cin = d->cin;
  p.cout = d->cout;
  p.r = d->r;
  p.s = d->s;
  p.stride = d->stride;
  p.pad = d->pad;
  p.ho = d->ho;
  p.wo = d->wo;
  p.ncols = d->r * d->s * d->cin;
  p.pixels = (long long)d->groups * d->n * d->ho * d->wo;
  p.pixels_per_split = ((p.pixels + splits - 1) / splits + 15) / 16 * 16;
  p.x_bytes = 4ll * d->groups * d->n * d->h * d->w * d->cin;
  p.ohw_div = make_fastdiv((unsigned)(d->ho * d->wo));
  p.wo_div = make_fastdiv((unsigned)d->wo);
  p.cin_div = make_fastdiv((unsigned)d->cin);
  p.s_div = make_fastdiv((unsigned)d->s);
  MVG_REQUIRE(p.pixels_per_split * d->cout * 4ll < 0x7FFFFFF0ll, "wgrad: split too large for 32-bit offsets");
  MVG_REQUIRE(4ll * (p.pixels_per_split / (d->ho * d->wo) + 2) * d->h * d->w * d->cin < 0x7FFFFFF0ll,
              "wgrad: split too large for 32-bit offsets");
  const TileChoice t = wgrad_tile(d);
  p.mtiles = ceil_div(d->cout, t.bm);
  p.ntiles = ceil_div(p.ncols, t.bn);
  p.out = splits == 1 ? dw : workspace;
  p.accumulate = (splits == 1) ? accumulate : 0;
  if (rc) {
    p.rc_feat = rc->feat;
    p.rc_rel = rc->rel;
    p.rc_row_img = rc->row_img;
    p.rc_row_src = rc->row_src;
    p.rc_cf = rc->cf;
    p.rc_nvec = rc->nvec;
    p.rc_img_bytes = 4ll * rc->img_rows * rc->cf;
    p.rc_feat_bytes = 4ll * rc->feat_rows * 3 * rc->nvec;
  }
  float *db_slab = workspace ? workspace + (size_t)splits * d->cout * p.ncols : nullptr;     // after the dw slabs
  p.db = db ? (splits == 1 ? db : db_slab) : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  {
    const double flops = 2.0 * (double)p.pixels * d->cout * d->r * d->s * alg_cin(d);
    const double bytes = 4.0 * ((double)d->groups * d->n * d->h * d->w * alg_cin(d) + (double)p.pixels * d->cout +
                                (double)d->cout * d->r * d->s * alg_cin(d));
    ProfScope ps(lin ? MVG_K_LINEAR_WGRAD : MVG_K_CONV_WGRAD, st, flops, bytes);
    MVG_REQUIRE((long long)p.mtiles * p.ntiles * splits < (1LL << 31), "wgrad: grid too large");
    dim3 grid(p.mtiles * p.ntiles * splits), block(256);
    // incremental pixel stepping: needs >= 32 pixels per image (one image wrap per step at most), 24-bit
    // factors in the offset multiplies and 32-bit x offsets
    const bool incr = (long long)d->ho * d->wo >= 32 && d->ho < (1 << 20) && d->wo < (1 << 20) &&
                      (long long)d->stride * d->w * d->cin * 4 < (1 << 24);
#define MVG_WGRAD_LAUNCH(BM_, BN_, WGM_, WGN_)                                                              \
  do {                                                                                                      \
    if (incr) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, 16, WGM_, WGN_, true>), grid, block, 0, st, p);    \
    else hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, 16, WGM_, WGN_, false>), grid, block, 0, st, p);        \
  } while (0)
    if (rc) {
      MVG_REQUIRE(t.bn == 128 && (t.bm == 128 || t.bm == 64), "fuser wgrad: shape not covered by the generated-input loader");
      if (t.bm == 128) hipLaunchKernelGGL((wgrad_kernel<128, 128, 16, 2, 2, false, 1>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((wgrad_kernel<64, 128, 16, 2, 2, false, 1>), grid, block, 0, st, p);
    } else if (t.bm == 128 && t.bn == 128) MVG_WGRAD_LAUNCH(128, 128, 2, 2);
    else if (t.bm == 64 && t.bn == 128) MVG_WGRAD_LAUNCH(64, 128, 2, 2);
    else if (t.bm == 64 && t.bn == 64) MVG_WGRAD_LAUNCH(64, 64, 2, 2);
    else if (t.bm == 32 && t.bn == 128) MVG_WGRAD_LAUNCH(32, 128, 1, 4);
    else MVG_WGRAD_LAUNCH(128, 32, 4, 1);
#undef MVG_WGRAD_LAUNCH
    if (check_launch("conv_wgrad")) return 1;
  }
  if (splits > 1) {
    const long long n = (long long)d->cout * p.ncols;
    MVG_REQUIRE(n % 4 == 0, "wgrad: weight elements %% 4 != 0");
    ProfScope ps(MVG_K_WGRAD_REDUCE, st, 0.0, 4.0 * n * (splits + 1));
    const int lanes = splits >= 32 ? 16 : (splits >= 8 ? 4 : 1);
    const long long blocks = (n / 4 + 256 / lanes - 1) / (256 / lanes);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, workspace, dw, n / 4, splits, accumulate,
                       lanes);
    if (check_launch("wgrad_reduce")) return 1;
    if (db) {
      const long long nb = d->cout;
      const long long bblocks = (nb / 4 + 256 / lanes - 1) / (256 / lanes);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)bblocks), dim3(256), 0, st, db_slab, db, nb / 4, splits, accumulate,
                         lanes);
      if (check_launch("wgrad_reduce(bias)")) return 1;
    }
  }
  return 0;
}

int mvg_conv_wgrad(const mvg_conv_desc *d, const float *x, const float *dy, float *dw, float *workspace, int splits,
                   int accumulate, void *stream) {
  return wgrad_impl(d, x, dy, dw, nullptr, workspace, splits, accumulate, stream);
}

static int check_rotcat(int rows, int cf, int nvec, int fout, const float *img_feat, const float *feat, const int32_t *row_img,
                        const int32_t *row_src, int img_rows, int feat_rows) {
  MVG_REQUIRE(img_feat && feat && row_img && row_src, "fuser GEMM: null argument");
  MVG_REQUIRE(rows > 0 && img_rows > 0 && feat_rows > 0, "fuser GEMM: bad sizes");
  MVG_REQUIRE(cf % 128 == 0 && nvec % 128 == 0 && ilog2_exact(nvec) >= 0 && fout % 64 == 0 && fout >= 64,
              "fuser GEMM: cf and nvec must be multiples of 128 (nvec a power of two), fout a multiple of 64");
  MVG_REQUIRE(4ll * img_rows * cf < 0x7FFFFFF0ll && 12ll * feat_rows * nvec < 0x7FFFFFF0ll, "fuser GEMM: operands exceed 2 GiB");
  return 0;
}

int mvg_fuser_fprop(const float *img_feat, const float *feat, const float *rel, const int32_t *row_img, const int32_t *row_src,
                    const float *w, const float *bias, int relu, float *y, int rows, int cf, int nvec, int fout, int img_rows,
                    int feat_rows, float *workspace, size_t ws_floats, void *stream) {
  if (check_rotcat(rows, cf, nvec, fout, img_feat, feat, row_img, row_src, img_rows, feat_rows)) return 2;
  const mvg_conv_desc d = linear_desc(rows, cf + 3 * nvec, fout);
  const RotCat rc = {feat, rel, row_img, row_src, cf, nvec, img_rows, feat_rows};
  return fprop_impl(&d, img_feat, w, y, bias, relu, nullptr, workspace, ws_floats, stream, nullptr, nullptr, &rc);
}

int mvg_fuser_wgrad(const float *img_feat, const float *feat, const float *rel, const int32_t *row_img, const int32_t *row_src,
                    const float *dy, float *dw, float *db, int rows, int cf, int nvec, int fout, int img_rows, int feat_rows,
                    float *workspace, int splits, int accumulate, void *stream) {
  if (check_rotcat(rows, cf, nvec, fout, img_feat, feat, row_img, row_src, img_rows, feat_rows)) return 2;
  const mvg_conv_desc d = linear_desc(rows, cf + 3 * nvec, fout);
  const RotCat rc = {feat, rel, row_img, row_src, cf, nvec, img_rows, feat_rows};
  return wgrad_impl(&d, img_feat, dy, dw, db, workspace, splits, accumulate, stream, &rc);
}

int mvg_linear_wgrad(const float *x, const float *dy, float *dw, float *db, int rows, int fin, int fout, float *workspace,
                     int splits, int accumulate, void *stream) {
  const mvg_conv_desc d = linear_desc(rows, fin, fout);
  MVG_REQUIRE(dw != nullptr, "linear_wgrad: dw is required");
  return wgrad_impl(&d, x, dy, dw, db, workspace, splits, accumulate, stream);
}

}  // extern "C"
