// bf16 matrix-core helpers shared by conv_bf16.hip (bf16 storage) and conv_split.hip (fp32 values held as three
// bf16 pieces): vector types, packing, and the LDS-staged tile epilogue.  Static / inline only (conv_shared.h).
#pragma once
#include "conv_shared.h"
#include "elem.h"

namespace mvg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {          // round-to-nearest-even, NaN-safe (plain casts)
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

__device__ __forceinline__ void ld8(const float *p, float (&v)[8]) {       // 8 floats, 16-byte aligned
  const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Epilogue shared by the bf16 GEMM kernels: BN partial statistics from the fp32 accumulators, then the tile goes
// through LDS (fp32, the operand buffers are free after the K loop) so that global stores are 16-byte vectors along
// the channel axis, with bias / ReLU / mask / addend applied in fp32 and ONE rounding to bf16 (F32IO: fp32 stores).
// Workgroup = WGM x 2 waves (WGM * 128 threads), wave tile (BM / WGM) x (BN / 2).  PASSES > 1: the staging tile
// holds BM / PASSES rows at a time (the wave rows take turns), for kernels whose LDS is smaller than the full tile.
// ACC16: the accumulators are 16 x 16 tiles of v_mfma_f32_16x16x32_bf16 (f32x4 acc[WTM / 16][WTN / 16]: column =
// lane & 15, row = 4 * (lane >> 4) + e) instead of 32 x 32 tiles of v_mfma_f32_32x32x16_bf16 (f32x16 acc[WTM / 32][WTN / 32]:
// column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)).
// BNF: the kernel carries the fused BatchNorm-backward reduce (IgemmParams::bn_*); its shared memory then holds
// bf16_epilogue_bytes<...>() bytes: the staging tile, the row offsets and 4 * BN floats of per-channel constants.
template <int BM, int BN, int PASSES, bool BNF>
constexpr int bf16_epilogue_bytes() {
  return (BM / PASSES) * (BN + 4) * 4 + BM * 4 + (BNF ? 4 * BN * 4 : 0);
}

// LIN (split kernels running a Linear layer of the fusion block): the epilogue can also (i) leave max |stored value| of an
// fp32 result in *p.out_absmax (atomicMax on the float's bits: order-independent), the bound the next split of that
// tensor scales by, and (ii) store an sp result times the power of two 2^k that a bound on the result allows -
// |acc| <= ktotal * 2^30 in the operands' scaled units, so |result| <= ktotal * 2^30 * osc + max |bias| - with
// *p.out_sinv = 2^-k for the consumers (IgemmParams::out_sinv / bias_absmax).
// WGN: waves along the tile's columns (2; 1 for the 256 x 64 tile of the split kernels: four wave rows of 64 x 64).
template <int BM, int BN, int WGM, bool DGRAD, bool F32IO, int PASSES = 1, bool ACC16 = false, bool BNF = false, bool LIN = false, int WGN = 2,
          class AccT>
__device__ __forceinline__ void bf16_epilogue(const IgemmParams &p, const IgemmClass &c, AccT &acc,
                                              unsigned short *smem, int tid, int g, int mtile, int ntile) {
  constexpr int NT = WGM * WGN * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TS = ACC16 ? 16 : 32, NE = ACC16 ? 4 : 16;      // tile side, accumulator registers per tile
  constexpr int TM = WTM / TS, TN = WTN / TS;
  constexpr int LDO = BN + 4;                // fp32 staging tile
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = ACC16 ? (lane & 15) : (lane & 31), lh = ACC16 ? (lane >> 4) : (lane >> 5);
  // row of accumulator register e inside its tile
  auto erow = [&](int e) { return ACC16 ? 4 * lh + e : (e & 3) + 8 * (e >> 2) + 4 * lh; };
  const int ohw = c.out_h * c.out_w;
  const long long row_base = (long long)mtile * BM + wm * WTM;
  // split operands carry per-tensor power-of-two scales: exact to undo on the accumulators
  const float osc = F32IO ? (p.a_sinv ? *p.a_sinv : 1.f) * (p.b_sinv ? *p.b_sinv : 1.f) : 1.f;
  float lin_scale = 1.f, lin_max = 0.f;
  if constexpr (LIN && F32IO) {
    if (!DGRAD && p.out_sp && p.out_sinv) {
      const float bound = (float)p.ktotal * 1073741824.f * osc + (p.bias_absmax ? *p.bias_absmax : 0.f);
      lin_scale = sp_scale_for(bound);
      if (tid == 0 && mtile == 0 && ntile == 0 && g == 0) *p.out_sinv = 1.f / lin_scale;
    }
  }
  if (!DGRAD && p.stats) {
    // per-wave partial over its WTM rows: column sum and sum of squares centred on the partial's own mean
    long long cnt_ll = c.rows_per_group - row_base;
    const int cnt = cnt_ll <= 0 ? 0 : (cnt_ll > WTM ? WTM : (int)cnt_ll);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = ntile * BN + wn * WTN + j * TS + li;
      float csum = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int r = i * TS + erow(e);
          csum += (r < cnt) ? acc[i][j][e] : 0.f;
        }
      csum += __shfl_xor(csum, 32, 64);
      if (ACC16) csum += __shfl_xor(csum, 16, 64);
      const float mean = cnt > 0 ? csum / (float)cnt : 0.f;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int r = i * TS + erow(e);
          const float dlt = acc[i][j][e] - mean;
          q += (r < cnt) ? dlt * dlt : 0.f;
        }
      q += __shfl_xor(q, 32, 64);
      if (ACC16) q += __shfl_xor(q, 16, 64);
      long long P = p.stats_partials ? p.stats_partials : (long long)c.mtiles_per_group * WGM;
      long long pi = (long long)mtile * WGM + wm;
      // (a wave row entirely beyond the rows holds no partial: with 256-row tiles it may lie beyond the caller's P = 2 per 128 rows)
      if (lh == 0 && col < p.ncols && (pi < P || p.stats_fold)) {
        int nc = p.ncols, ch = col;
        if (p.stats_fold) {                      // columns c and c + ncols/2 are the same channel (two output columns per row)
          nc = p.ncols >> 1;
          pi = 2 * pi + (col >= nc);
          ch = col - (col >= nc ? nc : 0);
          P *= 2;
        }
        float *st = p.stats + (((long long)g * P + pi) * 2) * nc;
        st[ch] = csum * osc;
        st[nc + ch] = q * (osc * osc);
      }
    }
  }
  // accumulators -> fp32 tile in LDS (the operand buffers are free: the K loop ended with a barrier)
  static_assert(WGM % PASSES == 0, "bf16_epilogue: the wave rows must divide into the passes");
  constexpr int PR = BM / PASSES;              // rows per pass
  float *ot = reinterpret_cast<float *>(smem);
  int *rowoff = reinterpret_cast<int *>(ot + PR * LDO);
  // element offset of every tile row inside its group's output tensor (-1: beyond the end)
  for (int r = tid; r < BM; r += NT) {
    const long long m = (long long)mtile * BM + r;
    int off = -1;
    if (m < c.rows_per_group) {
      if (DGRAD && p.cls_step == 2) {
        const int rr = (int)m;
        const int img = (int)fdiv((unsigned)rr, c.ohw_div), rem = rr - img * ohw;
        const int y2 = (int)fdiv((unsigned)rem, c.ow_div), x2 = rem - y2 * c.out_w;
        off = ((img * p.full_h + 2 * y2 + c.cls_py) * p.full_w + 2 * x2 + c.cls_px) * p.ncols;
      } else {
        off = (int)m * p.ncols;
      }
    }
    rowoff[r] = off;
  }
  const long long gelems = DGRAD ? (long long)p.imgs_per_group * p.full_h * p.full_w * p.ncols
                                 : c.rows_per_group * (long long)p.ncols;
  unsigned short *out_g = reinterpret_cast<unsigned short *>(p.out) + (long long)g * gelems;
  const unsigned short *add_g = p.addend ? reinterpret_cast<const unsigned short *>(p.addend) + (long long)g * gelems : nullptr;
  const unsigned short *mask_g = p.mask ? reinterpret_cast<const unsigned short *>(p.mask) + (long long)g * gelems : nullptr;
  float *out_f = p.out + (long long)g * gelems;
  const float *add_f = p.addend ? p.addend + (long long)g * gelems : nullptr;
  const float *mask_f = p.mask ? p.mask + (long long)g * gelems : nullptr;
  constexpr int CV = BN / 8;                 // 16-byte output vectors per tile row
  // Backward-data fused with the BatchNorm-backward REDUCE pass of the unit whose output gradient this launch produces
  // (IgemmParams::bn_*): the gradient is masked by that unit's ReLU here, stored masked, and s1 = sum(dz),
  // s2 = sum(dz * xhat) are added up per thread (its 8 channels are fixed: NT % CV == 0), per workgroup through
  // LDS, and written as one partial per (group, class, row tile) for bn_bwd_finalize.  bf16 storage: the sums are
  // those of the ROUNDED gradient, the values the apply pass will read.
  // The per-channel constants (mean, invstd, the ReLU's scale / shift) wait in LDS and are re-read in every iteration:
  // 32 values a thread would otherwise hold next to the accumulators and the 24 running sums - the kernels keep four
  // workgroups per CU (<= 128 VGPRs) with the reduce fused.
  const bool bnf = BNF && DGRAD && p.bn_part != nullptr;
  bool quad = false;                             // fp32 backward-data: two whole-line quads per lane (below)
  if constexpr (F32IO && DGRAD && !LIN) quad = !p.mask && (p.ncols % BN) == 0;
  float bn_s1[8], bn_s2[8], bn_mx[8];          // bn_mx: max |masked gradient| per channel
  const bool bn_aff = bnf && p.bn_rscale != nullptr;
  float *bnc = reinterpret_cast<float *>(smem) + (BM / PASSES) * LDO + BM;      // [4][BN]: mean, invstd, relu scale, relu shift
  if (bnf) {
    for (int cc = tid; cc < BN; cc += NT) {
      const int colc = ntile * BN + cc;
      const bool ok = colc < p.ncols;
      bnc[cc] = ok ? p.bn_mean[(long long)g * p.ncols + colc] : 0.f;
      bnc[BN + cc] = ok ? p.bn_invstd[(long long)g * p.ncols + colc] : 0.f;
      bnc[2 * BN + cc] = (ok && bn_aff) ? p.bn_rscale[(long long)g * p.ncols + colc] : 0.f;
      bnc[3 * BN + cc] = (ok && bn_aff) ? p.bn_rshift[(long long)g * p.ncols + colc] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) bn_s1[k] = bn_s2[k] = bn_mx[k] = 0.f;
  }
#pragma unroll
  for (int ph = 0; ph < PASSES; ++ph) {
  if (PASSES > 1 && ph > 0) __syncthreads();                  // the previous pass's rows have been read
  if (wm / (WGM / PASSES) == ph) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < NE; ++e)
          ot[((wm % (WGM / PASSES)) * WTM + i * TS + erow(e)) * LDO + wn * WTN + j * TS + li] = acc[i][j][e];
  }
  __syncthreads();
  // (with the reduce fused the iterations run one at a time: unrolled, the compiler keeps four iterations' loads in
  // flight on top of the accumulators and the running sums, and spills)
  auto store_rows = [&](int it) {
    const int v = tid + it * NT;
    const int rl = v / CV, cv = v - rl * CV;
    const int r = ph * PR + rl;
    const int col = ntile * BN + cv * 8;
    const int off = rowoff[r];
    if (off < 0 || col >= p.ncols) return;
    const float4 lo = *reinterpret_cast<const float4 *>(ot + rl * LDO + cv * 8);
    const float4 hi = *reinterpret_cast<const float4 *>(ot + rl * LDO + cv * 8 + 4);
    float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if constexpr (F32IO) {
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] *= osc;
    }
    if (!DGRAD && F32IO && p.scale) {          // inference: BatchNorm folded into a per-channel affine
      const float4 s0 = *reinterpret_cast<const float4 *>(p.scale + col), s1 = *reinterpret_cast<const float4 *>(p.scale + col + 4);
      x[0] *= s0.x; x[1] *= s0.y; x[2] *= s0.z; x[3] *= s0.w; x[4] *= s1.x; x[5] *= s1.y; x[6] *= s1.z; x[7] *= s1.w;
    }
    if (!DGRAD && p.bias) {
      const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + col), b1 = *reinterpret_cast<const float4 *>(p.bias + col + 4);
      x[0] += b0.x; x[1] += b0.y; x[2] += b0.z; x[3] += b0.w; x[4] += b1.x; x[5] += b1.y; x[6] += b1.z; x[7] += b1.w;
    }
    if (!DGRAD && p.relu && !(F32IO && p.addend)) {       // (forward with a residual: the ReLU follows the add, below)
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = fmaxf(x[k], 0.f);
    }
    if constexpr (F32IO) {
      if (!DGRAD && p.addend) {                // forward residual (fp32, or sp: the previous block's output), then ReLU
        if (p.addend_sp) {                     // (an activation in sp storage: scale 1)
          const uint4 *q = reinterpret_cast<const uint4 *>(p.addend) + (((long long)g * gelems + off + col) >> 3) * SP_NP;
          float r[8];
          merge2_chunk(q[0], q[1], r);
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] += r[k];
        } else {
          const float4 a0 = *reinterpret_cast<const float4 *>(add_f + off + col), a1 = *reinterpret_cast<const float4 *>(add_f + off + col + 4);
          x[0] += a0.x; x[1] += a0.y; x[2] += a0.z; x[3] += a0.w; x[4] += a1.x; x[5] += a1.y; x[6] += a1.z; x[7] += a1.w;
        }
        if (p.relu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = fmaxf(x[k], 0.f);
        }
      }
      if (!DGRAD && p.out_sp) {                // the next conv's operand, written directly (one chunk per lane)
        if constexpr (LIN) {
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] *= lin_scale;
        }
        uint4 q1, q2;
        split2_chunk(x, q1, q2);
        uint4 *dst = reinterpret_cast<uint4 *>(p.out) + (((long long)g * gelems + off + col) >> 3) * SP_NP;
        dst[0] = q1;
        dst[1] = q2;
        return;
      }
      if (DGRAD && mask_f) {
        float mm[8];
        if (p.mask_sp) {                       // the producer's activation in sp storage (a Linear's hidden layer)
          const uint4 *q = reinterpret_cast<const uint4 *>(p.mask) + (((long long)g * gelems + off + col) >> 3) * SP_NP;
          merge2_chunk(q[0], q[1], mm);
        } else {
          const float4 m0 = *reinterpret_cast<const float4 *>(mask_f + off + col), m1 = *reinterpret_cast<const float4 *>(mask_f + off + col + 4);
          mm[0] = m0.x; mm[1] = m0.y; mm[2] = m0.z; mm[3] = m0.w; mm[4] = m1.x; mm[5] = m1.y; mm[6] = m1.z; mm[7] = m1.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = mm[k] > 0.f ? x[k] : 0.f;
      }
      if (DGRAD && add_f) {
        const float4 a0 = *reinterpret_cast<const float4 *>(add_f + off + col), a1 = *reinterpret_cast<const float4 *>(add_f + off + col + 4);
        x[0] += a0.x; x[1] += a0.y; x[2] += a0.z; x[3] += a0.w; x[4] += a1.x; x[5] += a1.y; x[6] += a1.z; x[7] += a1.w;
      }
      if (bnf) {
        const float *yg = p.bn_y + (long long)g * gelems + off + col;
        const float4 y0 = *reinterpret_cast<const float4 *>(yg), y1 = *reinterpret_cast<const float4 *>(yg + 4);
        const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        unsigned bits = 0xFFu;                      // no ReLU on that unit: every element passes
        if (p.bn_bits) {
          const unsigned short two = *reinterpret_cast<const unsigned short *>(p.bn_bits + (((long long)g * gelems + off + col) >> 2));
          bits = (two & 0xFu) | ((two >> 4) & 0xF0u);
        }
        asm volatile("" ::: "memory");              // re-read the constants here (not hoisted into 32 live registers)
        if (bn_aff) {
          float bn_ra[8], bn_rb[8];
          ld8(bnc + 2 * BN + cv * 8, bn_ra);
          ld8(bnc + 3 * BN + cv * 8, bn_rb);
#pragma unroll
          for (int k = 0; k < 8; ++k) bits = (bits & ~(1u << k)) | ((__builtin_fmaf(yy[k], bn_ra[k], bn_rb[k]) > 0.f ? 1u : 0u) << k);
        }
        float bn_mu[8], bn_is[8];
        ld8(bnc + cv * 8, bn_mu);
        ld8(bnc + BN + cv * 8, bn_is);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          x[k] = ((bits >> k) & 1u) != 0u ? x[k] : 0.f;
          bn_mx[k] = fmaxf(bn_mx[k], fabsf(x[k]));
          bn_s1[k] += x[k];
          bn_s2[k] += x[k] * ((yy[k] - bn_mu[k]) * bn_is[k]);
        }
      }
      if constexpr (LIN) {
#pragma unroll
        for (int k = 0; k < 8; ++k) lin_max = fmaxf(lin_max, fabsf(x[k]));
      }
      if (p.nt_out) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v s0 = {x[0], x[1], x[2], x[3]}, s1 = {x[4], x[5], x[6], x[7]};
        __builtin_nontemporal_store(s0, reinterpret_cast<f4v *>(out_f + off + col));
        __builtin_nontemporal_store(s1, reinterpret_cast<f4v *>(out_f + off + col + 4));
      } else {
        *reinterpret_cast<float4 *>(out_f + off + col) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4 *>(out_f + off + col + 4) = make_float4(x[4], x[5], x[6], x[7]);
      }
      return;
    }
    if (mask_g) {
      const u32x4 m = *reinterpret_cast<const u32x4 *>(mask_g + off + col);
      const unsigned mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        x[2 * k] = bf_lo(mm[k]) > 0.f ? x[2 * k] : 0.f;
        x[2 * k + 1] = bf_hi(mm[k]) > 0.f ? x[2 * k + 1] : 0.f;
      }
    }
    if (add_g) {
      const u32x4 a = *reinterpret_cast<const u32x4 *>(add_g + off + col);
      const unsigned aa[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        x[2 * k] += bf_lo(aa[k]);
        x[2 * k + 1] += bf_hi(aa[k]);
      }
    }
    u32x4 o;
    if (bnf) {
      const u32x4 yv = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned short *>(p.bn_y) + (long long)g * gelems + off + col);
      const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w};
      unsigned bits = p.bn_bits ? (unsigned)p.bn_bits[((long long)g * gelems + off + col) >> 3] : 0xFFu;
      asm volatile("" ::: "memory");                // re-read the constants here (not hoisted into 32 live registers)
      if (bn_aff) {
        float bn_ra[8], bn_rb[8];
        ld8(bnc + 2 * BN + cv * 8, bn_ra);
        ld8(bnc + 3 * BN + cv * 8, bn_rb);
        bits = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bits |= (__builtin_fmaf(bf_lo(yw[k]), bn_ra[2 * k], bn_rb[2 * k]) > 0.f ? 1u : 0u) << (2 * k);
          bits |= (__builtin_fmaf(bf_hi(yw[k]), bn_ra[2 * k + 1], bn_rb[2 * k + 1]) > 0.f ? 1u : 0u) << (2 * k + 1);
        }
      }
      // the second sum stays UNCENTRED here - sum(dz * y); bn_bwd_finalize turns the totals into s2 = invstd * (sum(dz * y)
      // - mean * s1) in fp64 - so that this kernel, whose accumulators leave it ~50 registers, needs no constants
      unsigned ow[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float y0 = bf_lo(yw[k]), y1 = bf_hi(yw[k]);
        const bool on0 = ((bits >> (2 * k)) & 1u) != 0u, on1 = ((bits >> (2 * k + 1)) & 1u) != 0u;
        ow[k] = pack_bf2(on0 ? x[2 * k] : 0.f, on1 ? x[2 * k + 1] : 0.f);
        const float d0 = bf_lo(ow[k]), d1 = bf_hi(ow[k]);
        bn_s1[2 * k] += d0;
        bn_s1[2 * k + 1] += d1;
        bn_s2[2 * k] = __builtin_fmaf(d0, y0, bn_s2[2 * k]);
        bn_s2[2 * k + 1] = __builtin_fmaf(d1, y1, bn_s2[2 * k + 1]);
      }
      o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
    } else {
      o.x = pack_bf2(x[0], x[1]);
      o.y = pack_bf2(x[2], x[3]);
      o.z = pack_bf2(x[4], x[5]);
      o.w = pack_bf2(x[6], x[7]);
    }
    *reinterpret_cast<u32x4 *>(out_g + off + col) = o;
  };
  // The plain fp32 result (a training forward's y; backward-data without mask / addend / fused reduce) has nothing per 8 channels
  // to apply: a lane stores ONE 16-byte vector and 32 consecutive lanes cover a row's 512 bytes - whole 128-byte lines per store
  // instruction, where the 8-channel form writes every line as two half-line stores of different instructions (C3: forward
  // family 15.5 -> 14.9 ms, step 79.8 -> 79.1 ms on one box).  The same idea on the fused-reduce path - 4 channels per lane,
  // whole-line y loads - LOST 0.9 ms (twice the iterations), and requesting iteration it + 1's y / bits / addend before iteration it
  // is processed needs 26 spilled registers here and is worth 1.3 % where it fits (the two-workgroup pipelined form): that path
  // is bound by the bytes of y, not by its load round trips.
  if (quad) {
    // fp32 backward-data (with or without the fused reduce): a lane's 8 channels are TWO quads half a tile apart - columns
    // 4 cq .. 4 cq + 3 and BN/2 + 4 cq .. - so each of its two loads / stores per tensor is 16 bytes next to its neighbour
    // lanes': whole 128-byte lines per instruction (y, the addend, dx), same registers and iteration count as 8 adjacent channels
    // (C3: backward-data family 20.9 -> 20.2 ms, step 79.6 -> 79.3 ms on one box)
    constexpr int HB = BN / 2;
    const int cq = tid % CV;
    const int cA = cq * 4, cB = HB + cq * 4;
    const int colA = ntile * BN + cA, colB = ntile * BN + cB;
#pragma unroll 1
    for (int it = 0; it < PR * CV / NT; ++it) {
      const int rl = (tid + it * NT) / CV;
      const int off = rowoff[ph * PR + rl];
      if (off < 0) continue;
      const float4 qa = *reinterpret_cast<const float4 *>(ot + rl * LDO + cA);
      const float4 qb = *reinterpret_cast<const float4 *>(ot + rl * LDO + cB);
      float x[8] = {qa.x * osc, qa.y * osc, qa.z * osc, qa.w * osc, qb.x * osc, qb.y * osc, qb.z * osc, qb.w * osc};
      if (add_f) {
        const float4 a0 = *reinterpret_cast<const float4 *>(add_f + off + colA), a1 = *reinterpret_cast<const float4 *>(add_f + off + colB);
        x[0] += a0.x; x[1] += a0.y; x[2] += a0.z; x[3] += a0.w; x[4] += a1.x; x[5] += a1.y; x[6] += a1.z; x[7] += a1.w;
      }
      if (bnf) {
        const float *yg = p.bn_y + (long long)g * gelems + off;
        const float4 y0 = *reinterpret_cast<const float4 *>(yg + colA), y1 = *reinterpret_cast<const float4 *>(yg + colB);
        const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        unsigned bits = 0xFFu;                      // no ReLU on that unit: every element passes
        if (p.bn_bits) {
          const uint8_t *bb = p.bn_bits + (((long long)g * gelems + off) >> 2);
          bits = (bb[colA >> 2] & 0xFu) | ((bb[colB >> 2] & 0xFu) << 4);
        }
        asm volatile("" ::: "memory");              // re-read the constants here (not hoisted into 32 live registers)
        auto ld44 = [&](const float *base, float (&v)[8]) {
          const float4 u = *reinterpret_cast<const float4 *>(base + cA), w = *reinterpret_cast<const float4 *>(base + cB);
          v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; v[4] = w.x; v[5] = w.y; v[6] = w.z; v[7] = w.w;
        };
        if (bn_aff) {
          float bn_ra[8], bn_rb[8];
          ld44(bnc + 2 * BN, bn_ra);
          ld44(bnc + 3 * BN, bn_rb);
#pragma unroll
          for (int k = 0; k < 8; ++k) bits = (bits & ~(1u << k)) | ((__builtin_fmaf(yy[k], bn_ra[k], bn_rb[k]) > 0.f ? 1u : 0u) << k);
        }
        float bn_mu[8], bn_is[8];
        ld44(bnc, bn_mu);
        ld44(bnc + BN, bn_is);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          x[k] = ((bits >> k) & 1u) != 0u ? x[k] : 0.f;
          bn_mx[k] = fmaxf(bn_mx[k], fabsf(x[k]));
          bn_s1[k] += x[k];
          bn_s2[k] += x[k] * ((yy[k] - bn_mu[k]) * bn_is[k]);
        }
      }
      if (p.nt_out) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v s0 = {x[0], x[1], x[2], x[3]}, s1 = {x[4], x[5], x[6], x[7]};
        __builtin_nontemporal_store(s0, reinterpret_cast<f4v *>(out_f + off + colA));
        __builtin_nontemporal_store(s1, reinterpret_cast<f4v *>(out_f + off + colB));
      } else {
        *reinterpret_cast<float4 *>(out_f + off + colA) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4 *>(out_f + off + colB) = make_float4(x[4], x[5], x[6], x[7]);
      }
    }
    continue;
  }
  bool plain4 = false;
  if constexpr (F32IO && !LIN) plain4 = p.nt_out && !p.scale && !p.bias && !p.addend && !p.mask && !p.out_sp && !bnf && !p.relu;
  if (plain4) {
    constexpr int CV4 = BN / 4;
#pragma unroll
    for (int it = 0; it < PR * CV4 / NT; ++it) {
      const int v = tid + it * NT;
      const int rl = v / CV4, c4 = v - rl * CV4;
      const int col = ntile * BN + c4 * 4;
      const int off = rowoff[ph * PR + rl];
      if (off < 0 || col >= p.ncols) continue;
      const float4 q = *reinterpret_cast<const float4 *>(ot + rl * LDO + c4 * 4);
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v sv = {q.x * osc, q.y * osc, q.z * osc, q.w * osc};
      __builtin_nontemporal_store(sv, reinterpret_cast<f4v *>(out_f + off + col));
    }
  } else if constexpr (BNF) {
#pragma unroll 1
    for (int it = 0; it < PR * CV / NT; ++it) store_rows(it);
  } else {
#pragma unroll
    for (int it = 0; it < PR * CV / NT; ++it) store_rows(it);
  }
  }  // passes
  if constexpr (LIN && F32IO) {
    if (p.out_absmax) {                        // ONE atomic per workgroup (atomics on one address serialise at ~12 ns each)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) lin_max = fmaxf(lin_max, __shfl_xor(lin_max, o, 64));
      __syncthreads();                         // the staging tile has been read out
      float *wred = reinterpret_cast<float *>(smem);
      if (lane == 0) wred[wave] = lin_max;
      __syncthreads();
      if (tid == 0) {
        float mx = wred[0];
        for (int k = 1; k < NT / 64; ++k) mx = fmaxf(mx, wred[k]);
        if (mx > 0.f) atomicMax(p.out_absmax, __float_as_uint(mx));
      }
      __syncthreads();                         // (the fused reduce below reuses the tile)
    }
  }
  if (bnf) {
    constexpr int RL = NT / CV;                // threads (row lanes) per 8-channel column group
    __syncthreads();                           // the staging tile has been read out
    float *red = reinterpret_cast<float *>(smem);           // [3][RL][BN]
    const int cvt = tid % CV, rlt = tid / CV;
    const int nred = p.bn_part_rows;                        // 2: (s1, s2); 3: (s1, s2, max |dz|)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cc = quad ? (k < 4 ? cvt * 4 + k : BN / 2 + cvt * 4 + k - 4) : cvt * 8 + k;     // the thread's k-th channel
      red[(0 * RL + rlt) * BN + cc] = bn_s1[k];
      red[(1 * RL + rlt) * BN + cc] = bn_s2[k];
      if (F32IO) red[(2 * RL + rlt) * BN + cc] = bn_mx[k];       // (bf16 storage: two rows, no maximum)
    }
    __syncthreads();
    for (int idx = tid; idx < nred * BN; idx += NT) {
      const int which = idx / BN, cc = idx - which * BN;
      const int colr = ntile * BN + cc;
      if (colr >= p.ncols) continue;
      float t = 0.f;
      if (which < 2)
        for (int r = 0; r < RL; ++r) t += red[(which * RL + r) * BN + cc];       // fixed order
      else
        for (int r = 0; r < RL; ++r) t = fmaxf(t, red[(which * RL + r) * BN + cc]);
      p.bn_part[(((long long)g * p.bn_parts + c.part0 + mtile) * nred + which) * p.ncols + colr] = t;
    }
  }
}

// Transposing fragment read (ds_read_b64_tr_b16) from a pixel-major [k][m] LDS image with row pitch `ld` elements:
// the A / B fragment of v_mfma_f32_32x32x16_bf16 for rows m0 .. m0+31, k0 .. k0+15 (cdna_hip_programming.md T10).
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short *img, int ld, int k0, int m0, int lane) {
  // lane 4q+p of a 16-lane group supplies the address of block row q (k), columns 4p..4p+3 (m); it receives
  // column (lane & 15) of the 4 rows.  Groups: (lane>>4)&1 -> m half of the 32-row MFMA block, lane>>5 -> k half.
  const int t = lane & 15, q = t >> 2, pq = t & 3;
  const int mh = (lane >> 4) & 1, kh = lane >> 5;
  const unsigned short *a = img + (k0 + 8 * kh + q) * ld + m0 + 16 * mh + 4 * pq;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
  const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * ld));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  return __builtin_bit_cast(bf16x8, z);
}

}  // namespace mvg
