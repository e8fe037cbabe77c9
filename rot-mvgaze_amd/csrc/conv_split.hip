// fp32-accurate convolution / linear kernels on the gfx950 fp16 matrix cores ("split" operands).
//
// An fp32 value a is held as TWO fp16 pieces, a1 = fp16(a), a2 = fp16(a - a1) (round to nearest; a - a1 is exact):
// |a - a1 - a2| <= 2^-23 |a| - at most the last of the 24 significand bits is lost, three values in four are exact, rms
// 0.74 * 2^-24 |a| - provided the value sits inside fp16's exponent range, which a per-tensor power-of-two scale
// arranges (exact to apply and to undo; elem.h: sp_t).
// A product of two pieces is exact in fp32, so
//     a * b  =  a1 b1 + (a1 b2 + a2 b1)  + O(2^-22 |a b|)
// THREE fp16 MFMAs (v_mfma_f32_16x16x32_f16, fp32 accumulation) per 32 k.  Measured against fp64 (DESIGN.md 4a,
// profiles/r03_split_accuracy.txt): the result is as close as the fp32-MFMA kernel's, whose own k-ordered fp32
// accumulation error (1.5-4e-7 relative L2) is larger than what the operand rounding (6e-8) and the dropped a2 b2
// term (4e-8) contribute.  Round 2 used three bf16 pieces and six products ("s3", exact operands, 6 bytes per
// element): this format moves 4 bytes per element and half the MFMA work.
// This is the 1e-4 parity path's arithmetic, NOT the reduced-precision bf16 path of conv_bf16.hip.
//
// Operand format "sp" (written by the producers: bn.hip's apply passes, the weight prep below): channels in chunks
// of 8, the two pieces of a chunk adjacent - element (row, c, piece) at ushort offset
//     ((row * C/8 + c/8) * 2 + piece) * 8 + c % 8            (32 contiguous bytes per 8 channels, 4 bytes/element)
// Scales: activations are stored unscaled (BatchNorm keeps them O(1); fp16 reaches 65504); every gradient tensor dy
// and every weight copy carries a device scalar 2^-k written by its producer (bn_bwd_apply: from a bound on |dy|;
// the weight prep: from max |w|), which the consumer's epilogue multiplies back in.
//
// Kernels: igemm_split16_kernel (fprop and dgrad, uniform-tap shapes with channels % 32 == 0; below) and
// wgrad_split_kernel.  fp32 output through the LDS-staged epilogue of bf16_tile.h (BN statistics partials, addend;
// backward-data can carry the BatchNorm-backward reduce pass of the unit it feeds: mvg_conv_dgrad_split_bnreduce).
#include "bf16_tile.h"
#include "elem.h"

namespace mvg {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// acc[i][j] += the three piece products of fragments av[piece][i], bv[piece][j] (32x32x16 tiles: wgrad); smallest
// terms first: (a1 b2, a2 b1), a1 b1
#define SPLIT_ONE(PA, PB, av, bv, acc)                                                                         \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[PA][i], bv[PB][j], acc[i][j], 0, 0, 0);
#define SPLIT_PRODUCTS(av, bv, acc) SPLIT_ONE(0, 1, av, bv, acc) SPLIT_ONE(1, 0, av, bv, acc) SPLIT_ONE(0, 0, av, bv, acc)

constexpr int SP_BM = 128, SP_BK = 32;

// fp32 [n8 * 8] * scale -> sp (layout plumbing for tests and for tensors no kernel writes in sp directly)
__global__ __launch_bounds__(256) void split_f32_kernel(const float4 *__restrict__ x, uint4 *__restrict__ out, long long n8, float scale) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const float4 lo = x[2 * i], hi = x[2 * i + 1];
    const float v[8] = {lo.x * scale, lo.y * scale, lo.z * scale, lo.w * scale, hi.x * scale, hi.y * scale, hi.z * scale, hi.w * scale};
    uint4 q1, q2;
    split2_chunk(v, q1, q2);
    out[2 * i] = q1;
    out[2 * i + 1] = q2;
  }
}

// sp -> fp32: (piece 1 + piece 2) * inv_scale
__global__ __launch_bounds__(256) void merge_sp_kernel(const uint4 *__restrict__ x, float4 *__restrict__ out, long long n8, float inv_scale) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    float v[8];
    merge2_chunk(x[2 * i], x[2 * i + 1], v);
    out[2 * i] = make_float4(v[0] * inv_scale, v[1] * inv_scale, v[2] * inv_scale, v[3] * inv_scale);
    out[2 * i + 1] = make_float4(v[4] * inv_scale, v[5] * inv_scale, v[6] * inv_scale, v[7] * inv_scale);
  }
}

// A gradient g [rows][cols] of the fusion block on its way into the split kernels, ONE launch: g -> sp times the 2^k that
// max |g| allows (the maximum was left in *absmax - float bits - by g's producer: the LIN epilogue's atomicMax, or
// fuse_unbuild / skinny_bwd_dx), *out_sinv = 2^-k, and db (+)= the column sums of g (the Linear's bias gradient).
// Workgroup = 32 columns (4 chunks of 8) x 64 row lanes; the row lanes are added through LDS in lane order (reproducible).
__global__ __launch_bounds__(256) void split_colsum_kernel(const float *__restrict__ g, int rows, int cols, const unsigned *__restrict__ absmax,
                                                           uint4 *__restrict__ out, float *__restrict__ out_sinv, float *__restrict__ db,
                                                           int accumulate) {
  __shared__ float sh[64][33];
  const float scale = sp_scale_for(__uint_as_float(*absmax));
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_sinv = 1.f / scale;
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c8 = blockIdx.x * 4 + cl, c8n = cols >> 3;
  float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = rl; r < rows; r += 64) {
    const float4 *src = reinterpret_cast<const float4 *>(g + (long long)r * cols + c8 * 8);
    const float4 lo = src[0], hi = src[1];
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sum[k] += v[k];
      w[k] = v[k] * scale;
    }
    uint4 q1, q2;
    split2_chunk(w, q1, q2);
    uint4 *dst = out + SP_NP * ((long long)r * c8n + c8);
    dst[0] = q1;
    dst[1] = q2;
  }
  if (db == nullptr) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[rl][cl * 8 + k] = sum[k];
  __syncthreads();
  if (threadIdx.x < 32) {
    const int col = blockIdx.x * 32 + threadIdx.x;
    float t = accumulate ? db[col] : 0.f;
    for (int r = 0; r < 64; ++r) t += sh[r][threadIdx.x];
    db[col] = t;
  }
}

// Every conv's weight copies of one training step in TWO launches (grid.y = conv): (1) max |w| per conv, (2) the
// copies.  mode 1: fp32 KRSC -> sp KRSC (+ sp CRSK), both scaled by 2^k with max |w| * 2^k just below 2^15, and
// wstat[conv] = {max |w| bits, 2^-k} for the consumers' epilogues; mode 0: -> bf16 KRSC (cin zero-padded to cin_pad)
// (+ bf16 CRSK), conv_bf16.hip's layouts (no scale).
struct WPrepItem {
  const float *w;
  void *wk, *wt;
  int cout, rs, cin, cin_pad;
  float *stat;               // mode 1: [2] device floats {max |w| as uint bits (zero before the absmax launch), 2^-k}
};

// one record -> device memory, its stat pair cleared (mvg_split_weights: the single-conv form of the batched prep)
__global__ void wprep_stage_item_kernel(WPrepItem it, WPrepItem *__restrict__ dst) {
  if (threadIdx.x == 0) *dst = it;
  if (threadIdx.x < 2) it.stat[threadIdx.x] = 0.f;
}

__global__ __launch_bounds__(256) void weights_absmax_kernel(const WPrepItem *__restrict__ items) {
  const WPrepItem it = items[blockIdx.y];
  const long long n4 = (long long)it.cout * it.rs * it.cin / 4;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4 *>(it.w)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned *>(it.stat), __float_as_uint(m));
}

__global__ __launch_bounds__(256) void weights_prep_batch_kernel(const WPrepItem *__restrict__ items, int mode) {
  __shared__ float tile[64][65];             // 1x1 / Linear weights: the transposed copy goes through LDS
  const WPrepItem it = items[blockIdx.y];
  const float *__restrict__ w = it.w;
  const int cout = it.cout, rs = it.rs, cin = it.cin;
  if (mode == 1) {
    const float scale = sp_scale_for(__uint_as_float(*reinterpret_cast<const unsigned *>(it.stat)));
    if (blockIdx.x == 0 && threadIdx.x == 0) it.stat[1] = 1.f / scale;
    uint4 *wk = reinterpret_cast<uint4 *>(it.wk), *wt = reinterpret_cast<uint4 *>(it.wt);
    const int c8n = cin / 8, o8n = cout / 8;
    // 1x1 weights with 64-divisible sides (the fusion block's 3584 x 3584 Linears: 51 MB each): the CRSK copy as a tiled
    // transpose - rows read as float4, 8-output-channel chunks written 256 contiguous bytes per input channel.  (The
    // gather below reads 8 floats at a stride of cin per chunk: fine for the backbone's convs, 1 ms per Linear.)
    const bool tiled = wt != nullptr && rs == 1 && (cin % 64) == 0 && (cout % 64) == 0;
    const long long nk = (long long)cout * rs * c8n, nt = (wt && !tiled) ? (long long)cin * rs * o8n : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nk + nt; i += (long long)gridDim.x * 256) {
      float v[8];
      uint4 *dst;
      if (i < nk) {
        const float4 *src = reinterpret_cast<const float4 *>(w + i * 8);          // (o, tap, c8) is the KRSC order itself
        const float4 lo = src[0], hi = src[1];
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        dst = wk + SP_NP * i;
      } else {
        const long long j = i - nk;
        const int o8 = (int)(j % o8n);
        const long long t = j / o8n;
        const int tap = (int)(t % rs), c = (int)(t / rs);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = w[((long long)(o8 * 8 + k) * rs + tap) * cin + c];
        dst = wt + SP_NP * j;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= scale;
      uint4 q1, q2;
      split2_chunk(v, q1, q2);
      dst[0] = q1;
      dst[1] = q2;
    }
    if (tiled) {
      const int tc = cin / 64, ntiles = (cout / 64) * tc;
      for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int o0 = (t / tc) * 64, c0 = (t % tc) * 64;
        __syncthreads();                                       // the previous tile has been read out
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int r = (threadIdx.x >> 4) + 16 * ps, cq = threadIdx.x & 15;
          const float4 x = *reinterpret_cast<const float4 *>(w + (long long)(o0 + r) * cin + c0 + 4 * cq);
          tile[r][4 * cq] = x.x;
          tile[r][4 * cq + 1] = x.y;
          tile[r][4 * cq + 2] = x.z;
          tile[r][4 * cq + 3] = x.w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int idx = threadIdx.x + 256 * q, cl = idx >> 3, o8 = idx & 7;
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = tile[o8 * 8 + k][cl] * scale;
          uint4 q1, q2;
          split2_chunk(v, q1, q2);
          uint4 *dst = wt + SP_NP * ((long long)(c0 + cl) * o8n + (o0 >> 3) + o8);
          dst[0] = q1;
          dst[1] = q2;
        }
      }
    }
    return;
  }
  unsigned short *wk = reinterpret_cast<unsigned short *>(it.wk), *wt = reinterpret_cast<unsigned short *>(it.wt);
  const int cin_pad = it.cin_pad;
  const long long total = (long long)cout * rs * cin_pad;
  if (rs == 1 && cin == cin_pad && (cin % 64) == 0 && (cout % 64) == 0) {
    // Linear weights (the fusion block in the bf16 path: 3584 x 3584 and the like): 16-byte accesses for the plain copy,
    // the transposed copy as a tiled transpose (the per-element form below writes it 2 bytes at a stride of cout)
    uint4 *wk4 = reinterpret_cast<uint4 *>(wk);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total / 8; i += (long long)gridDim.x * 256) {
      const float4 lo = reinterpret_cast<const float4 *>(w + i * 8)[0], hi = reinterpret_cast<const float4 *>(w + i * 8)[1];
      wk4[i] = make_uint4(bf16_pack2(lo.x, lo.y), bf16_pack2(lo.z, lo.w), bf16_pack2(hi.x, hi.y), bf16_pack2(hi.z, hi.w));
    }
    if (wt) {
      const int tc = cin / 64, ntiles = (cout / 64) * tc;
      for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int o0 = (t / tc) * 64, c0 = (t % tc) * 64;
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int r = (threadIdx.x >> 4) + 16 * ps, cq = threadIdx.x & 15;
          const float4 x = *reinterpret_cast<const float4 *>(w + (long long)(o0 + r) * cin + c0 + 4 * cq);
          tile[r][4 * cq] = x.x;
          tile[r][4 * cq + 1] = x.y;
          tile[r][4 * cq + 2] = x.z;
          tile[r][4 * cq + 3] = x.w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int idx = threadIdx.x + 256 * q, cl = idx >> 3, o8 = idx & 7;
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = tile[o8 * 8 + k][cl];
          *reinterpret_cast<uint4 *>(wt + (long long)(c0 + cl) * cout + o0 + o8 * 8) =
              make_uint4(bf16_pack2(v[0], v[1]), bf16_pack2(v[2], v[3]), bf16_pack2(v[4], v[5]), bf16_pack2(v[6], v[7]));
        }
      }
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cin_pad);
    const long long t = i / cin_pad;
    const int tap = (int)(t % rs), o = (int)(t / rs);
    const float v = c < cin ? w[((long long)o * rs + tap) * cin + c] : 0.f;
    const __bf16 b = (__bf16)v;
    const unsigned short u = __builtin_bit_cast(unsigned short, b);
    wk[i] = u;
    if (wt) wt[((long long)c * rs + tap) * cout + o] = u;
  }
}

// ------------------------------------------------------------------------------------------
// igemm_split16_kernel: 128 x BN x 32 tile, 4 waves as 2 x 2 (wave tile 64 x BN/2), v_mfma_f32_16x16x32_f16, a
// "span" LDS-DMA loader, ONE stage of (128 + BN) x 128 bytes (32 KB) and three (backward-data) or four (forward)
// workgroups per CU, which cover each other's DMA waits and epilogues.
//
// What scripts/kloop_probe.hip measured (a GEMM sandbox with this tile and a 3x3 conv's operand reuse; MI355X,
// profiles/r03_kloop_probe*.txt), step by step from round 2's kernel (three bf16 pieces, 32x32x16 MFMAs,
// piece-major LDS image) at K = 2304 / 1024:
//   * span loader: 175 -> 196, 138 -> 164 TF/s.  One LDS-DMA wave-instruction used to fetch, for 16 rows, the four
//     16-byte chunks of ONE piece - 64 segments of 16 bytes at a 48-byte stride, the pieces' instructions re-touching
//     the same 128-byte lines.  Now an instruction covers 64 CONSECUTIVE 16-byte slots of a row-major LDS image whose
//     rows are the contiguous bytes a row contributes to a K-step (4 chunks x 2 pieces = 128 bytes): 8 whole rows.
//     No padding: the slot of (chunk cc, piece pc) in row R is (2 cc + pc) ^ h(R), h(R) = ((R >> 1) & 1) |
//     (((R >> 2) & 1) << 2), filled by permuting the SOURCE address inside the row's span (the DMA destination is
//     linear in the lane); a ds_read_b128 of a 16x16x32 fragment (lane: row l & 15, chunk l >> 4) is conflict-free
//     (brute-force check: DESIGN.md 4a).
//   * 16x16x32 instead of 32x32x16 MFMAs: 196 -> 214, 164 -> 177.  Same cycles per flop and the same LDS reads per
//     K-step, but an MFMA-dense loop is clock-limited by power on this chip and holds a higher clock on this shape
//     (MI355X_MICROARCH.md, DVFS give-back 7).
//   * two fp16 pieces and three products instead of three bf16 pieces and six: 214 -> 366, 177 -> 291.
// Row -> address: every lane needs the base offset and tap-validity mask of FOUR rows (one per A instruction it
// issues); thread r computes row r's once and the lanes pick theirs up through LDS.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SPLIT16_ONE(PA, PB)                                                                       \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[PA][i], bv[PB][j], acc[i][j], 0, 0, 0);

__device__ __forceinline__ int sp_row_swz(int R) { return ((R >> 1) & 1) | (((R >> 2) & 1) << 2); }

// WGM = 4 (BN = 64 only): a 256 x 64 tile as four wave rows of 64 x 64 - the same accumulators and fragment reuse per wave as
// the 128 x 128 tile - for the layers with 64 GEMM columns (64-channel 3x3 convs), where the 128 x 64 tile's 64 x 32 wave tiles
// re-read every A fragment for half the products.
//
// STAGES = 2 (launches that leave a CU one or two workgroups - the fusion block's Linears, 48 - 336 tiles of up to 112 K-steps on
// 256 CUs; every conv of a small batch - so that nobody covers a workgroup's waits): an explicit software pipeline, see the K loop.
template <int BN, bool DGRAD, bool LIN = false, int WGM = 2, int STAGES = 1>
__global__ __launch_bounds__(256, STAGES == 2 ? 2 : (DGRAD || WGM == 4) ? 3 : 4) void igemm_split16_kernel(IgemmParams p) {
  constexpr int BM = 64 * WGM, WGN = 4 / WGM, NW = 4;
  static_assert(WGM == 2 || (WGM == 4 && BN == 64), "tiles: 128 x BN (2 x 2 waves) or 256 x 64 (4 x 1)");
  static_assert(STAGES == 1 || STAGES == 2, "one LDS stage, or the two-stage pipeline");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int SLOTS = 4 * SP_NP;                            // 16-byte slots per LDS row (8)
  constexpr int ROWB = 16 * SLOTS, ROWS = BM + BN, STAGE_B = ROWS * ROWB;
  constexpr int NQ = ROWS * SLOTS / 64;                       // DMA wave-instructions per stage (32 / 24)
  constexpr int QA = BM * SLOTS / 64;                         // ... of which the first 16 fill the A rows
  constexpr int A_PER = QA / NW, B_PER = (NQ - QA) / NW;      // per wave: 4 and 4 / 2
  static_assert(QA % NW == 0 && (NQ - QA) % NW == 0, "whole instructions per wave");
  constexpr int EPI_B = bf16_epilogue_bytes<BM, BN, WGM, DGRAD>();          // one wave row (64 tile rows) per staging pass
  constexpr int INFO_OFF = STAGES * STAGE_B > EPI_B ? STAGES * STAGE_B : EPI_B;  // row table behind the stages / the epilogue tile
  constexpr int SMEM_B = INFO_OFF + BM * 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_B];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int nwg = gridDim.x;
  const int wg_all = p.no_remap ? (int)blockIdx.x : xcd_remap(blockIdx.x, nwg);
  int ci = 0;
  for (int i = 1; i < p.ncls; ++i) ci += wg_all >= p.cls[i].tile0;
  const IgemmClass &c = p.cls[ci];
  const int wg = wg_all - c.tile0;
  const int ntile = wg % p.ntiles;
  const int mt_all = wg / p.ntiles;
  const int g = mt_all / c.mtiles_per_group;
  const int mtile = mt_all - g * c.mtiles_per_group;
  const int KT = c.KT;
  const int ohw = c.out_h * c.out_w;

  // ---- row table: thread r < 128 -> (byte offset of row r's pixel at tap (0,0) channel 0, bit t = tap t in range)
  uint2 *rowinfo = reinterpret_cast<uint2 *>(smem + INFO_OFF);
  if (tid < BM) {
    const long long m = (long long)mtile * BM + tid;
    const bool ok = m < c.rows_per_group;
    const int mm = ok ? (int)m : 0;
    const int img = (int)fdiv((unsigned)mm, c.ohw_div);
    const int rem = mm - img * ohw;
    const int oy = (int)fdiv((unsigned)rem, c.ow_div), ox = rem - oy * c.out_w;
    const int y0 = DGRAD ? oy + c.cls_cy : oy * p.stride - p.pad;
    const int x0 = DGRAD ? ox + c.cls_cx : ox * p.stride_w - p.pad_w;
    const unsigned base = (unsigned)(img * p.src_img_stride * SP_BYTES) + (unsigned)((y0 * p.src_w + x0) * p.src_c) * (unsigned)SP_BYTES;
    unsigned msk = 0;
    for (int t = 0; t < c.ntaps; ++t) {
      const int fr = (int)fdiv((unsigned)t, c.tap_ns_div), fs = t - fr * c.tap_ns;
      const int iy = DGRAD ? y0 - fr : y0 + fr;
      const int ix = DGRAD ? x0 - fs : x0 + fs;
      msk |= (unsigned)(((unsigned)iy < (unsigned)p.src_h) & ((unsigned)ix < (unsigned)p.src_w)) << t;
    }
    rowinfo[tid] = make_uint2(base, ok ? msk : 0u);
  }
  __syncthreads();
  // ---- the instructions this wave issues: Q = wave + 4 i; lane -> linear slot 64 Q + lane -> (row, slot in row);
  // the slot holds source slot j = slot ^ h(row) of the row's 128-byte span (j = 2 cc + pc: the memory order)
  unsigned a_base[A_PER], a_vmask[A_PER], b_base[B_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int sl = (wave + NW * i) * 64 + lane;
    const int row = sl / SLOTS, j = (sl % SLOTS) ^ sp_row_swz(row);
    const uint2 ri = rowinfo[row];
    a_base[i] = ri.x + 16u * (unsigned)j;
    a_vmask[i] = ri.y;
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int sl = (wave + NW * (A_PER + i)) * 64 + lane - BM * SLOTS;
    const int row = sl / SLOTS, j = (sl % SLOTS) ^ sp_row_swz(row);        // h(BM + row) == h(row): BM is a multiple of 8
    const int n = ntile * BN + row;
    b_base[i] = pred_off(((unsigned)n * (unsigned)p.b_row_len) * (unsigned)SP_BYTES + 16u * (unsigned)j, n < p.ncols);
  }
  const __amdgpu_buffer_rsrc_t rs_a =
      make_rsrc(reinterpret_cast<const char *>(p.a) + (long long)g * p.imgs_per_group * p.src_img_stride * SP_BYTES, p.a_group_bytes);
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.b, p.b_bytes);
  typedef __attribute__((address_space(3))) void *lds_vp;

  auto issue = [&](int kt, int stage_off) {
    int kstart = kt * SP_BK;
    if (c.korder) {
      const int cblk = (int)fdiv((unsigned)kt, c.per_div), rem = kt - cblk * c.ntaps;
      kstart = (rem << p.src_c_shift) + cblk * SP_BK;
    }
    const int ks = __builtin_amdgcn_readfirstlane(kstart);
    const int tap_u = c.ntaps > 1 ? (ks >> p.src_c_shift) : 0;
    const int chb = ks - (tap_u << p.src_c_shift);
    const int fru = (int)fdiv((unsigned)tap_u, c.tap_ns_div), fsu = tap_u - fru * c.tap_ns;
    const int disp = (fru * p.src_w + fsu) * p.src_c;
    const unsigned sdelta = (unsigned)(((DGRAD ? -disp : disp) + chb) * SP_BYTES);
    unsigned kb = (unsigned)ks * (unsigned)SP_BYTES;
    if (DGRAD) {
      const int btap = (c.tap_r0 + p.tap_step * fru) * p.s + c.tap_s0 + p.tap_step * fsu;
      kb = (unsigned)(btap * p.src_c + chb) * (unsigned)SP_BYTES;
    }
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const bool ok = ((a_vmask[i] >> tap_u) & 1u) != 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_vp)(smem + stage_off + (wave + NW * i) * 1024), 16, (int)pred_off(a_base[i] + sdelta, ok), 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_vp)(smem + stage_off + (wave + NW * (A_PER + i)) * 1024), 16, (int)(b_base[i] + kb), 0, 0, 0);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // fragment addresses: lane -> row (l & 15) of a 16-row tile, chunk cc = l >> 4, piece pc -> slot (2 cc + pc) ^ h(row)
  const int cc_l = lane >> 4;
  int a_off[TM][SP_NP], b_off[TN][SP_NP];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int R = wm * WTM + i * 16 + (lane & 15);
#pragma unroll
    for (int pc = 0; pc < SP_NP; ++pc) a_off[i][pc] = R * ROWB + (((2 * cc_l + pc) ^ sp_row_swz(R)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int R = wn * WTN + j * 16 + (lane & 15);
#pragma unroll
    for (int pc = 0; pc < SP_NP; ++pc) b_off[j][pc] = (BM + R) * ROWB + (((2 * cc_l + pc) ^ sp_row_swz(R)) << 4);
  }

  auto load_frags = [&](const unsigned char *stage, f16x8 (&av)[SP_NP][TM], f16x8 (&bv)[SP_NP][TN]) {
#pragma unroll
    for (int pc = 0; pc < SP_NP; ++pc) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[pc][i] = *reinterpret_cast<const f16x8 *>(stage + a_off[i][pc]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[pc][j] = *reinterpret_cast<const f16x8 *>(stage + b_off[j][pc]);
    }
  };
  // smallest terms first: (a1 b2, a2 b1), a1 b1
  auto products = [&](const f16x8 (&av)[SP_NP][TM], const f16x8 (&bv)[SP_NP][TN]) { SPLIT16_ONE(0, 1) SPLIT16_ONE(1, 0) SPLIT16_ONE(0, 0) };
  if constexpr (STAGES == 1) {
    for (int kt = 0; kt < KT; ++kt) {
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      {
        f16x8 av[SP_NP][TM], bv[SP_NP][TN];
        load_frags(smem, av, bv);
        products(av, bv);
      }
      __syncthreads();                                         // everyone is done reading before the next DMA lands
    }
  } else {
    // Software pipeline, two levels: two K-steps of DMA in LDS / in flight, and the fragments of K-step kt + 1 read into a second
    // register set while K-step kt's MFMAs run (with one wave per SIMD nothing else overlaps the LDS reads with the matrix
    // pipe).  At the top of K-step kt: wait until K-step kt + 1 has landed (the only group in flight: vmcnt(0)) and this wave's
    // fragment reads of K-step kt are complete (lgkmcnt(0)), barrier - now K-step kt's stage is free for every wave and K-step
    // kt + 2 goes into it.  A bare s_barrier: __syncthreads() carries a fence that the compiler turns into s_waitcnt vmcnt(0)
    // wherever an LDS-DMA is pending, which in the prologue would wait for BOTH stages before the first multiply.  (Three and
    // four stages with counted vmcnt waits - one workgroup per CU - measured the same as two: profiles/r04_lin_kloop_stages_ab.txt.)
    constexpr int G = A_PER + B_PER;                           // DMA instructions per wave and K-step
    if (KT > 0) issue(0, 0);                                   // (KT = 0: a tap-less class of a fused-reduce launch, epilogue only)
    if (KT > 1) {
      issue(1, STAGE_B);
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G) : "memory");     // K-step 0 has landed, K-step 1 is in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    f16x8 a0[SP_NP][TM], b0[SP_NP][TN], a1[SP_NP][TM], b1[SP_NP][TN];
    load_frags(smem, a0, b0);
    int cur = 0;                                               // byte offset of K-step kt's stage
    auto step = [&](int kt, const f16x8 (&ca)[SP_NP][TM], const f16x8 (&cb)[SP_NP][TN], f16x8 (&na)[SP_NP][TM], f16x8 (&nb)[SP_NP][TN]) {
      if (kt + 1 < KT) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < KT) issue(kt + 2, cur);
        cur ^= STAGE_B;
        load_frags(smem + cur, na, nb);
      }
      products(ca, cb);
    };
    for (int kt = 0; kt < KT; kt += 2) {
      step(kt, a0, b0, a1, b1);
      if (kt + 1 < KT) step(kt + 1, a1, b1, a0, b0);
    }
    __syncthreads();                                           // the epilogue reuses the stages
  }
  bf16_epilogue<BM, BN, WGM, DGRAD, true, WGM, true, DGRAD, LIN, WGN>(p, c, acc, reinterpret_cast<unsigned short *>(smem), tid, g, mtile, ntile);
}

// ------------------------------------------------------------------------------------------
// wgrad: dw[o][tap][c] = sum over pixels of dy[pix][o] * x[pix at tap][c] with both operands in sp.  Like the bf16
// kernel (conv_bf16.hip): M = cout, N = (tap, c), K = pixels split into slabs; the LDS images stay pixel-major
// [piece][k][m] (rows padded by 32 elements) and the fragments come from ds_read_b64_tr_b16 (v_mfma_f32_32x32x16_f16,
// three products per 16 pixels).  A K-step is 16 pixels: 20 KB per stage, two stages; a thread's two piece vectors of
// one 8-channel chunk are 32 contiguous bytes in memory.  The result is multiplied by dy's 2^-k (p.dy_sinv).
// Round 4 measured the forward kernel's recipe here too (LDS-DMA loader into a row-contiguous image, ds_read_b64_tr_b16
// fragments, v_mfma_f32_16x16x32_f16, K-step 32; commit history + profiles/r04_wgrad_dma_*_ab.txt): 19.1 ms per C3 step with
// one 32 KB stage at three workgroups per CU, 29.4 ms with two stages at two per CU, against this kernel's 15.2-15.9 ms -
// K here is the pixel axis (hundreds of steps, both operands streamed once), and the register-staged two-stage pipeline
// with a 16-pixel step keeps more loads in flight per CU than a DMA stage that must drain before it is multiplied.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, bool INCR>
__global__ __launch_bounds__(256, BN >= 256 ? 2 : 3) void wgrad_split_kernel(WgradParams p) {
  constexpr int BK = 16, WGM = 2, WGN = 2;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + 32, LDB = BN + 32;
  constexpr int MV = BM / 8, NVB = BN / 8;
  constexpr int A_TPR = 256 / BK;                       // threads per k-row (16)
  constexpr int A_CPT = MV / A_TPR > 0 ? MV / A_TPR : 1;   // chunks per thread
  constexpr int B_CPT = (NVB + A_TPR - 1) / A_TPR;         // (192 columns: 24 chunks on 16 lanes - the second round half idle)
  constexpr int A_ELEMS = BK * LDA, B_ELEMS = BK * LDB;    // one piece
  constexpr int STAGE = SP_NP * (A_ELEMS + B_ELEMS);
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles = p.mtiles * p.ntiles;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int split = logical / tiles;
  const int tile = logical - split * tiles;
  const int ntile = tile % p.ntiles, mtile = tile / p.ntiles;
  const long long m_begin = (long long)split * p.pixels_per_split;
  long long m_end = m_begin + p.pixels_per_split;
  if (m_end > p.pixels) m_end = p.pixels;
  const int m_count = m_end > m_begin ? (int)(m_end - m_begin) : 0;
  const int ohw = p.ho * p.wo;
  const long long img0 = m_begin / ohw;
  const unsigned rem0 = (unsigned)(m_begin - img0 * ohw);

  const char *dy = reinterpret_cast<const char *>(p.dy);
  const char *x = reinterpret_cast<const char *>(p.x);
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(dy + m_begin * p.cout * SP_BYTES, (long long)SP_BYTES * m_count * p.cout);
  const long long x_img_elems = (long long)p.h * p.w * p.cin;
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(x + img0 * x_img_elems * SP_BYTES, p.x_bytes - (long long)SP_BYTES * img0 * x_img_elems);
  // both loaders: pixel row i_row = tid / 16, chunk lane i_v0 = tid % 16 (+ 16 per extra chunk)
  const int i_row = tid / A_TPR, i_v0 = tid % A_TPR;
  unsigned a_off[A_CPT];
  bool a_on[A_CPT];
#pragma unroll
  for (int j = 0; j < A_CPT; ++j) {
    const int cv = i_v0 + j * A_TPR;
    const int col = mtile * BM + cv * 8;
    a_on[j] = cv < MV;
    a_off[j] = (a_on[j] && col < p.cout) ? (unsigned)(i_row * p.cout + col) * (unsigned)SP_BYTES : 0x80000000u;
  }
  int i_dy[B_CPT], i_dx[B_CPT];
  unsigned i_tconst[B_CPT];
  bool i_cok[B_CPT], b_on[B_CPT];
#pragma unroll
  for (int j = 0; j < B_CPT; ++j) {
    const int cv = i_v0 + j * A_TPR;
    const int col = ntile * BN + cv * 8;
    b_on[j] = cv < NVB;
    i_cok[j] = b_on[j] && col < p.ncols;
    const int tap = (int)fdiv((unsigned)(i_cok[j] ? col : 0), p.cin_div);
    const int cc = (i_cok[j] ? col : 0) - tap * p.cin;
    const int fr = (int)fdiv((unsigned)tap, p.s_div), fs = tap - fr * p.s;
    i_dy[j] = fr - p.pad;
    i_dx[j] = fs - p.pad_w;
    i_tconst[j] = (unsigned)((i_dy[j] * p.w + i_dx[j]) * p.cin + cc) * (unsigned)SP_BYTES;
  }
  const unsigned row_bytes = (unsigned)(p.stride * p.w * p.cin) * (unsigned)SP_BYTES, col_bytes = (unsigned)(p.stride_w * p.cin) * (unsigned)SP_BYTES;
  const unsigned img_bytes = (unsigned)x_img_elems * (unsigned)SP_BYTES;
  int s_oy = 0, s_ox = 0;
  unsigned s_imgoff = 0;
  if (INCR) {
    const unsigned pix = rem0 + (unsigned)i_row;
    const unsigned img = fdiv(pix, p.ohw_div);
    const unsigned rem = pix - img * (unsigned)ohw;
    const unsigned oy = fdiv(rem, p.wo_div);
    s_oy = (int)oy;
    s_ox = (int)(rem - oy * (unsigned)p.wo);
    s_imgoff = img * img_bytes;
  }
  u32x4 a_reg[A_CPT][SP_NP], b_reg[B_CPT][SP_NP];
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int j = 0; j < A_CPT; ++j) {
#pragma unroll
      for (int pc = 0; pc < SP_NP; ++pc)             // rows >= m_count: beyond the descriptor = zeros
        a_reg[j][pc] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_off[j] + 16u * pc, 0, 0);
      a_off[j] += (unsigned)(BK * p.cout) * (unsigned)SP_BYTES;
    }
    const bool mok = kt * BK + i_row < m_count;
    int oy, ox;
    unsigned imgoff;
    if constexpr (INCR) {
      oy = s_oy;
      ox = s_ox;
      imgoff = s_imgoff;
    } else {
      const unsigned pix = rem0 + (unsigned)(kt * BK + i_row);
      const unsigned img = fdiv(pix, p.ohw_div);
      const unsigned rem = pix - img * (unsigned)ohw;
      const unsigned uy = fdiv(rem, p.wo_div);
      oy = (int)uy;
      ox = (int)(rem - uy * (unsigned)p.wo);
      imgoff = img * img_bytes;
    }
    const int iy0 = oy * p.stride, ix0 = ox * p.stride_w;
    const unsigned pixoff = imgoff + (unsigned)oy * row_bytes + (unsigned)ox * col_bytes;
#pragma unroll
    for (int j = 0; j < B_CPT; ++j) {
      const bool ok = mok & i_cok[j] & ((unsigned)(iy0 + i_dy[j]) < (unsigned)p.h) & ((unsigned)(ix0 + i_dx[j]) < (unsigned)p.w);
#pragma unroll
      for (int pc = 0; pc < SP_NP; ++pc)
        b_reg[j][pc] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, pred_off(pixoff + i_tconst[j] + 16u * pc, ok), 0, 0);
    }
    if constexpr (INCR) {
      // advance BK pixels: columns wrap into rows, rows into the next image (at most once: ho*wo >= 2*BK)
      unsigned nx = (unsigned)s_ox + BK;
      const unsigned q = fdiv(nx, p.wo_div);
      nx -= q * (unsigned)p.wo;
      s_ox = (int)nx;
      const int noy = s_oy + (int)q;
      const bool wrap = noy >= p.ho;
      s_oy = wrap ? noy - p.ho : noy;
      s_imgoff += wrap ? img_bytes : 0u;
    }
  };
  auto store_tiles = [&](int buf) {
    unsigned short *As = smem + buf * STAGE;
    unsigned short *Bs = As + SP_NP * A_ELEMS;
#pragma unroll
    for (int j = 0; j < A_CPT; ++j)
      if (a_on[j]) {
#pragma unroll
        for (int pc = 0; pc < SP_NP; ++pc)
          *reinterpret_cast<u32x4 *>(As + pc * A_ELEMS + i_row * LDA + (i_v0 + j * A_TPR) * 8) = a_reg[j][pc];
      }
#pragma unroll
    for (int j = 0; j < B_CPT; ++j)
      if (b_on[j]) {
#pragma unroll
        for (int pc = 0; pc < SP_NP; ++pc)
          *reinterpret_cast<u32x4 *>(Bs + pc * B_ELEMS + i_row * LDB + (i_v0 + j * A_TPR) * 8) = b_reg[j][pc];
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
    const unsigned short *As = smem + cur * STAGE;
    const unsigned short *Bs = As + SP_NP * A_ELEMS;
    f16x8 av[SP_NP][TM], bv[SP_NP][TN];
#pragma unroll
    for (int pc = 0; pc < SP_NP; ++pc) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[pc][i] = __builtin_bit_cast(f16x8, tr_frag(As + pc * A_ELEMS, LDA, 0, wm * WTM + i * 32, lane));
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[pc][j] = __builtin_bit_cast(f16x8, tr_frag(Bs + pc * B_ELEMS, LDB, 0, wn * WTN + j * 32, lane));
    }
    SPLIT_PRODUCTS(av, bv, acc)
    __syncthreads();
  }

  float *out = p.out + (long long)split * p.cout * p.ncols;
  const float osc = (p.dy_sinv ? *p.dy_sinv : 1.f) * (p.x_sinv ? *p.x_sinv : 1.f);
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
          float v = acc[i][j][e] * osc;
          if (p.accumulate) v += out[off];
          out[off] = v;
        }
      }
    }
}

// The slab sums of SEVERAL weight gradients in one launch (a residual block's 3-4 convs: round 3 ran one 10 us reduce
// launch behind every wgrad launch, 63 per ResNet-50 step).  Same arithmetic and order as wgrad_reduce_kernel per item.
struct ReduceItem {
  const float *slabs;
  float *dw;
  long long n4;
  int splits, accumulate, lanes, block0;          // block0: first workgroup of this item
};
struct ReduceBatch {
  ReduceItem it[8];
  int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(ReduceBatch b) {
  __shared__ float4 sh[256];
  int k = 0;
  for (int i = 1; i < b.n; ++i) k += (int)blockIdx.x >= b.it[i].block0;
  const ReduceItem &r = b.it[k];
  const int cols = 256 / r.lanes;
  const int c = threadIdx.x % cols, l = threadIdx.x / cols;
  const long long i = (long long)(blockIdx.x - r.block0) * cols + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < r.n4)
    for (int q = l; q < r.splits; q += r.lanes) {
      const float4 v = reinterpret_cast<const float4 *>(r.slabs)[(long long)q * r.n4 + i];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (l == 0 && i < r.n4) {
    float4 t = r.accumulate ? reinterpret_cast<const float4 *>(r.dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < r.lanes; ++q) {
      const float4 v = sh[q * cols + c];
      t.x += v.x;
      t.y += v.y;
      t.z += v.z;
      t.w += v.w;
    }
    reinterpret_cast<float4 *>(r.dw)[i] = t;
  }
}

// The stem's row-window operand (see mvg_stem_fprop_split): one thread per 8-value chunk = image columns (2 ox - 4 + 2 q,
// + 1) x 4 stored channels of window (n, y, ox), written as the chunk's two fp16 pieces.
__global__ __launch_bounds__(256) void stem_rowwindow_kernel(const float4 *__restrict__ x, uint4 *__restrict__ xw, long long n, int h, int w) {
  const int wo = w >> 1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int q = (int)(i & 3);
    const long long t = i >> 2;
    const int ox = (int)(t % wo);
    const long long row = t / wo;                     // image * h + y
    const int c0 = 2 * ox - 4 + 2 * q;                // even: both columns in range or both out (w is even)
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 >= 0 && c0 < w) {
      const float4 a = x[row * w + c0], b = x[row * w + c0 + 1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    uint4 q1, q2;
    split2_chunk(v, q1, q2);
    xw[SP_NP * i] = q1;
    xw[SP_NP * i + 1] = q2;
  }
}

// ... and straight from the module's NCHW fp32 input (3 planes), skipping the NHWC4 image
__global__ __launch_bounds__(256) void stem_rowwindow_nchw_kernel(const float *__restrict__ x, uint4 *__restrict__ xw, long long n, int h, int w) {
  const int wo = w >> 1;
  const long long plane = (long long)h * w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int q = (int)(i & 3);
    const long long t = i >> 2;
    const int ox = (int)(t % wo);
    const long long row = t / wo;                     // image * h + y
    const long long img = row / h;
    const int yy = (int)(row - img * h);
    const int c0 = 2 * ox - 4 + 2 * q;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 >= 0 && c0 < w) {
      const float *src = x + img * 3 * plane + (long long)yy * w + c0;
      const float2 r = *reinterpret_cast<const float2 *>(src), g = *reinterpret_cast<const float2 *>(src + plane),
                   b = *reinterpret_cast<const float2 *>(src + 2 * plane);
      v[0] = r.x; v[1] = g.x; v[2] = b.x; v[4] = r.y; v[5] = g.y; v[6] = b.y;
    }
    uint4 q1, q2;
    split2_chunk(v, q1, q2);
    xw[SP_NP * i] = q1;
    xw[SP_NP * i + 1] = q2;
  }
}

// A stride-2 parity class without taps: dx = addend (or zero) on that class's pixels (float4 vectors)
__global__ __launch_bounds__(256) void dgrad_empty_class_split_kernel(float4 *__restrict__ dx, const float4 *__restrict__ addend,
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
static int validate_split(const mvg_conv_desc *d) {
  if (validate(d)) return 2;
  MVG_REQUIRE(d->cin % 32 == 0 && d->cout % 32 == 0, "split conv: cin and cout must be multiples of 32 (got %d, %d)", d->cin,
              d->cout);
  MVG_REQUIRE(d->r * d->s <= 32, "split conv: at most 32 filter taps");
  return 0;
}

// Row-tile height of a launch: 256 x 64 tiles when the GEMM has fewer than 128 columns, more than one tap (the 64-channel 3x3
// convs: K = 576; the 1x1 layers with 64 columns are HBM-bound either way) and enough rows to fill the chip with the larger
// tiles.  A function of the descriptor alone: mvg_conv_dgrad_bn_partials_split sizes the fused reduce's partials with it.
static int split_tile_rows(int ncols, int taps, long long rows) { return (ncols < 128 && taps > 1 && rows >= 65536) ? 256 : SP_BM; }

template <bool DGRAD>
static int launch_igemm_split(IgemmParams &p, hipStream_t st, bool lin = false, int bm = SP_BM) {
  // (128 x 64 tiles for the short-K, write-heavy 1x1 layers - 64 -> 256 at 56 x 56 and the like - were measured in round 4:
  // within 2 % of 128 x 128 on every such shape, forward and backward-data)
  int bn = p.ncols >= 128 ? 128 : 64;
  if (lin && bn == 128 && p.ncls == 1) {
    // a Linear whose 128 x 128 tiles would leave most CUs idle (C3's head layer: 48 tiles; C4's per-GPU share: 12 - 84):
    // a lone workgroup takes in ~41 GB/s, so the launch is as fast as its busiest CU's operand bytes - 128 x 64 tiles
    // (24 KB instead of 32 KB per K-step) on twice the CUs
    const long long t128 = (long long)p.groups * ceil_div(p.cls[0].rows_per_group, bm) * ceil_div(p.ncols, 128);
    if (2 * t128 <= compute_cus()) bn = 64;
  }
  MVG_REQUIRE(bm == SP_BM || (bm == 256 && bn == 64 && !lin), "split conv: 256-row tiles go with 64 columns");
  p.ntiles = ceil_div(p.ncols, bn);
  p.splits = 1;
  p.sk_tiles = 0;
  long long tiles = 0;
  p.bn_parts = 0;
  for (int i = 0; i < p.ncls; ++i) {
    IgemmClass &c = p.cls[i];
    c.mtiles_per_group = ceil_div(c.rows_per_group, bm);
    c.KT = ceil_div(c.ktotal, SP_BK);              // 0: a class without taps (fused reduce only) - its tiles are epilogue only
    c.korder = c.ntaps > 1 ? 1 : 0;
    c.per_div = make_fastdiv((unsigned)(c.ntaps > 0 ? c.ntaps : 1));
    c.tile0 = (int)tiles;
    c.unit0 = 0;
    c.part0 = p.bn_parts;
    p.bn_parts += c.mtiles_per_group;
    tiles += (long long)p.groups * c.mtiles_per_group * p.ntiles;
    MVG_REQUIRE((c.ntaps >= 1 || (DGRAD && p.bn_part)) && c.ntaps <= 32 && c.ktotal % SP_BK == 0, "split conv: class shape not covered");
  }
  MVG_REQUIRE(tiles < (1LL << 31), "split conv: grid too large");
  if (tiles <= 0) return 0;
  // The backbone's fp32 results (y, dx: 0.2 - 1.7 GB per launch, next read by a BatchNorm pass that streams them once) leave
  // with non-temporal stores: C3 81.4 -> 80.9 ms per step on one box (the passes that follow find more of their other
  // operand in the Infinity Cache: bn_apply 8.8 -> 8.4 ms).  Not the stride-2 parity classes - they write every other pixel,
  // which wants the cache to merge lines (0.96 -> 1.01 ms on 256 -> 512 at 56 x 56) - and not the fusion block's Linears,
  // whose results are re-read at once.
  p.nt_out = (!lin && !(DGRAD && p.cls_step == 2)) ? 1 : 0;
  dim3 grid((unsigned)tiles), block(256);
  // At most two workgroups per CU: nobody covers a workgroup's waits - the two-stage software pipeline.  Measured per shape
  // (scripts/linear_split_bench.py, C3's fusion rows: fprop 431 -> 320 us per iteration, dgrad 246 -> 210;
  // scripts/conv_bench.py 50 32 4, C4's per-GPU share: 15.2 -> 14.6 ms over the net, 512-channel 3x3 at 7x7 0.169 -> 0.127 ms;
  // C3's conv launches all have >= 784 tiles).  Three and four stages - one workgroup per CU - and an L2-blocked tile order
  // changed nothing: a lone workgroup takes in ~41 GB/s whatever it keeps in flight.
  // Up to four per CU it still wins where the K loop is long (ResNet-50's 7x7 stage at C3, 784 tiles: 512-channel 3x3
  // backward-data + reduce 0.399 -> 0.322 ms, 2048 <- 512 backward-data 0.218 -> 0.185); with every slot filled the
  // single-stage loop at four workgroups per CU is faster (17.4 vs 19.5 ms over the forward net).
  int kt_max = 0;
  for (int i = 0; i < p.ncls; ++i) kt_max = p.cls[i].KT > kt_max ? p.cls[i].KT : kt_max;
  const bool pipelined = tiles <= 2LL * compute_cus() || (tiles <= 4LL * compute_cus() && kt_max >= 48);
  if (lin) {                 // a Linear of the fusion block: the epilogue's scale / abs-max features compiled in
    if (bn == 128 && pipelined) hipLaunchKernelGGL((igemm_split16_kernel<128, DGRAD, true, 2, 2>), grid, block, 0, st, p);
    else if (bn == 128) hipLaunchKernelGGL((igemm_split16_kernel<128, DGRAD, true>), grid, block, 0, st, p);
    else if (pipelined) hipLaunchKernelGGL((igemm_split16_kernel<64, DGRAD, true, 2, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_split16_kernel<64, DGRAD, true>), grid, block, 0, st, p);
  } else if (bm == 256) hipLaunchKernelGGL((igemm_split16_kernel<64, DGRAD, false, 4>), grid, block, 0, st, p);
  else if (bn == 128 && pipelined) hipLaunchKernelGGL((igemm_split16_kernel<128, DGRAD, false, 2, 2>), grid, block, 0, st, p);
  else if (bn == 128) hipLaunchKernelGGL((igemm_split16_kernel<128, DGRAD>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((igemm_split16_kernel<64, DGRAD>), grid, block, 0, st, p);
  return check_launch(DGRAD ? "conv_dgrad_split" : "conv_fprop_split");
}

}  // namespace mvg

using namespace mvg;

extern "C" {

int mvg_split_f32(const float *x, void *out_sp, int64_t n, float scale, void *stream) {
  MVG_REQUIRE(x && out_sp && n >= 0 && n % 8 == 0, "split_f32: null argument or n %% 8 != 0");
  MVG_REQUIRE(scale > 0.f, "split_f32: scale must be positive (a power of two keeps the round trip exact)");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, (4.0 + SP_BYTES) * (double)n);
  long long blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float4 *)x, (uint4 *)out_sp, (long long)(n / 8), scale);
  return check_launch("split_f32");
}

int mvg_merge_sp(const void *x_sp, float *out, int64_t n, float inv_scale, void *stream) {
  MVG_REQUIRE(x_sp && out && n >= 0 && n % 8 == 0, "merge_sp: null argument or n %% 8 != 0");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, (4.0 + SP_BYTES) * (double)n);
  long long blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(merge_sp_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint4 *)x_sp, (float4 *)out, (long long)(n / 8), inv_scale);
  return check_launch("merge_sp");
}

int mvg_split_weights(const mvg_conv_desc *d, const float *w, void *w_krsc_sp, void *w_crsk_sp, float *stat2, void *items_dev,
                      void *stream) {
  // one conv through the batched kernels: items_dev = 64 bytes of device memory for the one-record table, stat2 =
  // two device floats that receive {max |w| bits, 2^-k} (the consumer's b_sinv = stat2 + 1)
  MVG_REQUIRE(d && w && w_krsc_sp && stat2 && items_dev, "split_weights: null argument");
  if (validate_split(d)) return 2;
  hipStream_t st = (hipStream_t)stream;
  WPrepItem it;
  memset(&it, 0, sizeof(it));
  it.w = w;
  it.wk = w_krsc_sp;
  it.wt = w_crsk_sp;
  it.cout = d->cout;
  it.rs = d->r * d->s;
  it.cin = d->cin;
  it.cin_pad = d->cin;
  it.stat = stat2;
  static_assert(sizeof(WPrepItem) <= 64, "WPrepItem grew: update the callers' table record size");
  // the record travels as a kernel argument (captured at launch): an asynchronous copy from this stack frame could be read after
  // the frame is gone when the stream is busy
  hipLaunchKernelGGL(wprep_stage_item_kernel, dim3(1), dim3(64), 0, st, it, (WPrepItem *)items_dev);
  if (check_launch("split_weights: staging the table record")) return 1;
  const long long n8 = (long long)d->cout * it.rs * d->cin / 8;
  long long blocks = (n8 + 256 * 8 - 1) / (256 * 8);             // ~8 chunks per thread
  return mvg_weights_prep_batch(items_dev, 1, 1, (int)(blocks < 16 ? 16 : (blocks > 2048 ? 2048 : blocks)), stream);
}

int mvg_weights_prep_batch(const void *items_dev, int n, int mode, int blocks_per_item, void *stream) {
  MVG_REQUIRE(items_dev != nullptr && n > 0 && (mode == 0 || mode == 1) && blocks_per_item >= 0 && blocks_per_item <= 4096,
              "weights_prep_batch: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 0.0);
  // workgroups per record: 64 suits the backbone's convs (53 records, <= 2.4 M weights each); the fusion block's Linears
  // (9 records of up to 12.8 M weights) want a few hundred
  const unsigned bx = blocks_per_item > 0 ? (unsigned)blocks_per_item : 64u;
  if (mode == 1) {           // max |w| per conv first (the records' stat[0] must be zero: the caller clears them)
    hipLaunchKernelGGL(weights_absmax_kernel, dim3(bx > 64 ? bx / 2 : 16, (unsigned)n), dim3(256), 0, st, (const WPrepItem *)items_dev);
    if (check_launch("weights_absmax")) return 1;
  }
  hipLaunchKernelGGL(weights_prep_batch_kernel, dim3(bx, (unsigned)n), dim3(256), 0, st, (const WPrepItem *)items_dev, mode);
  return check_launch("weights_prep_batch");
}

int mvg_conv_stats_partials_split(const mvg_conv_desc *d, int32_t *rows_per_partial) {
  if (validate_split(d)) return -1;
  const long long rows = (long long)d->n * d->ho * d->wo;
  if (rows_per_partial) *rows_per_partial = 64;               // one wave tile of rows
  // two partials per 128-row tile (the last tile's second one may lie beyond the rows: count 0, ignored by bn_finalize)
  return ceil_div(rows, SP_BM) * 2;
}

struct SplitAffine {       // inference forward: y = acc * scale + shift (+ residual) [relu], result fp32 or sp
  const float *scale, *shift;
  const void *residual;
  int residual_sp, relu, out_sp;
  // Linear layers of the fusion block (LIN kernels): see IgemmParams::out_absmax / out_sinv / bias_absmax
  int lin;
  float *out_absmax, *out_sinv;
  const float *bias_absmax;
};

// stride_w / pad_w >= 0: the horizontal stride / padding differ from d->stride / d->pad (the stem's row-window form, whose
// descriptor the caller has checked itself)
static int fprop_split_impl(const mvg_conv_desc *d, const void *x_sp, const float *x_sinv, const void *w_sp, const float *w_sinv,
                            void *y, float *stats, void *stream, const SplitAffine *aff, int stride_w = -1, int pad_w = -1) {
  if (stride_w < 0 && validate_split(d)) return 2;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.a = (const float *)x_sp;
  p.b = (const float *)w_sp;
  p.a_sinv = x_sinv;
  p.b_sinv = w_sinv;
  p.out = (float *)y;
  p.stats = stats;
  if (aff) {
    p.scale = aff->scale;
    p.bias = aff->shift;
    p.addend = (const float *)aff->residual;
    p.addend_sp = aff->residual_sp;
    p.relu = aff->relu;
    p.out_sp = aff->out_sp;
    p.out_absmax = (unsigned *)aff->out_absmax;
    p.out_sinv = aff->out_sinv;
    p.bias_absmax = aff->bias_absmax;
  }
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
  p.pad = d->pad;
  p.stride_w = stride_w >= 0 ? stride_w : d->stride;
  p.pad_w = stride_w >= 0 ? pad_w : d->pad;
  p.ktotal = d->r * d->s * d->cin;
  p.b_row_len = p.ktotal;
  p.cin = d->cin;
  p.rows_per_group = (long long)d->n * d->ho * d->wo;
  p.src_img_stride = (long long)d->h * d->w * d->cin;
  p.imgs_per_group = d->n;
  p.ntaps = d->r * d->s;
  p.tap_ns = d->s;
  p.tap_step = 1;
  p.cls_step = 1;
  p.a_group_bytes = (long long)SP_BYTES * d->n * p.src_img_stride;
  p.b_bytes = (long long)SP_BYTES * d->cout * p.ktotal;
  MVG_REQUIRE(p.a_group_bytes < 0x7FFFFFF0ll && p.b_bytes < 0x7FFFFFF0ll, "split conv: a group / the weights exceed 2 GiB");
  MVG_REQUIRE(p.rows_per_group * (long long)d->cout < (1ll << 31), "split conv: a group of the output exceeds 2^31 elements");
  p.tap_ns_div = make_fastdiv((unsigned)p.tap_ns);
  p.ohw_div = make_fastdiv((unsigned)(p.out_h * p.out_w));
  p.ow_div = make_fastdiv((unsigned)p.out_w);
  // (the stem's row-window form multiplies 7 x 32 values per output where the filter has 7 x 7 x 3: count the filter's)
  const double flops = 2.0 * d->groups * (double)p.rows_per_group * d->cout * d->r * d->s * d->cin * (stride_w >= 0 ? 147.0 / 224.0 : 1.0);
  const double bytes = (double)SP_BYTES * (d->groups * (double)d->n * d->h * d->w * d->cin + (double)d->cout * d->r * d->s * d->cin) +
                       4.0 * d->groups * (double)p.rows_per_group * d->cout;
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  ProfScope ps(lin ? MVG_K_LINEAR_FPROP : MVG_K_CONV_FPROP, (hipStream_t)stream, flops, bytes);
  p.stats_partials = ceil_div(p.rows_per_group, SP_BM) * 2;    // = mvg_conv_stats_partials_split
  p.ncls = 1;
  class_from_params(p.cls[0], p);
  const bool lin_k = aff && aff->lin;
  return launch_igemm_split<false>(p, (hipStream_t)stream, lin_k, lin_k ? SP_BM : split_tile_rows(d->cout, d->r * d->s, p.rows_per_group));
}

int mvg_conv_fprop_split(const mvg_conv_desc *d, const void *x_sp, const float *x_sinv, const void *w_sp, const float *w_sinv, float *y,
                         float *stats, void *stream) {
  return fprop_split_impl(d, x_sp, x_sinv, w_sp, w_sinv, y, stats, stream, nullptr);
}

int mvg_conv_fprop_split_affine(const mvg_conv_desc *d, const void *x_sp, const float *x_sinv, const void *w_sp, const float *w_sinv,
                                void *out, int out_sp, const float *scale, const float *shift, const void *residual, int residual_sp,
                                int relu, void *stream) {
  MVG_REQUIRE(scale && shift && out, "fprop_split_affine: scale, shift and out are required");
  const SplitAffine a = {scale, shift, residual, residual_sp, relu, out_sp, 0, nullptr, nullptr, nullptr};
  return fprop_split_impl(d, x_sp, x_sinv, w_sp, w_sinv, out, nullptr, stream, &a);
}

struct SplitBnFuse {       // fused BatchNorm-backward reduce of the unit whose output gradient dx is (IgemmParams::bn_*)
  const float *y;
  const uint8_t *bits;
  const float *mean, *invstd, *rscale, *rshift;
  float *part;
  int part_rows;
};

static int dgrad_split_impl(const mvg_conv_desc *d, const void *dy_sp, const float *dy_sinv, const void *w_crsk_sp, const float *w_sinv,
                            float *dx, const float *addend, void *stream, const SplitBnFuse *bnf, const void *relu_mask_sp = nullptr,
                            bool lin_kernel = false, float *out_absmax = nullptr) {
  if (validate_split(d)) return 2;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.a = (const float *)dy_sp;
  p.b = (const float *)w_crsk_sp;
  p.a_sinv = dy_sinv;
  p.b_sinv = w_sinv;
  p.out = dx;
  p.addend = addend;
  p.mask = (const float *)relu_mask_sp;
  p.mask_sp = relu_mask_sp != nullptr;
  p.out_absmax = (unsigned *)out_absmax;
  if (bnf) {
    p.bn_y = bnf->y;
    p.bn_bits = bnf->bits;
    p.bn_mean = bnf->mean;
    p.bn_invstd = bnf->invstd;
    p.bn_rscale = bnf->rscale;
    p.bn_rshift = bnf->rshift;
    p.bn_part = bnf->part;
    p.bn_part_rows = bnf->part_rows;
  }
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
  p.pad = d->pad;
  p.ktotal = d->r * d->s * d->cout;
  p.b_row_len = d->r * d->s * d->cout;
  p.cin = d->cin;
  p.src_img_stride = (long long)d->ho * d->wo * d->cout;
  p.imgs_per_group = d->n;
  p.full_h = d->h;
  p.full_w = d->w;
  p.a_group_bytes = (long long)SP_BYTES * d->n * p.src_img_stride;
  p.b_bytes = (long long)SP_BYTES * d->cin * p.b_row_len;
  MVG_REQUIRE(p.a_group_bytes < 0x7FFFFFF0ll && p.b_bytes < 0x7FFFFFF0ll, "split conv: a group / the weights exceed 2 GiB");
  MVG_REQUIRE((long long)d->n * d->h * d->w * d->cin < (1ll << 31), "split conv: a group of dx exceeds 2^31 elements");
  const double flops = 2.0 * d->groups * (double)d->n * d->ho * d->wo * d->cout * d->r * d->s * d->cin;
  // (with the BatchNorm reduce on board the launch also reads that unit's y and mask bits: its algorithmic bytes)
  const double bytes = (double)SP_BYTES * (d->groups * (double)d->n * d->ho * d->wo * d->cout + (double)d->cout * d->r * d->s * d->cin) +
                       (bnf ? 8.25 : 4.0) * d->groups * (double)d->n * d->h * d->w * d->cin;
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  ProfScope ps(lin ? MVG_K_LINEAR_DGRAD : MVG_K_CONV_DGRAD, (hipStream_t)stream, flops, bytes);
  const int step = d->stride;
  IgemmParams m = p;
  m.ncls = 0;
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
      if (q.ntaps == 0 && !bnf) {
        if (addend != dx || !addend) {                 // nothing to do when the caller accumulates in place
          const long long n = (long long)d->groups * d->n * sub_h * sub_w * (d->cin / 4);
          long long blocks = (n + 255) / 256;
          if (blocks > 4096) blocks = 4096;
          hipLaunchKernelGGL(dgrad_empty_class_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)dx,
                             (const float4 *)addend, n, sub_h, sub_w, d->h, d->w, d->cin / 4, py, px);
          if (check_launch("dgrad_split(empty class)")) return 1;
        }
        continue;
      }
      if (m.ncls == 0) {
        m.tap_step = step;
        m.cls_step = step;
        m.rows_per_group = q.rows_per_group;
        m.ktotal = q.ktotal;
        m.out_h = q.out_h;
        m.out_w = q.out_w;
      }
      cls_k[m.ncls] = q.ktotal;
      class_from_params(m.cls[m.ncls++], q);
    }
  if (m.ncls == 0) return 0;
  for (int i = 1; i < m.ncls; ++i)                       // longest class first
    for (int j = i; j > 0 && cls_k[j] > cls_k[j - 1]; --j) {
      const IgemmClass tc = m.cls[j];
      m.cls[j] = m.cls[j - 1];
      m.cls[j - 1] = tc;
      const int tk = cls_k[j];
      cls_k[j] = cls_k[j - 1];
      cls_k[j - 1] = tk;
    }
  m.no_remap = m.ncls > 1;
  return launch_igemm_split<true>(m, (hipStream_t)stream, lin_kernel,
                                  lin_kernel ? SP_BM : split_tile_rows(d->cin, d->r * d->s, (long long)d->n * d->h * d->w));
}

int mvg_conv_dgrad_split(const mvg_conv_desc *d, const void *dy_sp, const float *dy_sinv, const void *w_crsk_sp, const float *w_sinv,
                         float *dx, const float *addend, const void *relu_mask_sp, void *stream) {
  MVG_REQUIRE(!relu_mask_sp || d->stride == 1, "dgrad_split: the sp ReLU mask is for stride-1 launches (Linear layers)");
  return dgrad_split_impl(d, dy_sp, dy_sinv, w_crsk_sp, w_sinv, dx, addend, stream, nullptr, relu_mask_sp);
}

int mvg_conv_dgrad_bn_partials_split(const mvg_conv_desc *d) {
  if (validate_split(d)) return -1;
  return dgrad_bn_partials(d, split_tile_rows(d->cin, d->r * d->s, (long long)d->n * d->h * d->w));
}

int mvg_conv_dgrad_split_bnreduce(const mvg_conv_desc *d, const void *dy_sp, const float *dy_sinv, const void *w_crsk_sp,
                                  const float *w_sinv, float *dx, const float *addend, const float *bn_y, const uint8_t *bn_bits,
                                  const float *bn_mean, const float *bn_invstd, const float *relu_scale, const float *relu_shift,
                                  float *partials, float *s1, float *s2, float *dgamma, float *dbeta, int accumulate,
                                  float *mx, const float *bn_gamma, float *dx_dy_sinv, void *stream) {
  // (dx_dy_sinv: receives the 2^-k of the dy that mvg_bn_bwd_apply_split will make from dx - the NEXT unit down the chain)
  MVG_REQUIRE(bn_y && bn_mean && bn_invstd && partials && s1 && s2, "dgrad_split_bnreduce: null argument");
  MVG_REQUIRE((bn_gamma == nullptr) == (dx_dy_sinv == nullptr) && (!dx_dy_sinv || mx), "dgrad_split_bnreduce: bn_gamma, dx_dy_sinv (and mx) go together");
  MVG_REQUIRE(!(bn_bits && relu_scale) && ((relu_scale == nullptr) == (relu_shift == nullptr)),
              "dgrad_split_bnreduce: give the ReLU mask either as bits or as (relu_scale, relu_shift)");
  const int P = mvg_conv_dgrad_bn_partials_split(d);
  MVG_REQUIRE(P > 0, "dgrad_split_bnreduce: bad descriptor");
  const SplitBnFuse f = {bn_y, bn_bits, bn_mean, bn_invstd, relu_scale, relu_shift, partials, mx ? 3 : 2};
  if (dgrad_split_impl(d, dy_sp, dy_sinv, w_crsk_sp, w_sinv, dx, addend, stream, &f)) return 1;
  ProfScope ps(MVG_K_BN_BWD_REDUCE, (hipStream_t)stream, 0.0, 8.0 * d->groups * (double)P * d->cin);
  return bn_bwd_finalize_launch(partials, d->groups, P, d->cin, s1, s2, dgamma, dbeta, accumulate, (hipStream_t)stream, mx, nullptr, nullptr,
                                bn_gamma, bn_invstd, (long long)d->n * d->h * d->w, dx_dy_sinv);
}

static void wgrad_split_tile(const mvg_conv_desc *d, int &bm, int &bn) {
  const int ncols = d->r * d->s * d->cin;
  bm = d->cout >= 128 ? 128 : 64;
  bn = ncols >= 128 ? 128 : 64;
  // 128 x 256 tiles (wave tile 64 x 128, two workgroups per CU) where the columns divide: 24 KB of operands per 16-pixel K-step for
  // twice the products of the 128 x 128 tile's 16 KB - like every kernel of this family wgrad runs into the CU's operand intake,
  // not the matrix pipe (scripts/conv_bench.py at C3: 5 - 14 % per layer, e.g. 512-channel 3x3 stride 2 0.377 -> 0.326 ms;
  // inside the step the family 15.4 -> 14.7 ms, C3 79.2 -> 78.3 ms on one box)
  if (bm == 128 && ncols >= 256 && ncols % 256 == 0) bn = 256;
  // ... and 192-column tiles for the 64- and 128-channel 3x3 layers (576 / 1152 columns = 3 / 6 whole tiles where 128-column tiles
  // leave a half-empty last one): 64 -> 64 at 56 x 56 0.553 -> 0.477 ms, 128 -> 128 0.367 -> 0.349 (15.66 -> 15.38 ms over C3's net)
  else if (ncols % 192 == 0) bn = 192;
}

int mvg_conv_wgrad_splits_split(const mvg_conv_desc *d) {
  if (validate_split(d)) return -1;
  int bm, bn;
  wgrad_split_tile(d, bm, bn);
  const int ncols = d->r * d->s * d->cin;
  const long long tiles = (long long)ceil_div(d->cout, bm) * ceil_div(ncols, bn);
  const long long pixels = (long long)d->groups * d->n * d->ho * d->wo;
  const int cus = compute_cus();
  // one resident round: three workgroups per CU (128-column tiles; measured at C3: 2 / 3 / 4 / 6 per CU -> 17.8 / 16.8 / 16.7 /
  // 17.7 ms of wgrad per step), two with the 256-column tiles (2 / 3 / 4 / 6 -> 15.6 / 16.8 / 16.4 / 17.3 ms)
  long long want = ((bn == 256 ? 2LL : 3LL) * cus) / tiles;
  long long maxs = pixels / 256;                           // at least 256 pixels (16 K-steps) per split
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

static int wgrad_split_impl(const mvg_conv_desc *d, const void *x_sp, const void *dy_sp, const float *dy_sinv, float *dw, float *workspace,
                            int splits, int accumulate, void *stream, int stride_w = -1, int pad_w = -1, const float *x_sinv = nullptr,
                            bool slabs_only = false) {
  if (stride_w < 0 && validate_split(d)) return 2;
  MVG_REQUIRE(splits >= 1, "wgrad_split: splits < 1");
  MVG_REQUIRE(splits == 1 || workspace != nullptr, "wgrad_split: workspace required for splits > 1");
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const float *)x_sp;
  p.dy = (const float *)dy_sp;
  p.dy_sinv = dy_sinv;
  p.x_sinv = x_sinv;
  p.h = d->h;
  p.w = d->w;
  p.cin = d->cin;
  p.cout = d->cout;
  p.r = d->r;
  p.s = d->s;
  p.stride = d->stride;
  p.pad = d->pad;
  p.stride_w = stride_w >= 0 ? stride_w : d->stride;
  p.pad_w = stride_w >= 0 ? pad_w : d->pad;
  p.ho = d->ho;
  p.wo = d->wo;
  p.ncols = d->r * d->s * d->cin;
  p.pixels = (long long)d->groups * d->n * d->ho * d->wo;
  p.pixels_per_split = ((p.pixels + splits - 1) / splits + 15) / 16 * 16;
  p.x_bytes = (long long)SP_BYTES * d->groups * d->n * d->h * d->w * d->cin;
  p.ohw_div = make_fastdiv((unsigned)(d->ho * d->wo));
  p.wo_div = make_fastdiv((unsigned)d->wo);
  p.cin_div = make_fastdiv((unsigned)d->cin);
  p.s_div = make_fastdiv((unsigned)d->s);
  MVG_REQUIRE(p.pixels_per_split * d->cout * SP_BYTES < 0x7FFFFFF0ll, "wgrad_split: split too large for 32-bit offsets");
  MVG_REQUIRE((long long)SP_BYTES * (p.pixels_per_split / (d->ho * d->wo) + 2) * d->h * d->w * d->cin < 0x7FFFFFF0ll,
              "wgrad_split: split too large for 32-bit offsets");
  int bm, bn;
  wgrad_split_tile(d, bm, bn);
  p.mtiles = ceil_div(d->cout, bm);
  p.ntiles = ceil_div(p.ncols, bn);
  p.out = splits == 1 ? dw : workspace;
  p.accumulate = (splits == 1) ? accumulate : 0;
  hipStream_t st = (hipStream_t)stream;
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  {
    const double flops = 2.0 * (double)p.pixels * d->cout * d->r * d->s * d->cin * (stride_w >= 0 ? 147.0 / 224.0 : 1.0);
    const double bytes = (double)SP_BYTES * ((double)d->groups * d->n * d->h * d->w * d->cin + (double)p.pixels * d->cout) +
                         4.0 * (double)d->cout * d->r * d->s * d->cin;
    ProfScope ps(lin ? MVG_K_LINEAR_WGRAD : MVG_K_CONV_WGRAD, st, flops, bytes);
    MVG_REQUIRE((long long)p.mtiles * p.ntiles * splits < (1LL << 31), "wgrad_split: grid too large");
    dim3 grid(p.mtiles * p.ntiles * splits), block(256);
    const bool incr = (long long)d->ho * d->wo >= 32;        // at most one image wrap per 16-pixel step
#define MVG_WGRAD_SPLIT(BM_, BN_)                                                                  \
  do {                                                                                             \
    if (incr) hipLaunchKernelGGL((wgrad_split_kernel<BM_, BN_, true>), grid, block, 0, st, p);     \
    else hipLaunchKernelGGL((wgrad_split_kernel<BM_, BN_, false>), grid, block, 0, st, p);         \
  } while (0)
    if (bm == 128 && bn == 256) MVG_WGRAD_SPLIT(128, 256);
    else if (bm == 128 && bn == 192) MVG_WGRAD_SPLIT(128, 192);
    else if (bm == 64 && bn == 192) MVG_WGRAD_SPLIT(64, 192);
    else if (bm == 128 && bn == 128) MVG_WGRAD_SPLIT(128, 128);
    else if (bm == 64 && bn == 128) MVG_WGRAD_SPLIT(64, 128);
    else if (bm == 128 && bn == 64) MVG_WGRAD_SPLIT(128, 64);
    else MVG_WGRAD_SPLIT(64, 64);
#undef MVG_WGRAD_SPLIT
    if (check_launch("conv_wgrad_split")) return 1;
  }
  if (splits > 1 && !slabs_only) {
    const long long n = (long long)d->cout * p.ncols;
    ProfScope ps(MVG_K_WGRAD_REDUCE, st, 0.0, 4.0 * n * (splits + 1));
    const int lanes = splits >= 32 ? 16 : (splits >= 8 ? 4 : 1);
    const long long blocks = (n / 4 + 256 / lanes - 1) / (256 / lanes);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, workspace, dw, n / 4, splits, accumulate, lanes);
    if (check_launch("wgrad_reduce")) return 1;
  }
  return 0;
}

int mvg_conv_wgrad_split(const mvg_conv_desc *d, const void *x_sp, const void *dy_sp, const float *dy_sinv, float *dw, float *workspace,
                         int splits, int accumulate, void *stream) {
  return wgrad_split_impl(d, x_sp, dy_sp, dy_sinv, dw, workspace, splits, accumulate, stream);
}

int mvg_conv_wgrad_split_slabs(const mvg_conv_desc *d, const void *x_sp, const void *dy_sp, const float *dy_sinv, float *workspace, int splits,
                               void *stream) {
  MVG_REQUIRE(splits > 1 && workspace, "wgrad_split_slabs: splits > 1 and a workspace (the slabs ARE the result)");
  return wgrad_split_impl(d, x_sp, dy_sp, dy_sinv, workspace, workspace, splits, 0, stream, -1, -1, nullptr, true);
}

int mvg_wgrad_reduce_batch(const float *const *host_slabs, float *const *host_dw, const int64_t *host_n, const int32_t *host_splits,
                           const int32_t *host_accumulate, int n, void *stream) {
  MVG_REQUIRE(host_slabs && host_dw && host_n && host_splits && host_accumulate && n >= 1 && n <= 8, "wgrad_reduce_batch: 1..8 items");
  ReduceBatch b;
  memset(&b, 0, sizeof(b));
  b.n = n;
  long long blocks = 0;
  double bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    MVG_REQUIRE(host_slabs[i] && host_dw[i] && host_n[i] > 0 && host_n[i] % 4 == 0 && host_splits[i] > 1, "wgrad_reduce_batch: bad item");
    ReduceItem &r = b.it[i];
    r.slabs = host_slabs[i];
    r.dw = host_dw[i];
    r.n4 = host_n[i] / 4;
    r.splits = host_splits[i];
    r.accumulate = host_accumulate[i];
    r.lanes = r.splits >= 32 ? 16 : (r.splits >= 8 ? 4 : 1);
    r.block0 = (int)blocks;
    blocks += (r.n4 + 256 / r.lanes - 1) / (256 / r.lanes);
    bytes += 4.0 * host_n[i] * (r.splits + 1);
  }
  MVG_REQUIRE(blocks < (1LL << 31), "wgrad_reduce_batch: grid too large");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_WGRAD_REDUCE, st, 0.0, bytes);
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
  return check_launch("wgrad_reduce_batch");
}

// ---- the fusion block's Linear layers on the split kernels (heads.py: a Linear = a 1x1 conv on a 1x1 map) ----------------
static mvg_conv_desc linear_desc(int rows, int fin, int fout) {
  mvg_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.groups = 1; d.n = rows; d.h = d.w = d.ho = d.wo = 1; d.cin = fin; d.cout = fout; d.r = d.s = 1; d.stride = 1; d.pad = 0;
  return d;
}

int mvg_split_colsum(const float *g, int rows, int cols, const float *absmax, void *out_sp, float *out_sinv, float *db, int accumulate,
                     void *stream) {
  MVG_REQUIRE(g && absmax && out_sp && out_sinv && rows > 0 && cols > 0 && cols % 32 == 0, "split_colsum: null argument or cols %% 32 != 0");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MVG_K_COLSUM, st, 0.0, (4.0 + SP_BYTES) * (double)rows * cols);
  hipLaunchKernelGGL(split_colsum_kernel, dim3(cols / 32), dim3(256), 0, st, g, rows, cols, (const unsigned *)absmax, (uint4 *)out_sp, out_sinv,
                     db, accumulate);
  return check_launch("split_colsum");
}

int mvg_linear_fprop_split(int rows, int fin, int fout, const void *x_sp, const float *x_sinv, const void *w_sp, const float *w_sinv,
                           const float *bias, int relu, void *out, int out_sp, float *out_sinv, const float *bias_absmax,
                           float *out_absmax, void *stream) {
  MVG_REQUIRE(bias != nullptr && out != nullptr, "linear_fprop_split: bias and out are required");
  MVG_REQUIRE(!out_sp || out_sinv, "linear_fprop_split: an sp result needs out_sinv (it is stored scaled)");
  MVG_REQUIRE(!(out_sp && out_absmax), "linear_fprop_split: out_absmax is for fp32 results");
  const mvg_conv_desc d = linear_desc(rows, fin, fout);
  const SplitAffine a = {nullptr, bias, nullptr, 0, relu, out_sp, 1, out_absmax, out_sinv, bias_absmax};
  return fprop_split_impl(&d, x_sp, x_sinv, w_sp, w_sinv, out, nullptr, stream, &a);
}

int mvg_linear_dgrad_split(int rows, int fin, int fout, const void *dy_sp, const float *dy_sinv, const void *wt_sp, const float *w_sinv,
                           float *dx, const float *addend, const void *relu_mask_sp, float *out_absmax, void *stream) {
  const mvg_conv_desc d = linear_desc(rows, fin, fout);
  return dgrad_split_impl(&d, dy_sp, dy_sinv, wt_sp, w_sinv, dx, addend, stream, nullptr, relu_mask_sp, true, out_absmax);
}

int mvg_linear_wgrad_split(int rows, int fin, int fout, const void *x_sp, const float *x_sinv, const void *dy_sp, const float *dy_sinv,
                           float *dw, float *workspace, int splits, int accumulate, void *stream) {
  const mvg_conv_desc d = linear_desc(rows, fin, fout);
  return wgrad_split_impl(&d, x_sp, dy_sp, dy_sinv, dw, workspace, splits, accumulate, stream, -1, -1, x_sinv);
}

// ---- the 7x7 stride-2 stem on the split kernels ("row-window" form) ------------------------------------------------
// A 3-channel (stored 4) image has too few channels for a K-step of the implicit GEMM: the fp32-MFMA kernel runs it at
// ~75 % of ITS peak (117 TFLOP/s with the padding channel).  Rewritten: xw[n][y][ox][j][c], j = 0..7, c = 0..3 holds
// image column 2 ox - 4 + j (zero outside the image) - the 8 columns the 7 taps of output column ox touch, plus one
// in front so that pixel pairs stay 16-byte aligned - and the stem becomes a 7 x 1 filter with 32 "channels", vertical
// stride 2 / padding 3, horizontal stride 1 / padding 0, over ho x wo windows: K = 7 x 32 = 224, the shape
// igemm_split16_kernel / wgrad_split_kernel are built for.  Weights w'[o][r][j][c] = w[o][r][j - 1][c] (j = 0, c = 3: zero).
static int stem_desc(const mvg_conv_desc *d, mvg_conv_desc *rw) {
  MVG_REQUIRE(d != nullptr, "stem (row-window): null descriptor");
  MVG_REQUIRE(d->r == 7 && d->s == 7 && d->stride == 2 && d->pad == 3 && d->cin == 4, "stem (row-window): 7x7 stride 2 pad 3, 4 stored channels");
  MVG_REQUIRE(d->w % 2 == 0 && d->ho == (d->h - 1) / 2 + 1 && d->wo == d->w / 2, "stem (row-window): even width; ho, wo inconsistent");
  MVG_REQUIRE(d->cout % 32 == 0 && d->groups > 0 && d->n > 0, "stem (row-window): cout must be a multiple of 32");
  *rw = *d;
  rw->w = d->wo;            // windows per image row
  rw->cin = 32;
  rw->s = 1;
  return 0;
}

int mvg_stem_rowwindow_split(const float *x_nhwc4, void *xw_sp, int64_t images, int h, int w, void *stream) {
  MVG_REQUIRE(x_nhwc4 && xw_sp && images > 0 && h > 0 && w > 0 && w % 2 == 0, "stem_rowwindow: bad arguments (even width)");
  hipStream_t st = (hipStream_t)stream;
  const long long n = images * h * (w / 2) * 4;                 // one thread per 8-value chunk (two pixels)
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 16.0 * (double)images * h * w + 32.0 * (double)n);
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(stem_rowwindow_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float4 *)x_nhwc4, (uint4 *)xw_sp, n, h, w);
  return check_launch("stem_rowwindow");
}

int mvg_stem_rowwindow_split_nchw(const float *x_nchw, void *xw_sp, int64_t images, int h, int w, void *stream) {
  MVG_REQUIRE(x_nchw && xw_sp && images > 0 && h > 0 && w > 0 && w % 2 == 0, "stem_rowwindow (NCHW): bad arguments (even width)");
  hipStream_t st = (hipStream_t)stream;
  const long long n = images * h * (w / 2) * 4;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 12.0 * (double)images * h * w + 32.0 * (double)n);
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(stem_rowwindow_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x_nchw, (uint4 *)xw_sp, n, h, w);
  return check_launch("stem_rowwindow_nchw");
}

int mvg_stem_fprop_split(const mvg_conv_desc *d, const void *xw_sp, const void *w_sp, const float *w_sinv, float *y, float *stats,
                         void *stream) {
  mvg_conv_desc rw;
  if (stem_desc(d, &rw)) return 2;
  return fprop_split_impl(&rw, xw_sp, nullptr, w_sp, w_sinv, y, stats, stream, nullptr, 1, 0);
}

int mvg_stem_wgrad_splits_split(const mvg_conv_desc *d) {
  mvg_conv_desc rw;
  if (stem_desc(d, &rw)) return -1;
  const long long tiles = (long long)ceil_div(rw.cout, rw.cout >= 128 ? 128 : 64) * ceil_div(7 * 32, 128);
  const long long pixels = (long long)rw.groups * rw.n * rw.ho * rw.wo;
  long long want = (3LL * compute_cus()) / tiles, maxs = pixels / 256;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

int mvg_stem_wgrad_split(const mvg_conv_desc *d, const void *xw_sp, const void *dy_sp, const float *dy_sinv, float *dw_rw, float *workspace,
                         int splits, int accumulate, void *stream) {
  mvg_conv_desc rw;
  if (stem_desc(d, &rw)) return 2;
  return wgrad_split_impl(&rw, xw_sp, dy_sp, dy_sinv, dw_rw, workspace, splits, accumulate, stream, 1, 0);
}

}  // extern "C"
