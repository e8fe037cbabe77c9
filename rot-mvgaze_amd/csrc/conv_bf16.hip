// Implicit-GEMM convolution / linear kernels on the gfx950 bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulation) - the "bf16 MFMA path" of BASELINE
// config C5.  Activations and weights are bf16 in HBM (NHWC / KRSC), BatchNorm statistics, biases and
// weight gradients stay fp32.  The fp32 path (conv_igemm.hip) is the parity path (1e-4); this one has
// its own declared tolerance (tests/test_bf16_gpu.py).
//
//   fprop : y[m][o]       = sum_{tap,c} x[pix(m,tap)][c]   * w [o][tap][c]     A gathered, B = KRSC weights
//   dgrad : dx[m][c]      = sum_{tap,o} dy[pix'(m,tap)][o] * wT[c][tap][o]     A gathered, B = the weights
//                                                                              transposed once per step (CRSK)
//   wgrad : dw[o][tap][c] = sum_m       dy[m][o]           * x[pix(m,tap)][c]  both operands pixel-major in
//                                                                              LDS, fragments by ds_read_b64_tr_b16
//
// Both fprop/dgrad operands are k-contiguous, so ONE kernel serves both: 256 threads = 4 waves, tile
// 128 x BN x 64 (a K-step is 128 bytes per row = one cache line per gathered pixel), 16-byte buffer loads
// with out-of-range predication -> registers -> LDS rows padded to 144 bytes (conflict-free
// ds_read_b128 fragments), three-stage pipeline like the fp32 kernel.  MFMA operand map
// (cdna_hip_programming.md 3): lane l holds A[row l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][col l&31].
// The epilogue goes through LDS (fp32 tile) so that global stores are 16-byte vectors of 8 bf16 along
// the channel axis; BN partial statistics come from the fp32 accumulators.
#include "bf16_tile.h"

namespace mvg {

constexpr int BF_BM = 128, BF_BK = 64, BF_LDK = BF_BK + 8;

// F32IO (the fusion block's Linear layers in the bf16 path): the gathered operand, the output and the epilogue
// operands (mask, addend) are fp32 in memory - only the matrix product runs in bf16: the loader reads 32 bytes
// per 8 k, rounds to bf16 on the way into LDS, the epilogue stores fp32.  The weights are the bf16 copies.
template <int BN, bool DGRAD, bool FASTA, bool F32IO = false>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(IgemmParams p) {
  constexpr int BM = BF_BM, BK = BF_BK, LDK = BF_LDK, WGM = 2, WGN = 2;
  constexpr unsigned EA = F32IO ? 4u : 2u;   // bytes per element of the gathered operand
  constexpr int AL = F32IO ? 2 : 1;          // 16-byte loads per 8 k
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KV = BK / 8;                 // 16-byte vectors per row
  constexpr int RPP = 256 / KV;              // rows per loader pass
  constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
  constexpr int A_ELEMS = BM * LDK, B_ELEMS = BN * LDK;
  constexpr int LDO = BN + 4;                // fp32 staging tile of the epilogue
  static_assert(2 * (A_ELEMS + B_ELEMS) * 2 >= BM * LDO * 4 + BM * 4, "the epilogue tile must fit into the operand buffers");
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (A_ELEMS + B_ELEMS)];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int a_kv = tid % KV, a_r0 = tid / KV;
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

  // ---- loader state
  unsigned a_img[A_PASSES];
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
      a_x0[i] = ox * p.stride_w - p.pad_w;
    }
    a_img[i] = (unsigned)(img * p.src_img_stride) * EA;
  }
  unsigned a_base[A_PASSES], a_vmask[A_PASSES], b_base[B_PASSES];
  bool b_ok[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int n = ntile * BN + a_r0 + i * RPP;
    b_ok[i] = n < p.ncols;
    b_base[i] = ((unsigned)n * (unsigned)p.b_row_len + (unsigned)a_kv * 8u) * 2u;
  }
  if (FASTA) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      a_base[i] = a_img[i] + (unsigned)((a_y0[i] * p.src_w + a_x0[i]) * p.src_c) * EA + (unsigned)a_kv * 8u * EA;
      unsigned m = 0;
      for (int t = 0; t < c.ntaps; ++t) {
        const int fr = (int)fdiv((unsigned)t, c.tap_ns_div), fs = t - fr * c.tap_ns;
        const int iy = DGRAD ? a_y0[i] - fr : a_y0[i] + fr;
        const int ix = DGRAD ? a_x0[i] - fs : a_x0[i] + fs;
        m |= (unsigned)(((unsigned)iy < (unsigned)p.src_h) & ((unsigned)ix < (unsigned)p.src_w)) << t;
      }
      a_vmask[i] = a_ok[i] ? m : 0u;
    }
  }
  const __amdgpu_buffer_rsrc_t rs_a =
      make_rsrc(reinterpret_cast<const char *>(p.a) + (long long)g * p.imgs_per_group * p.src_img_stride * EA, p.a_group_bytes);
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.b, p.b_bytes);

  u32x4 a_reg[A_PASSES][AL], b_reg[B_PASSES];
  auto load_a = [&](int i, unsigned off, bool ok) {
    a_reg[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, pred_off(off, ok), 0, 0);
    if constexpr (F32IO) a_reg[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, pred_off(off + 16u, ok), 0, 0);
  };
  auto load_tiles = [&](int kt) {
    // first k of this K-step; korder: (64-channel block, tap) instead of (tap, channel block) so that the
    // taps of a 3x3 filter revisit a pixel's 128-byte line within consecutive K-steps (L2 locality)
    int kstart = kt * BK;
    if (c.korder) {
      const int cblk = (int)fdiv((unsigned)kt, c.per_div), rem = kt - cblk * c.ntaps;
      kstart = (rem << p.src_c_shift) + cblk * BK;
    }
    if constexpr (FASTA) {
      const int ks = __builtin_amdgcn_readfirstlane(kstart);
      const int tap_u = c.ntaps > 1 ? (ks >> p.src_c_shift) : 0;
      const int chb = ks - (tap_u << p.src_c_shift);
      const int fru = (int)fdiv((unsigned)tap_u, c.tap_ns_div), fsu = tap_u - fru * c.tap_ns;
      const int disp = (fru * p.src_w + fsu) * p.src_c;
      const unsigned sdelta = (unsigned)((DGRAD ? -disp : disp) + chb) * EA;
      const bool kok_u = (kt < KT) & (ks < c.ktotal);
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        const bool ok = kok_u & (((a_vmask[i] >> tap_u) & 1u) != 0u);
        load_a(i, a_base[i] + sdelta, ok);
      }
      unsigned kb = (unsigned)ks * 2u;
      if (DGRAD) {
        const int btap = (c.tap_r0 + p.tap_step * fru) * p.s + c.tap_s0 + p.tap_step * fsu;
        kb = (unsigned)(btap * p.src_c + chb) * 2u;
      }
#pragma unroll
      for (int i = 0; i < B_PASSES; ++i)
        b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, pred_off(b_base[i] + kb, b_ok[i] & kok_u), 0, 0);
      return;
    }
    const int k0 = kstart + a_kv * 8;
    int tap = 0, ch = k0;
    if (c.ntaps > 1) {
      tap = k0 >> p.src_c_shift;
      ch = k0 - (tap << p.src_c_shift);
    }
    const int fr = (int)fdiv((unsigned)tap, c.tap_ns_div), fs = tap - fr * c.tap_ns;
    const bool kok = (kt < KT) & (k0 < c.ktotal);
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int iy = DGRAD ? a_y0[i] - fr : a_y0[i] + fr;
      const int ix = DGRAD ? a_x0[i] - fs : a_x0[i] + fs;
      const bool ok = a_ok[i] & kok & ((unsigned)iy < (unsigned)p.src_h) & ((unsigned)ix < (unsigned)p.src_w);
      load_a(i, a_img[i] + (unsigned)((iy * p.src_w + ix) * p.src_c + ch) * EA, ok);
    }
    unsigned koff = (unsigned)k0;
    if (DGRAD) {
      const int btap = (c.tap_r0 + p.tap_step * fr) * p.s + c.tap_s0 + p.tap_step * fs;
      koff = (unsigned)(btap * p.src_c + ch);
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i)
      b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(
          rs_b, pred_off(b_base[i] - (unsigned)a_kv * 16u + koff * 2u, b_ok[i] & kok), 0, 0);
  };
  auto store_tiles = [&](int buf) {
    unsigned short *As = smem + buf * (A_ELEMS + B_ELEMS);
    unsigned short *Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      u32x4 v = a_reg[i][0];
      if constexpr (F32IO) {          // 8 fp32 -> 8 bf16 (round to nearest even)
        const u32x4 lo = a_reg[i][0], hi = a_reg[i][1];
        v.x = pack_bf2(__uint_as_float(lo.x), __uint_as_float(lo.y));
        v.y = pack_bf2(__uint_as_float(lo.z), __uint_as_float(lo.w));
        v.z = pack_bf2(__uint_as_float(hi.x), __uint_as_float(hi.y));
        v.w = pack_bf2(__uint_as_float(hi.z), __uint_as_float(hi.w));
      }
      *reinterpret_cast<u32x4 *>(As + (a_r0 + i * RPP) * LDK + a_kv * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) *reinterpret_cast<u32x4 *>(Bs + (a_r0 + i * RPP) * LDK + a_kv * 8) = b_reg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tiles(0);
  store_tiles(0);
  load_tiles(1);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    store_tiles(cur ^ 1);
    load_tiles(kt + 2);
    const unsigned short *As = smem + cur * (A_ELEMS + B_ELEMS);
    const unsigned short *Bs = As + A_ELEMS;
    bf16x8 av[2][TM], bv[2][TN];
    auto load_frags = [&](int kg, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[slot][i] = *reinterpret_cast<const bf16x8 *>(As + (wm * WTM + i * 32 + li) * LDK + kg * 16 + lh * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bv[slot][j] = *reinterpret_cast<const bf16x8 *>(Bs + (wn * WTN + j * 32 + li) * LDK + kg * 16 + lh * 8);
    };
    load_frags(0, 0);
#pragma unroll
    for (int kg = 0; kg < BK / 16; ++kg) {
      if (kg + 1 < BK / 16) load_frags(kg + 1, (kg + 1) & 1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kg & 1][i], bv[kg & 1][j], acc[i][j], 0, 0, 0);
    }
    {
      constexpr int NLOADS = A_PASSES * AL + B_PASSES;
      constexpr int NMFMA = TM * TN * (BK / 16);
      constexpr int PER = NMFMA / NLOADS > 0 ? NMFMA / NLOADS : 1;
#pragma unroll
      for (int l = 0; l < NLOADS; ++l) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------
  bf16_epilogue<BF_BM, BN, 2, DGRAD, F32IO>(p, c, acc, smem, tid, g, mtile, ntile);
}

// ------------------------------------------------------------------------------------------
// LDS-DMA form of the same GEMM (uniform-tap shapes, bf16 operands): the loader is `buffer_load_dwordx4 ... lds`
// - global memory straight into LDS, no staging registers, no ds_write (a ds_write_b128 costs ~13 LDS cycles
// per wave-instruction: eight of them per thread and K-step held the register-staged kernel's LDS pipe busier
// than its matrix pipe).  The DMA writes lane l of a wave-instruction at base + 16*l, so the image is
// lane-linear: unpadded 128-byte rows, eight rows per wave-instruction, and the bank spread comes from the
// SOURCE side (cdna_hip_programming.md 5): the lane that fills 16-byte slot p of row r fetches chunk
// p ^ ((r >> 1) & 7) of that row's K-step, and a fragment read of chunk cc goes to slot cc ^ ((r >> 1) & 7):
// the 16 rows of a ds_read_b128 lane group then cover 16 different 16-byte bank slots (rows r and r + 1
// differ in the 128-byte half, the XOR spreads the other eight).  Out-of-range taps / rows / columns use the
// descriptor's range check: the DMA writes zeros.  Two LDS buffers, the next K-step's DMA is issued before
// this K-step's MFMAs; vmcnt(0) + barrier per K-step, two workgroups per CU cover each other's waits.
// ------------------------------------------------------------------------------------------
// STAGES = 1 (short K: a tile is a few K-steps between a cold prologue and the epilogue): one 32 KB stage, the
// epilogue staged in two passes (34 KB), four workgroups per CU that cover each other's DMA waits and epilogues.
template <int BN, bool DGRAD, int STAGES = 2>
__global__ __launch_bounds__(256, STAGES == 1 ? 4 : 2) void igemm_bf16_dma_kernel(IgemmParams p) {
  constexpr int BM = BF_BM, BK = BF_BK, WGM = 2, WGN = 2;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int ROW = BK;                     // elements per (unpadded) LDS row = 128 bytes
  constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;      // 32 rows (8 rows x 4 waves) per pass
  constexpr int A_ELEMS = BM * ROW, B_ELEMS = BN * ROW;
  constexpr int OP_ELEMS = STAGES * (A_ELEMS + B_ELEMS);      // ushort
  constexpr int EPI_PASSES = STAGES == 1 ? 2 : 1;
  constexpr int EPI_ELEMS = bf16_epilogue_bytes<BM, BN, EPI_PASSES, DGRAD>() / 2;
  constexpr int SMEM = OP_ELEMS > EPI_ELEMS ? OP_ELEMS : EPI_ELEMS;
  __shared__ __attribute__((aligned(16))) unsigned short smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
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

  // this lane fills slot (tid & 7) of row (tid >> 3) of every 32-row pass with source chunk a_kv
  const int r_in_pass = tid >> 3;
  const int a_kv = (tid & 7) ^ ((tid >> 4) & 7);              // (row >> 1) & 7 with row = 32 i + (tid >> 3)
  unsigned a_base[A_PASSES], a_vmask[A_PASSES], b_base[B_PASSES];
  bool b_ok[B_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const long long m = (long long)mtile * BM + r_in_pass + i * 32;
    const bool ok = m < c.rows_per_group;
    const int mm = ok ? (int)m : 0;
    const int img = (int)fdiv((unsigned)mm, c.ohw_div);
    const int rem = mm - img * ohw;
    const int oy = (int)fdiv((unsigned)rem, c.ow_div), ox = rem - oy * c.out_w;
    const int y0 = DGRAD ? oy + c.cls_cy : oy * p.stride - p.pad;
    const int x0 = DGRAD ? ox + c.cls_cx : ox * p.stride_w - p.pad_w;
    a_base[i] = (unsigned)(img * p.src_img_stride * 2) + (unsigned)((y0 * p.src_w + x0) * p.src_c) * 2u + (unsigned)a_kv * 16u;
    unsigned msk = 0;
    for (int t = 0; t < c.ntaps; ++t) {
      const int fr = (int)fdiv((unsigned)t, c.tap_ns_div), fs = t - fr * c.tap_ns;
      const int iy = DGRAD ? y0 - fr : y0 + fr;
      const int ix = DGRAD ? x0 - fs : x0 + fs;
      msk |= (unsigned)(((unsigned)iy < (unsigned)p.src_h) & ((unsigned)ix < (unsigned)p.src_w)) << t;
    }
    a_vmask[i] = ok ? msk : 0u;
  }
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int n = ntile * BN + r_in_pass + i * 32;
    b_ok[i] = n < p.ncols;
    b_base[i] = ((unsigned)n * (unsigned)p.b_row_len + (unsigned)a_kv * 8u) * 2u;
  }
  const __amdgpu_buffer_rsrc_t rs_a =
      make_rsrc(reinterpret_cast<const unsigned short *>(p.a) + (long long)g * p.imgs_per_group * p.src_img_stride, p.a_group_bytes);
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.b, p.b_bytes);
  typedef __attribute__((address_space(3))) void *lds_vp;

  auto issue = [&](int kt, int buf) {
    int kstart = kt * BK;
    if (c.korder) {
      const int cblk = (int)fdiv((unsigned)kt, c.per_div), rem = kt - cblk * c.ntaps;
      kstart = (rem << p.src_c_shift) + cblk * BK;
    }
    const int ks = __builtin_amdgcn_readfirstlane(kstart);
    const int tap_u = c.ntaps > 1 ? (ks >> p.src_c_shift) : 0;
    const int chb = ks - (tap_u << p.src_c_shift);
    const int fru = (int)fdiv((unsigned)tap_u, c.tap_ns_div), fsu = tap_u - fru * c.tap_ns;
    const int disp = (fru * p.src_w + fsu) * p.src_c;
    const unsigned sdelta = (unsigned)(((DGRAD ? -disp : disp) + chb) * 2);
    unsigned kb = (unsigned)ks * 2u;
    if (DGRAD) {
      const int btap = (c.tap_r0 + p.tap_step * fru) * p.s + c.tap_s0 + p.tap_step * fsu;
      kb = (unsigned)(btap * p.src_c + chb) * 2u;
    }
    unsigned short *As = smem + buf * (A_ELEMS + B_ELEMS) + wave * 8 * ROW;       // this wave's 8 rows of pass 0 (wave-uniform)
    unsigned short *Bs = smem + buf * (A_ELEMS + B_ELEMS) + A_ELEMS + wave * 8 * ROW;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const bool ok = ((a_vmask[i] >> tap_u) & 1u) != 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_vp)(As + i * 32 * ROW), 16, (int)pred_off(a_base[i] + sdelta, ok), 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_vp)(Bs + i * 32 * ROW), 16, (int)pred_off(b_base[i] + kb, b_ok[i]), 0, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses: row R, chunk cc = 2 kg + lh -> slot cc ^ ((R >> 1) & 7)
  int a_row[TM], b_row[TN], a_sw[TM], b_sw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    a_row[i] = (wm * WTM + i * 32 + li) * ROW;
    a_sw[i] = ((wm * WTM + i * 32 + li) >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    b_row[j] = (wn * WTN + j * 32 + li) * ROW;
    b_sw[j] = ((wn * WTN + j * 32 + li) >> 1) & 7;
  }

  if constexpr (STAGES == 2) {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = STAGES == 2 ? (kt & 1) : 0;
    if constexpr (STAGES == 2) {
      if (kt + 1 < KT) issue(kt + 1, cur ^ 1);
    } else {
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const unsigned short *As = smem + cur * (A_ELEMS + B_ELEMS);
    const unsigned short *Bs = As + A_ELEMS;
#pragma unroll
    for (int kg = 0; kg < BK / 16; ++kg) {
      bf16x8 av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const bf16x8 *>(As + a_row[i] + (((2 * kg + lh) ^ a_sw[i]) << 3));
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const bf16x8 *>(Bs + b_row[j] + (((2 * kg + lh) ^ b_sw[j]) << 3));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the next K-step's DMA has landed
    __syncthreads();                                         // ... for every wave, and everyone is done reading `cur`
  }
  bf16_epilogue<BF_BM, BN, 2, DGRAD, false, EPI_PASSES, false, DGRAD>(p, c, acc, smem, tid, g, mtile, ntile);
}

// ------------------------------------------------------------------------------------------
// wgrad: dw[o][tap][c] = sum over pixels of dy[pix][o] * x[pix at tap][c]; M = cout, N = (tap, c),
// K = pixels split into slabs (fixed-order reduce: wgrad_reduce_kernel).  Both operands arrive
// pixel-major (NHWC rows), i.e. k-strided for the MFMA: the LDS images stay pixel-major [k][m] and the
// fragments are read with the hardware transpose read ds_read_b64_tr_b16 (4 k x 16 m per 16-lane group,
// cdna_hip_programming.md T10); rows of (BM + 32) * 2 bytes put the four k-rows of a read on disjoint
// bank ranges (conflict-free).
// ------------------------------------------------------------------------------------------

// F32IN (Linear layers of the fusion block in the bf16 path): x and dy are fp32 in memory and rounded to bf16 on
// the way into LDS; the bias gradient (column sums of the fp32 dy) rides along like in the fp32 kernel.
template <int BM, int BN, bool INCR, bool F32IN = false>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(WgradParams p) {
  constexpr int BK = 64, WGM = 2, WGN = 2;
  constexpr unsigned EB = F32IN ? 4u : 2u;   // bytes per element of x and dy
  constexpr int AL = F32IN ? 2 : 1;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + 32, LDB = BN + 32;
  constexpr int MV = BM / 8, NVB = BN / 8;
  constexpr int A_KRPP = 256 / MV;
  constexpr int A_PASSES = BK / A_KRPP;
  constexpr int A_ELEMS = BK * LDA, B_ELEMS = BK * LDB;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (A_ELEMS + B_ELEMS)];

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
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(dy + m_begin * p.cout * EB, (long long)EB * m_count * p.cout);
  const long long x_img_elems = (long long)p.h * p.w * p.cin;
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(x + img0 * x_img_elems * EB, p.x_bytes - (long long)EB * img0 * x_img_elems);
  const int a_mv = tid % MV, a_k0 = tid / MV;
  const int a_col = mtile * BM + a_mv * 8;
  const bool a_cok = a_col < p.cout;
  unsigned a_off[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i)
    a_off[i] = a_cok ? (unsigned)((a_k0 + i * A_KRPP) * p.cout + a_col) * EB : 0x80000000u;

  // B (x gather): a thread's loads of one K-step all come from ONE pixel row (row = tid / B_TPR) and differ
  // in the column group; INCR carries (oy, ox, image offset) of that pixel and steps it by BK pixels per K-step.
  constexpr int B_TPR = 256 / BK;                       // threads per k-row
  constexpr int B_CPT = NVB / B_TPR;                    // column groups per thread
  static_assert(NVB % B_TPR == 0 && B_CPT >= 1, "wgrad loader mapping");
  const int i_row = tid / B_TPR, i_nv0 = tid % B_TPR;
  int i_dy[B_CPT], i_dx[B_CPT];
  unsigned i_tconst[B_CPT];
  bool i_cok[B_CPT];
#pragma unroll
  for (int j = 0; j < B_CPT; ++j) {
    const int col = ntile * BN + (i_nv0 + j * B_TPR) * 8;
    i_cok[j] = col < p.ncols;
    const int tap = (int)fdiv((unsigned)(i_cok[j] ? col : 0), p.cin_div);
    const int cc = (i_cok[j] ? col : 0) - tap * p.cin;
    const int fr = (int)fdiv((unsigned)tap, p.s_div), fs = tap - fr * p.s;
    i_dy[j] = fr - p.pad;
    i_dx[j] = fs - p.pad_w;
    i_tconst[j] = (unsigned)((i_dy[j] * p.w + i_dx[j]) * p.cin + cc) * EB;
  }
  const unsigned row_bytes = (unsigned)(p.stride * p.w * p.cin) * EB, col_bytes = (unsigned)(p.stride_w * p.cin) * EB;
  const unsigned img_bytes = (unsigned)x_img_elems * EB;
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
  u32x4 a_reg[A_PASSES][AL], b_reg[B_CPT][AL];
  const bool do_bias = F32IN && (p.db != nullptr) && (ntile == 0);
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      a_reg[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_off[i], 0, 0);  // rows >= m_count: beyond the descriptor = zeros
      if constexpr (F32IN) a_reg[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_off[i] + 16u, 0, 0);
      a_off[i] += (unsigned)(BK * p.cout) * EB;
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
      b_reg[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, pred_off(pixoff + i_tconst[j], ok), 0, 0);
      if constexpr (F32IN) b_reg[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, pred_off(pixoff + i_tconst[j] + 16u, ok), 0, 0);
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
  auto to_bf16 = [&](const u32x4 (&r)[AL]) -> u32x4 {
    if constexpr (!F32IN) return r[0];
    u32x4 v;
    v.x = pack_bf2(__uint_as_float(r[0].x), __uint_as_float(r[0].y));
    v.y = pack_bf2(__uint_as_float(r[0].z), __uint_as_float(r[0].w));
    v.z = pack_bf2(__uint_as_float(r[AL - 1].x), __uint_as_float(r[AL - 1].y));
    v.w = pack_bf2(__uint_as_float(r[AL - 1].z), __uint_as_float(r[AL - 1].w));
    return v;
  };
  auto store_tiles = [&](int buf) {
    unsigned short *As = smem + buf * (A_ELEMS + B_ELEMS);
    unsigned short *Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) *reinterpret_cast<u32x4 *>(As + (a_k0 + i * A_KRPP) * LDA + a_mv * 8) = to_bf16(a_reg[i]);
    if constexpr (F32IN) {
      if (do_bias) {                   // every K-step's rows exactly once, in order, from the fp32 values
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
          bsum[0] += __uint_as_float(a_reg[i][0].x); bsum[1] += __uint_as_float(a_reg[i][0].y);
          bsum[2] += __uint_as_float(a_reg[i][0].z); bsum[3] += __uint_as_float(a_reg[i][0].w);
          bsum[4] += __uint_as_float(a_reg[i][AL - 1].x); bsum[5] += __uint_as_float(a_reg[i][AL - 1].y);
          bsum[6] += __uint_as_float(a_reg[i][AL - 1].z); bsum[7] += __uint_as_float(a_reg[i][AL - 1].w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < B_CPT; ++j) *reinterpret_cast<u32x4 *>(Bs + i_row * LDB + (i_nv0 + j * B_TPR) * 8) = to_bf16(b_reg[j]);
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
    const unsigned short *As = smem + cur * (A_ELEMS + B_ELEMS);
    const unsigned short *Bs = As + A_ELEMS;
#pragma unroll
    for (int kg = 0; kg < BK / 16; ++kg) {
      bf16x8 av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = tr_frag(As, LDA, kg * 16, wm * WTM + i * 32, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = tr_frag(Bs, LDB, kg * 16, wn * WTN + j * 32, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  if constexpr (F32IN) {
    if (do_bias) {                     // loader rows (a_k0) summed through LDS in a fixed order; the K loop ended with a barrier
      float *sh = reinterpret_cast<float *>(smem);
#pragma unroll
      for (int k = 0; k < 8; ++k) sh[tid * 8 + k] = bsum[k];
      __syncthreads();
      if (a_k0 == 0 && a_cok) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = sh[a_mv * 8 + k];
        for (int r = 1; r < A_KRPP; ++r)
#pragma unroll
          for (int k = 0; k < 8; ++k) t[k] += sh[(r * MV + a_mv) * 8 + k];
        float *dst = p.db + (long long)split * p.cout + a_col;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = (p.accumulate ? dst[k] : 0.f) + t[k];
      }
    }
  }
  float *out = p.out + (long long)split * p.cout * p.ncols;
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

// A stride-2 parity class without taps: dx = addend (or zero) on that class's pixels (16-byte vectors of 8 bf16).
__global__ __launch_bounds__(256) void dgrad_empty_class_bf16_kernel(u32x4 *__restrict__ dx, const u32x4 *__restrict__ addend,
                                                                     long long n, int sub_h, int sub_w, int full_h, int full_w,
                                                                     int c8, int py, int px) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int cc = (int)(i % c8);
    long long t = i / c8;
    const int x2 = (int)(t % sub_w);
    t /= sub_w;
    const int y2 = (int)(t % sub_h);
    const long long img = t / sub_h;
    const long long off = ((img * full_h + 2 * y2 + py) * full_w + 2 * x2 + px) * c8 + cc;
    u32x4 z = {0u, 0u, 0u, 0u};
    dx[off] = addend ? addend[off] : z;
  }
}

// The stem's folded row-window operand (mvg_stem_fprop_bf16): window m of an image row holds image columns 4 m - 4 ..
// 4 m + 11 x 4 channels (3 real) in bf16 - 64 values, the K-step of the LDS-DMA kernel - straight from the NCHW fp32
// input.  One thread per 8-value chunk = two columns.
__global__ __launch_bounds__(256) void stem_rowwindow_bf16_kernel(const float *__restrict__ x, uint4 *__restrict__ xw, long long n, int h, int w) {
  const int wm = w >> 2;
  const long long plane = (long long)h * w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int q = (int)(i & 7);
    const long long t = i >> 3;
    const int m = (int)(t % wm);
    const long long row = t / wm;                    // image * h + y
    const long long img = row / h;
    const int yy = (int)(row - img * h);
    const int c0 = 4 * m - 4 + 2 * q;                 // even: both columns in range or both out (w % 4 == 0)
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (c0 >= 0 && c0 < w) {
      const float *src = x + img * 3 * plane + (long long)yy * w + c0;
      const float2 r = *reinterpret_cast<const float2 *>(src), g = *reinterpret_cast<const float2 *>(src + plane),
                   b = *reinterpret_cast<const float2 *>(src + 2 * plane);
      o = make_uint4(bf16_pack2(r.x, g.x), bf16_pack2(b.x, 0.f), bf16_pack2(r.y, g.y), bf16_pack2(b.y, 0.f));
    }
    xw[i] = o;
  }
}

// fp32 KRSC weights -> bf16 KRSC (cin zero-padded to cin_pad) and, optionally, the transposed CRSK copy the
// backward-data kernel reads ([cin_pad][r*s][cout]).  One thread per (o, tap, c).
__global__ __launch_bounds__(256) void cast_weights_bf16_kernel(const float *__restrict__ w, unsigned short *__restrict__ wk,
                                                                unsigned short *__restrict__ wt, int cout, int rs, int cin,
                                                                int cin_pad) {
  const long long total = (long long)cout * rs * cin_pad;
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
// host side
// ------------------------------------------------------------------------------------------
static int validate_bf16(const mvg_conv_desc *d) {
  if (validate(d)) return 2;
  MVG_REQUIRE(d->cin % 8 == 0 && d->cout % 8 == 0, "bf16 conv: cin and cout must be multiples of 8 (got %d, %d)", d->cin, d->cout);
  return 0;
}

template <bool DGRAD>
static int launch_igemm_bf16(IgemmParams &p, hipStream_t st, bool f32io = false) {
  const int bn = p.ncols >= 128 ? 128 : 64;
  p.ntiles = ceil_div(p.ncols, bn);
  p.splits = 1;
  p.sk_tiles = 0;
  long long tiles = 0;
  bool fasta = true;
  const bool bnf = DGRAD && p.bn_part != nullptr;
  p.bn_parts = 0;
  for (int i = 0; i < p.ncls; ++i) {
    IgemmClass &c = p.cls[i];
    c.mtiles_per_group = ceil_div(c.rows_per_group, BF_BM);
    c.KT = c.ktotal > 0 ? ceil_div(c.ktotal, BF_BK) : (bnf ? 0 : 1);     // fused reduce: a class without taps is epilogue only
    c.korder = (c.ntaps > 1 && p.src_c % BF_BK == 0) ? 1 : 0;
    c.per_div = make_fastdiv((unsigned)(c.ntaps > 0 ? c.ntaps : 1));
    c.tile0 = (int)tiles;
    c.unit0 = 0;
    c.part0 = p.bn_parts;
    p.bn_parts += c.mtiles_per_group;
    tiles += (long long)p.groups * c.mtiles_per_group * p.ntiles;
    fasta = fasta && (c.ntaps >= 1 || bnf) && c.ntaps <= 32 && c.ktotal % BF_BK == 0 && p.src_c % BF_BK == 0;
  }
  MVG_REQUIRE(tiles < (1LL << 31), "bf16 conv: grid too large");
  MVG_REQUIRE(!bnf || (fasta && !f32io), "bf16 dgrad with a fused BatchNorm reduce: bf16 operands, channel counts in multiples of 64");
  if (tiles <= 0) return 0;
  dim3 grid((unsigned)tiles), block(256);
  if (f32io) {
    if (bn == 128) {
      if (fasta) hipLaunchKernelGGL((igemm_bf16_kernel<128, DGRAD, true, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((igemm_bf16_kernel<128, DGRAD, false, true>), grid, block, 0, st, p);
    } else {
      if (fasta) hipLaunchKernelGGL((igemm_bf16_kernel<64, DGRAD, true, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((igemm_bf16_kernel<64, DGRAD, false, true>), grid, block, 0, st, p);
    }
  } else if (fasta) {
    // uniform-tap shapes: the LDS-DMA kernel in its one-stage, four-workgroups-per-CU form (measured faster than the
    // two-stage form at every ResNet-50 shape: C5 fprop 10.4 -> 8.4 ms, dgrad 9.5 -> 7.5 ms)
    if (bn == 128) hipLaunchKernelGGL((igemm_bf16_dma_kernel<128, DGRAD, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_bf16_dma_kernel<64, DGRAD, 1>), grid, block, 0, st, p);
  } else if (bn == 128) {                   // the stem (4-channel taps): register-staged loader
    hipLaunchKernelGGL((igemm_bf16_kernel<128, DGRAD, false>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((igemm_bf16_kernel<64, DGRAD, false>), grid, block, 0, st, p);
  }
  return check_launch(DGRAD ? "conv_dgrad_bf16" : "conv_fprop_bf16");
}

}  // namespace mvg

using namespace mvg;

extern "C" {

int mvg_conv_stats_partials_bf16(const mvg_conv_desc *d, int32_t *rows_per_partial) {
  if (validate_bf16(d)) return -1;
  const long long rows = (long long)d->n * d->ho * d->wo;
  if (rows_per_partial) *rows_per_partial = BF_BM / 2;
  return ceil_div(rows, BF_BM) * 2;
}

int mvg_cast_weights_bf16(const mvg_conv_desc *d, const float *w, int cin_src, void *w_krsc, void *w_crsk, void *stream) {
  MVG_REQUIRE(d && w && w_krsc, "cast_weights_bf16: null argument");
  MVG_REQUIRE(cin_src > 0 && cin_src <= d->cin, "cast_weights_bf16: cin_src must be in (0, cin]");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)d->cout * d->r * d->s * d->cin;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, (w_crsk ? 8.0 : 6.0) * (double)total);
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_weights_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, (unsigned short *)w_krsc,
                     (unsigned short *)w_crsk, d->cout, d->r * d->s, cin_src, d->cin);
  return check_launch("cast_weights_bf16");
}

// stride_w >= 0: horizontal stride / padding differ from d->stride / d->pad, and the BatchNorm partials of GEMM columns
// c and c + cout/2 are two partials of channel c (the stem's folded row-window form, whose descriptor the caller checked)
static int fprop_bf16_impl(const mvg_conv_desc *d, const void *x, const void *wgt, void *y, const float *bias, int relu,
                           float *stats, void *stream, bool f32io, int stride_w = -1, int pad_w = -1) {
  if (stride_w < 0 && validate_bf16(d)) return 2;
  const long long EA = f32io ? 4 : 2;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.a = (const float *)x;
  p.b = (const float *)wgt;
  p.out = (float *)y;
  p.bias = bias;
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
  p.pad = d->pad;
  p.stride_w = stride_w >= 0 ? stride_w : d->stride;
  p.pad_w = stride_w >= 0 ? pad_w : d->pad;
  p.stats_fold = stride_w >= 0 ? 1 : 0;
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
  p.a_group_bytes = EA * d->n * p.src_img_stride;
  p.b_bytes = 2ll * d->cout * p.ktotal;
  MVG_REQUIRE(p.a_group_bytes < 0x7FFFFFF0ll && p.b_bytes < 0x7FFFFFF0ll, "bf16 conv: a group / the weights exceed 2 GiB");
  MVG_REQUIRE(p.rows_per_group * (long long)d->cout < (1ll << 31), "bf16 conv: a group of the output exceeds 2^31 elements");
  p.tap_ns_div = make_fastdiv((unsigned)p.tap_ns);
  p.ohw_div = make_fastdiv((unsigned)(p.out_h * p.out_w));
  p.ow_div = make_fastdiv((unsigned)p.out_w);
  const double acin = d->cin == 8 && d->r == 7 ? 3.0 : (double)d->cin;       // the stem's channels 3..7 are zero padding
  const double flops = 2.0 * d->groups * (double)p.rows_per_group * d->cout * d->r * d->s * acin * (stride_w >= 0 ? 147.0 / 448.0 : 1.0);
  const double bytes = 2.0 * (d->groups * (double)d->n * d->h * d->w * acin + (double)d->cout * d->r * d->s * acin +
                              d->groups * (double)p.rows_per_group * d->cout);
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  ProfScope ps(lin ? MVG_K_LINEAR_FPROP : MVG_K_CONV_FPROP, (hipStream_t)stream, flops, bytes);
  p.ncls = 1;
  class_from_params(p.cls[0], p);
  return launch_igemm_bf16<false>(p, (hipStream_t)stream, f32io);
}

int mvg_conv_fprop_bf16(const mvg_conv_desc *d, const void *x, const void *wgt, void *y, const float *bias, int relu,
                        float *stats, void *stream) {
  return fprop_bf16_impl(d, x, wgt, y, bias, relu, stats, stream, false);
}

struct Bf16BnFuse {        // fused BatchNorm-backward reduce of the unit whose output gradient dx is (IgemmParams::bn_*)
  const void *y;           // bf16, like dx
  const uint8_t *bits;     // one byte per 8 channels (mvg_bn_apply_bits_bf16)
  const float *mean, *invstd, *rscale, *rshift;
  float *part;
};

static int dgrad_bf16_impl(const mvg_conv_desc *d, const void *dy, const void *wgt_crsk, void *dx, const void *mask,
                           const void *addend, void *stream, bool f32io, const Bf16BnFuse *bnf = nullptr) {
  if (validate_bf16(d)) return 2;
  const long long EA = f32io ? 4 : 2;
  MVG_REQUIRE(!f32io || d->stride == 1, "bf16 dgrad with fp32 operands: stride 1 only");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.a = (const float *)dy;
  p.b = (const float *)wgt_crsk;
  p.out = (float *)dx;
  p.mask = (const float *)mask;
  p.addend = (const float *)addend;
  if (bnf) {
    p.bn_y = (const float *)bnf->y;
    p.bn_bits = bnf->bits;
    p.bn_mean = bnf->mean;
    p.bn_invstd = bnf->invstd;
    p.bn_rscale = bnf->rscale;
    p.bn_rshift = bnf->rshift;
    p.bn_part = bnf->part;
    p.bn_part_rows = 2;
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
  p.a_group_bytes = EA * d->n * p.src_img_stride;
  p.b_bytes = 2ll * d->cin * p.b_row_len;
  MVG_REQUIRE(p.a_group_bytes < 0x7FFFFFF0ll && p.b_bytes < 0x7FFFFFF0ll, "bf16 conv: a group / the weights exceed 2 GiB");
  MVG_REQUIRE((long long)d->n * d->h * d->w * d->cin < (1ll << 31), "bf16 conv: a group of dx exceeds 2^31 elements");
  const double flops = 2.0 * d->groups * (double)d->n * d->ho * d->wo * d->cout * d->r * d->s * d->cin;
  const double bytes = 2.0 * (d->groups * (double)d->n * d->ho * d->wo * d->cout + (double)d->cout * d->r * d->s * d->cin) +
                       (bnf ? 4.125 : 2.0) * d->groups * (double)d->n * d->h * d->w * d->cin;      // (fused reduce: + y and the mask bits)
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
          const long long n = (long long)d->groups * d->n * sub_h * sub_w * (d->cin / 8);
          long long blocks = (n + 255) / 256;
          if (blocks > 4096) blocks = 4096;
          hipLaunchKernelGGL(dgrad_empty_class_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (u32x4 *)dx,
                             (const u32x4 *)addend, n, sub_h, sub_w, d->h, d->w, d->cin / 8, py, px);
          if (check_launch("dgrad_bf16(empty class)")) return 1;
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
  return launch_igemm_bf16<true>(m, (hipStream_t)stream, f32io);
}

int mvg_conv_dgrad_bf16(const mvg_conv_desc *d, const void *dy, const void *wgt_crsk, void *dx, const void *mask,
                        const void *addend, void *stream) {
  return dgrad_bf16_impl(d, dy, wgt_crsk, dx, mask, addend, stream, false);
}

int mvg_conv_dgrad_bn_partials_bf16(const mvg_conv_desc *d) {
  if (validate_bf16(d)) return -1;
  return dgrad_bn_partials(d, BF_BM);
}

int mvg_conv_dgrad_bf16_bnreduce(const mvg_conv_desc *d, const void *dy, const void *wgt_crsk, void *dx, const void *addend,
                                 const void *bn_y, const uint8_t *bn_bits, const float *bn_mean, const float *bn_invstd,
                                 const float *relu_scale, const float *relu_shift, float *partials, float *s1, float *s2,
                                 float *dgamma, float *dbeta, int accumulate, void *stream) {
  MVG_REQUIRE(bn_y && bn_mean && bn_invstd && partials && s1 && s2, "dgrad_bf16_bnreduce: null argument");
  MVG_REQUIRE(!(bn_bits && relu_scale) && ((relu_scale == nullptr) == (relu_shift == nullptr)),
              "dgrad_bf16_bnreduce: give the ReLU mask either as bits or as (relu_scale, relu_shift)");
  const int P = mvg_conv_dgrad_bn_partials_bf16(d);
  MVG_REQUIRE(P > 0, "dgrad_bf16_bnreduce: bad descriptor");
  const Bf16BnFuse f = {bn_y, bn_bits, bn_mean, bn_invstd, relu_scale, relu_shift, partials};
  if (dgrad_bf16_impl(d, dy, wgt_crsk, dx, nullptr, addend, stream, false, &f)) return 1;
  ProfScope ps(MVG_K_BN_BWD_REDUCE, (hipStream_t)stream, 0.0, 8.0 * d->groups * (double)P * d->cin);
  return bn_bwd_finalize_launch(partials, d->groups, P, d->cin, s1, s2, dgamma, dbeta, accumulate, (hipStream_t)stream, nullptr, bn_mean,
                                bn_invstd);
}

// ---- the 7x7 stride-2 stem on the LDS-DMA kernels: folded row windows -------------------------------------------------
// These kernels' K-step is 64 channels of ONE tap; the stem offers 8 stored channels per tap (3 real): round 2 ran it on
// the register-staged kernel with K = 49 x 8 = 392 (1.05 ms of a C5 step).  Folded: window m of an image row = image
// columns 4 m - 4 .. 4 m + 11 x 4 channels = 64 values serves TWO output columns - ox = 2 m through filter taps at
// j = s + 1, ox = 2 m + 1 at j = s + 3 - as 2 x cout GEMM columns, whose [.., wo/2, 2 cout] output IS y [.., wo, cout]
// in memory.  A 7 x 1 filter over 64 "channels", vertical stride 2 / pad 3, horizontal stride 1 / pad 0: K = 448 per
// two outputs (224 each; the filter has 147).  BatchNorm partials: columns c and c + cout are two partials of channel c.
static int stem_fold_desc(const mvg_conv_desc *d, mvg_conv_desc *rw) {
  MVG_REQUIRE(d != nullptr, "stem (folded windows): null descriptor");
  MVG_REQUIRE(d->r == 7 && d->s == 7 && d->stride == 2 && d->pad == 3, "stem (folded windows): 7x7 stride 2 pad 3");
  MVG_REQUIRE(d->w % 4 == 0 && d->ho == (d->h - 1) / 2 + 1 && d->wo == d->w / 2, "stem (folded windows): width %% 4; ho, wo inconsistent");
  MVG_REQUIRE(d->cout % 32 == 0 && d->groups > 0 && d->n > 0, "stem (folded windows): cout must be a multiple of 32");
  MVG_REQUIRE(((long long)d->n * d->ho * (d->wo / 2)) % 64 == 0, "stem (folded windows): n * ho * wo / 2 must be a multiple of 64 (BatchNorm partials)");
  *rw = *d;
  rw->w = d->w / 4;
  rw->wo = d->wo / 2;
  rw->cin = 64;
  rw->cout = 2 * d->cout;
  rw->s = 1;
  return 0;
}

int mvg_stem_rowwindow_bf16(const float *x_nchw, void *xw, int64_t images, int h, int w, void *stream) {
  MVG_REQUIRE(x_nchw && xw && images > 0 && h > 0 && w > 0 && w % 4 == 0, "stem_rowwindow_bf16: bad arguments (width %% 4)");
  hipStream_t st = (hipStream_t)stream;
  const long long n = images * h * (w / 4) * 8;
  ProfScope ps(MVG_K_LAYOUT, st, 0.0, 12.0 * (double)images * h * w + 16.0 * (double)n);
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(stem_rowwindow_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x_nchw, (uint4 *)xw, n, h, w);
  return check_launch("stem_rowwindow_bf16");
}

int mvg_stem_fprop_bf16(const mvg_conv_desc *d, const void *xw, const void *w_fold, void *y, float *stats, void *stream) {
  mvg_conv_desc rw;
  if (stem_fold_desc(d, &rw)) return 2;
  return fprop_bf16_impl(&rw, xw, w_fold, y, nullptr, 0, stats, stream, false, 1, 0);
}

static mvg_conv_desc linear_desc_bf16(int rows, int fin, int fout) {
  mvg_conv_desc d = {1, rows, 1, 1, fin, fout, 1, 1, 1, 0, 1, 1};
  return d;
}

int mvg_linear_fprop_mixed(const float *x, const void *w_bf16, const float *bias, int relu, float *y, int rows, int fin, int fout,
                           void *stream) {
  const mvg_conv_desc d = linear_desc_bf16(rows, fin, fout);
  return fprop_bf16_impl(&d, x, w_bf16, y, bias, relu, nullptr, stream, true);
}

int mvg_linear_dgrad_mixed(const float *dy, const void *wt_bf16, const float *mask, const float *addend, float *dx, int rows,
                           int fin, int fout, void *stream) {
  const mvg_conv_desc d = linear_desc_bf16(rows, fin, fout);
  return dgrad_bf16_impl(&d, dy, wt_bf16, dx, mask, addend, stream, true);
}

static void wgrad_bf16_tile(const mvg_conv_desc *d, int &bm, int &bn) {
  const int ncols = d->r * d->s * d->cin;
  bm = d->cout >= 128 ? 128 : 64;
  bn = ncols >= 128 ? 128 : 64;
}

int mvg_conv_wgrad_splits_bf16(const mvg_conv_desc *d) {
  if (validate_bf16(d)) return -1;
  int bm, bn;
  wgrad_bf16_tile(d, bm, bn);
  const int ncols = d->r * d->s * d->cin;
  const long long tiles = (long long)ceil_div(d->cout, bm) * ceil_div(ncols, bn);
  const long long pixels = (long long)d->groups * d->n * d->ho * d->wo;
  const int cus = compute_cus();
  long long want = (2LL * cus) / tiles;                    // one resident round at two workgroups per CU
  long long maxs = pixels / 512;                           // at least 512 pixels (8 K-steps) per split
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

static int wgrad_bf16_impl(const mvg_conv_desc *d, const void *x, const void *dy, float *dw, float *db, float *workspace,
                           int splits, int accumulate, void *stream, bool f32in, int stride_w = -1, int pad_w = -1, bool slabs_only = false) {
  if (stride_w < 0 && validate_bf16(d)) return 2;
  const long long EB = f32in ? 4 : 2;
  MVG_REQUIRE(!db || f32in, "wgrad_bf16: the bias gradient rides on the fp32-operand form only");
  MVG_REQUIRE(splits >= 1, "wgrad_bf16: splits < 1");
  MVG_REQUIRE(splits == 1 || workspace != nullptr, "wgrad_bf16: workspace required for splits > 1");
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const float *)x;
  p.dy = (const float *)dy;
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
  p.pixels_per_split = ((p.pixels + splits - 1) / splits + 63) / 64 * 64;
  p.x_bytes = EB * d->groups * d->n * d->h * d->w * d->cin;
  p.ohw_div = make_fastdiv((unsigned)(d->ho * d->wo));
  p.wo_div = make_fastdiv((unsigned)d->wo);
  p.cin_div = make_fastdiv((unsigned)d->cin);
  p.s_div = make_fastdiv((unsigned)d->s);
  MVG_REQUIRE(p.pixels_per_split * d->cout * EB < 0x7FFFFFF0ll, "wgrad_bf16: split too large for 32-bit offsets");
  MVG_REQUIRE(EB * (p.pixels_per_split / (d->ho * d->wo) + 2) * d->h * d->w * d->cin < 0x7FFFFFF0ll,
              "wgrad_bf16: split too large for 32-bit offsets");
  int bm, bn;
  wgrad_bf16_tile(d, bm, bn);
  p.mtiles = ceil_div(d->cout, bm);
  p.ntiles = ceil_div(p.ncols, bn);
  p.out = splits == 1 ? dw : workspace;
  p.accumulate = (splits == 1) ? accumulate : 0;
  float *db_slab = workspace ? workspace + (size_t)splits * d->cout * p.ncols : nullptr;
  p.db = db ? (splits == 1 ? db : db_slab) : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  {
    const double acin = d->cin == 8 && d->r == 7 ? 3.0 : (double)d->cin;
    const double flops = 2.0 * (double)p.pixels * d->cout * d->r * d->s * acin * (stride_w >= 0 ? 147.0 / 448.0 : 1.0);
    const double bytes = 2.0 * ((double)d->groups * d->n * d->h * d->w * acin + (double)p.pixels * d->cout) +
                         4.0 * (double)d->cout * d->r * d->s * acin;
    ProfScope ps(lin ? MVG_K_LINEAR_WGRAD : MVG_K_CONV_WGRAD, st, flops, bytes);
    MVG_REQUIRE((long long)p.mtiles * p.ntiles * splits < (1LL << 31), "wgrad_bf16: grid too large");
    dim3 grid(p.mtiles * p.ntiles * splits), block(256);
    const bool incr = (long long)d->ho * d->wo >= 128;       // at most one image wrap per 64-pixel step
#define MVG_WGRAD_BF16(BM_, BN_)                                                                          \
  do {                                                                                                    \
    if (f32in) hipLaunchKernelGGL((wgrad_bf16_kernel<BM_, BN_, false, true>), grid, block, 0, st, p);     \
    else if (incr) hipLaunchKernelGGL((wgrad_bf16_kernel<BM_, BN_, true>), grid, block, 0, st, p);        \
    else hipLaunchKernelGGL((wgrad_bf16_kernel<BM_, BN_, false>), grid, block, 0, st, p);                 \
  } while (0)
    if (bm == 128 && bn == 128) MVG_WGRAD_BF16(128, 128);
    else if (bm == 64 && bn == 128) MVG_WGRAD_BF16(64, 128);
    else if (bm == 128 && bn == 64) MVG_WGRAD_BF16(128, 64);
    else MVG_WGRAD_BF16(64, 64);
#undef MVG_WGRAD_BF16
    if (check_launch("conv_wgrad_bf16")) return 1;
  }
  if (splits > 1 && !slabs_only) {
    const long long n = (long long)d->cout * p.ncols;
    MVG_REQUIRE(n % 4 == 0, "wgrad_bf16: weight elements %% 4 != 0");
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

int mvg_conv_wgrad_bf16(const mvg_conv_desc *d, const void *x, const void *dy, float *dw, float *workspace, int splits,
                        int accumulate, void *stream) {
  return wgrad_bf16_impl(d, x, dy, dw, nullptr, workspace, splits, accumulate, stream, false);
}

int mvg_conv_wgrad_bf16_slabs(const mvg_conv_desc *d, const void *x, const void *dy, float *workspace, int splits, void *stream) {
  MVG_REQUIRE(splits > 1 && workspace, "wgrad_bf16_slabs: splits > 1 and a workspace (the slabs ARE the result)");
  return wgrad_bf16_impl(d, x, dy, workspace, nullptr, workspace, splits, 0, stream, false, -1, -1, true);
}

int mvg_stem_wgrad_splits_bf16(const mvg_conv_desc *d) {
  mvg_conv_desc rw;
  if (stem_fold_desc(d, &rw)) return -1;
  const long long tiles = (long long)ceil_div(rw.cout, 128) * ceil_div(7 * 64, 128);
  const long long pixels = (long long)rw.groups * rw.n * rw.ho * rw.wo;
  long long want = (2LL * compute_cus()) / tiles, maxs = pixels / 512;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

int mvg_stem_wgrad_bf16(const mvg_conv_desc *d, const void *xw, const void *dy, float *dw_fold, float *workspace, int splits,
                        int accumulate, void *stream) {
  mvg_conv_desc rw;
  if (stem_fold_desc(d, &rw)) return 2;
  return wgrad_bf16_impl(&rw, xw, dy, dw_fold, nullptr, workspace, splits, accumulate, stream, false, 1, 0);
}

int mvg_linear_wgrad_mixed(const float *x, const float *dy, float *dw, float *db, int rows, int fin, int fout, float *workspace,
                           int splits, int accumulate, void *stream) {
  const mvg_conv_desc d = linear_desc_bf16(rows, fin, fout);
  return wgrad_bf16_impl(&d, x, dy, dw, db, workspace, splits, accumulate, stream, true);
}

}  // extern "C"
