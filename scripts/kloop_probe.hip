// K-loop sandbox for the split-operand GEMM kernels (conv_split.hip): a plain GEMM C[M][N] = A[M][K] . B[N][K]^T on s3
// operands with the production kernels' tile shapes, used to measure loop structures in isolation before they go
// into the implicit-GEMM kernels.  K runs over (tap, 32-channel block) of a [rows][C] s3 tensor like a 3x3 conv's
// (tap t shifts the row by toff[t]), so that the L2 / Infinity-Cache reuse of the operand matches a convolution's.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kloop_probe.hip -o gpurun_out/kloop_probe
//   gpurun_out/kloop_probe [M C N taps reps]
//
// Variants (selected by name on the command line or all):
//   small1     round 2's igemm_split_small_kernel loop: 128 x 128 x 32, 4 waves, ONE stage, 3 workgroups / CU,
//              piece-major LDS image, one DMA wave-instruction = 16 rows x 4 x 16 B of one piece (48-byte stride)
//   small1s    the same loop with SPAN DMA: a wave-instruction covers 64 consecutive 16-byte slots of [row][13 slots]
//              (12 data slots = the row's 192 contiguous bytes of all three pieces, one pad slot; 208-byte rows are
//              conflict-free for ds_read_b128)
//   big2       round 2's igemm_split_kernel loop: 256 x 128 x 32, 8 waves, two stages, 1 workgroup / CU
//   big2s      ... with SPAN DMA
//   big2si     ... SPAN DMA issued in between the MFMA groups
//   sw32_* / sw16_*   UNPADDED swizzled span image (the layout igemm_split16_kernel ships), 32x32x16 / 16x16x32 MFMAs
//   *_nodma / *_nomfma   floors: only the first K-step's DMA / one of the six products
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <string>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_vp;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ unsigned pred_off(unsigned off, bool ok) { return off | ((unsigned)(!ok) << 31); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, long long bytes) {
  unsigned n = bytes > 0x7FFFFFF0ll ? 0x7FFFFFF0u : (bytes < 0 ? 0u : (unsigned)bytes);
  const unsigned long long b = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  n = __builtin_amdgcn_readfirstlane(n);
  void *ub = (void *)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)n, 0x00020000);
}
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const int base = (xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  return base + (orig >> 3);
}

#define SPLIT_ONE(PA, PB, av, bv, acc)                                                                         \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA][i], bv[PB][j], acc[i][j], 0, 0, 0);
#define SPLIT_PRODUCTS(av, bv, acc)                                                                            \
  SPLIT_ONE(0, 2, av, bv, acc) SPLIT_ONE(2, 0, av, bv, acc) SPLIT_ONE(1, 1, av, bv, acc) SPLIT_ONE(0, 1, av, bv, acc) \
  SPLIT_ONE(1, 0, av, bv, acc) SPLIT_ONE(0, 0, av, bv, acc)

struct P {
  const unsigned short *a;   // [pad + M + pad][C] s3
  const unsigned short *b;   // [N][taps * C] s3
  float *c;                  // [M][N]
  int M, C, N, taps, KT;     // KT = taps * C / 32
  int pad_rows;
  int toff[9];
  int mtiles, ntiles;
  long long a_bytes, b_bytes;
};

// direct store of the accumulators (same for every variant; the probe runs long K)
template <int TM, int TN>
__device__ __forceinline__ void store_acc(const P &p, f32x16 (&acc)[TM][TN], int row0, int col0, int lane) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, cc = col0 + j * 32 + li;
        if (r < p.M && cc < p.N) p.c[(long long)r * p.N + cc] = acc[i][j][e];
      }
}

// ------------------------------------------------------------------------------------------------------------
// piece-major image, 16-row DMA groups (the shipped loaders)
// ------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int NW, int STAGES, int WPC>
__global__ __launch_bounds__(NW * 64, WPC) void k_piece(P p) {
  constexpr int BK = 32, WGN = 2, WGM = NW / 2;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32;
  constexpr int ROW = BK, A_ELEMS = BM * ROW, B_ELEMS = BN * ROW, STAGE = 3 * (A_ELEMS + B_ELEMS);
  constexpr int A_GROUPS = BM / 16 / NW, B_GROUPS_T = BN / 16;      // B groups in total
  __shared__ __attribute__((aligned(16))) unsigned short smem[STAGES * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN, li = lane & 31, lh = lane >> 5;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = wg % p.ntiles, mtile = wg / p.ntiles;
  const int r_in_grp = lane >> 2, a_kv = (lane & 3) ^ ((lane >> 4) & 3);
  unsigned a_base[A_GROUPS];
  bool a_ok[A_GROUPS];
#pragma unroll
  for (int i = 0; i < A_GROUPS; ++i) {
    const int m = mtile * BM + (wave + NW * i) * 16 + r_in_grp;
    a_ok[i] = m < p.M;
    a_base[i] = (unsigned)(m + p.pad_rows) * (unsigned)p.C * 6u + (unsigned)a_kv * 48u;
  }
  constexpr int B_PER = (B_GROUPS_T + NW - 1) / NW;
  unsigned b_base[B_PER];
  bool b_ok[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int n = ntile * BN + (wave + NW * i) * 16 + r_in_grp;
    b_ok[i] = (wave + NW * i < B_GROUPS_T) && n < p.N;
    b_base[i] = (unsigned)n * (unsigned)(p.taps * p.C) * 6u + (unsigned)a_kv * 48u;
  }
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes), rs_b = make_rsrc(p.b, p.b_bytes);
  const int cblks = p.C / 32;
  auto issue = [&](int kt, int buf) {
    const int tap = __builtin_amdgcn_readfirstlane(kt / cblks), cb = __builtin_amdgcn_readfirstlane(kt - tap * cblks);
    const unsigned sdelta = (unsigned)((p.toff[tap] * p.C + cb * 32) * 6);
    const unsigned kb = (unsigned)((tap * p.C + cb * 32) * 6);
    unsigned short *st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_GROUPS; ++i)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_vp)(st + pc * A_ELEMS + (wave + NW * i) * 16 * ROW), 16,
                                                 (int)pred_off(a_base[i] + sdelta + 16u * pc, a_ok[i]), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      if (wave + NW * i < B_GROUPS_T) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_vp)(st + 3 * A_ELEMS + pc * B_ELEMS + (wave + NW * i) * 16 * ROW), 16,
                                                   (int)pred_off(b_base[i] + kb + 16u * pc, b_ok[i]), 0, 0, 0);
      }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int a_row[TM], b_row[TN], a_sw[TM], b_sw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int R = wm * WTM + i * 32 + li;
    a_row[i] = R * ROW;
    a_sw[i] = (R >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int R = wn * WTN + j * 32 + li;
    b_row[j] = R * ROW;
    b_sw[j] = (R >> 2) & 3;
  }
  auto compute = [&](int buf) {
    const unsigned short *As = smem + buf * STAGE, *Bs = As + 3 * A_ELEMS;
#pragma unroll
    for (int kg = 0; kg < BK / 16; ++kg) {
      bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          av[pc][i] = *reinterpret_cast<const bf16x8 *>(As + pc * A_ELEMS + a_row[i] + (((2 * kg + lh) ^ a_sw[i]) << 3));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[pc][j] = *reinterpret_cast<const bf16x8 *>(Bs + pc * B_ELEMS + b_row[j] + (((2 * kg + lh) ^ b_sw[j]) << 3));
      }
      SPLIT_PRODUCTS(av, bv, acc)
    }
  };
  const int KT = p.KT;
  if (STAGES == 1) {
    for (int kt = 0; kt < KT; ++kt) {
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();
    }
  } else {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < KT) issue(kt + 1, cur ^ 1);
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  store_acc<TM, TN>(p, acc, mtile * BM + wm * WTM, ntile * BN + wn * WTN, lane);
}

// ------------------------------------------------------------------------------------------------------------
// span image: [row][13 slots of 16 B] (slots 0..11 = the row's 192 contiguous bytes: chunk cc, piece pc at slot
// 3 cc + pc; slot 12 = pad).  A wave-instruction Q covers linear slots 64 Q .. 64 Q + 63 of the (BM + BN) x 13 slot
// stage; wave w issues Q = w, w + NW, ...  MODE 0: all DMA issued at the top of the step; MODE 1: spread over the
// MFMA stream.  NODMA / NOMFMA: floors (only the first K-step's DMA / no MFMAs).
// ------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int NW, int STAGES, int WPC, int MODE, bool NODMA, bool NOMFMA>
__global__ __launch_bounds__(NW * 64, WPC) void k_span(P p) {
  constexpr int BK = 32, WGN = 2, WGM = NW / 2;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32;
  constexpr int SLOTS = 13, ROWB = SLOTS * 16;                          // bytes per LDS row
  constexpr int ROWS = BM + BN, STAGE_B = ROWS * ROWB;
  constexpr int NQ = (ROWS * SLOTS + 63) / 64;                          // DMA wave-instructions per stage
  constexpr int QPW = (NQ + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN, li = lane & 31, lh = lane >> 5;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = wg % p.ntiles, mtile = wg / p.ntiles;
  // per-lane source base of every instruction this wave issues (row-dependent part; the K-step adds a uniform delta)
  unsigned q_base[QPW];
  unsigned q_isb = 0;                // bit i: instruction i reads B (the uniform delta differs)
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int Q = wave + NW * i;
    const int s = Q * 64 + lane;
    const int row = s / SLOTS, slot = s - row * SLOTS;
    const bool isb = row >= BM;
    bool ok = Q < NQ && row < ROWS && slot < 12;
    unsigned base;
    if (!isb) {
      const int m = mtile * BM + row;
      ok = ok && m < p.M;
      base = (unsigned)(m + p.pad_rows) * (unsigned)p.C * 6u + 16u * slot;
    } else {
      const int n = ntile * BN + (row - BM);
      ok = ok && n < p.N;
      base = (unsigned)n * (unsigned)(p.taps * p.C) * 6u + 16u * slot;
    }
    q_base[i] = pred_off(base, ok);
    // rows BM-1 / BM can share an instruction: the delta is then per lane (keep it simple: per-lane select)
    q_isb |= (unsigned)isb << i;
  }
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes), rs_b = make_rsrc(p.b, p.b_bytes);
  const int cblks = p.C / 32;
  int is_tap = 0, is_cb = 0;
  unsigned d_a = 0, d_b = 0;
  auto step_delta = [&](int kt) {
    const int tap = __builtin_amdgcn_readfirstlane(kt / cblks), cb = __builtin_amdgcn_readfirstlane(kt - tap * cblks);
    d_a = (unsigned)((p.toff[tap] * p.C + cb * 32) * 6);
    d_b = (unsigned)((tap * p.C + cb * 32) * 6);
    (void)is_tap; (void)is_cb;
  };
  auto dma_one = [&](int i, int buf) {
    const int Q = wave + NW * i;
    if (Q < NQ) {
      const bool isb = (q_isb >> i) & 1u;
      // an instruction never straddles A and B when BM * 13 % 64 == 0 (128 x 13 = 26 x 64, 256 x 13 = 52 x 64)
      if (isb)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_vp)(smem + buf * STAGE_B + Q * 1024), 16, (int)(q_base[i] + d_b), 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_vp)(smem + buf * STAGE_B + Q * 1024), 16, (int)(q_base[i] + d_a), 0, 0, 0);
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int a_row[TM], b_row[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_row[i] = (wm * WTM + i * 32 + li) * ROWB;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_row[j] = (BM + wn * WTN + j * 32 + li) * ROWB;

  auto load_frags = [&](int buf, int kg, bf16x8 (&av)[3][TM], bf16x8 (&bv)[3][TN]) {
    const unsigned char *S = smem + buf * STAGE_B;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[pc][i] = *reinterpret_cast<const bf16x8 *>(S + a_row[i] + ((2 * kg + lh) * 3 + pc) * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[pc][j] = *reinterpret_cast<const bf16x8 *>(S + b_row[j] + ((2 * kg + lh) * 3 + pc) * 16);
    }
  };
  const int KT = p.KT;
  if (STAGES == 1) {
    for (int kt = 0; kt < KT; ++kt) {
      if (!NODMA || kt == 0) {
        step_delta(kt);
#pragma unroll
        for (int i = 0; i < QPW; ++i) dma_one(i, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        bf16x8 av[3][TM], bv[3][TN];
        load_frags(0, kg, av, bv);
        if (!NOMFMA) {
          SPLIT_PRODUCTS(av, bv, acc)
        } else {
          SPLIT_ONE(0, 0, av, bv, acc)
          asm volatile("" ::"v"(av[1][0]), "v"(av[2][0]), "v"(bv[1][0]), "v"(bv[2][0]));
        }
      }
      __syncthreads();
    }
  } else {
    step_delta(0);
#pragma unroll
    for (int i = 0; i < QPW; ++i) dma_one(i, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int cur = kt & 1;
      const bool more = (kt + 1 < KT) && !NODMA;
      if (more) step_delta(kt + 1);
      if (MODE == 0) {
        if (more) {
#pragma unroll
          for (int i = 0; i < QPW; ++i) dma_one(i, cur ^ 1);
        }
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
          bf16x8 av[3][TM], bv[3][TN];
          load_frags(cur, kg, av, bv);
          if (!NOMFMA) {
            SPLIT_PRODUCTS(av, bv, acc)
          } else {
            SPLIT_ONE(0, 0, av, bv, acc)
            asm volatile("" ::"v"(av[1][0]), "v"(av[2][0]), "v"(bv[1][0]), "v"(bv[2][0]));
          }
        }
      } else {
        // the step's DMA instructions go out in between the six product groups of the two k-groups (12 slots)
        constexpr int NSLOT = 12;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
          bf16x8 av[3][TM], bv[3][TN];
          load_frags(cur, kg, av, bv);
#define PROD_DMA(PA, PB, SL)                                                             \
  SPLIT_ONE(PA, PB, av, bv, acc)                                                         \
  if (more) {                                                                            \
    _Pragma("unroll") for (int i = 0; i < QPW; ++i) if (i * NSLOT / QPW == (SL)) dma_one(i, cur ^ 1); \
  }
          PROD_DMA(0, 2, kg * 6 + 0)
          PROD_DMA(2, 0, kg * 6 + 1)
          PROD_DMA(1, 1, kg * 6 + 2)
          PROD_DMA(0, 1, kg * 6 + 3)
          PROD_DMA(1, 0, kg * 6 + 4)
          PROD_DMA(0, 0, kg * 6 + 5)
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  store_acc<TM, TN>(p, acc, mtile * BM + wm * WTM, ntile * BN + wn * WTN, lane);
}


// ------------------------------------------------------------------------------------------------------------
// swizzled span image, NO padding: rows of 192 B = 12 slots, slot of (chunk cc, piece pc) in row R =
// 4 pc + (cc ^ g(R)); g(R) = (R >> 2) & 3 for the 32x32x16 fragments, (-(R >> 2)) & 3 for the 16x16x32 ones (both
// conflict-free for ds_read_b128: see DESIGN.md).  SHAPE16: v_mfma_f32_16x16x32_bf16 instead of 32x32x16.
// ------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int BM, int BN, int NW, int STAGES, int WPC, bool SHAPE16, bool NODMA>
__global__ __launch_bounds__(NW * 64, WPC) void k_sw(P p) {
  constexpr int WGN = 2, WGM = NW / 2;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int ROWB = 192, ROWS = BM + BN, STAGE_B = ROWS * ROWB;
  constexpr int NQ = ROWS * 12 / 64, QPW = (NQ + NW - 1) / NW;
  static_assert((BM * 12) % 64 == 0, "an instruction must not straddle A and B");
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = wg % p.ntiles, mtile = wg / p.ntiles;
  auto gsw = [](int R) { return SHAPE16 ? ((-(R >> 2)) & 3) : ((R >> 2) & 3); };
  unsigned q_base[QPW];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int Q = wave + NW * i;
    const int s = Q * 64 + lane;
    const int row = s / 12, j = s - row * 12;
    const int pc = j >> 2, cc = (j & 3) ^ gsw(row);
    const bool isb = row >= BM;
    bool ok = Q < NQ;
    unsigned base;
    if (!isb) {
      const int m = mtile * BM + row;
      ok = ok && m < p.M;
      base = (unsigned)(m + p.pad_rows) * (unsigned)p.C * 6u + 16u * (cc * 3 + pc);
    } else {
      const int n = ntile * BN + (row - BM);
      ok = ok && n < p.N;
      base = (unsigned)n * (unsigned)(p.taps * p.C) * 6u + 16u * (cc * 3 + pc);
    }
    q_base[i] = pred_off(base, ok);
  }
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes), rs_b = make_rsrc(p.b, p.b_bytes);
  const int cblks = p.C / 32;
  unsigned d_a = 0, d_b = 0;
  auto step_delta = [&](int kt) {
    const int tap = __builtin_amdgcn_readfirstlane(kt / cblks), cb = __builtin_amdgcn_readfirstlane(kt - tap * cblks);
    d_a = (unsigned)((p.toff[tap] * p.C + cb * 32) * 6);
    d_b = (unsigned)((tap * p.C + cb * 32) * 6);
  };
  auto dma_all = [&](int buf) {
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
      const int Q = wave + NW * i;
      if (Q < NQ) {
        if (Q * 64 >= BM * 12)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_vp)(smem + buf * STAGE_B + Q * 1024), 16, (int)(q_base[i] + d_b), 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_vp)(smem + buf * STAGE_B + Q * 1024), 16, (int)(q_base[i] + d_a), 0, 0, 0);
      }
    }
  };
  constexpr int TM = SHAPE16 ? WTM / 16 : WTM / 32, TN = SHAPE16 ? WTN / 16 : WTN / 32;
  typedef typename std::conditional<SHAPE16, f32x4, f32x16>::type acc_t;
  acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < (SHAPE16 ? 4 : 16); ++e) acc[i][j][e] = 0.f;
  // fragment addresses
  int a_off[TM], b_off[TN], a_g[TM], b_g[TN];
  const int lr = SHAPE16 ? (lane & 15) : (lane & 31);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int R = wm * WTM + i * (SHAPE16 ? 16 : 32) + lr;
    a_off[i] = R * ROWB;
    a_g[i] = gsw(R);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int R = wn * WTN + j * (SHAPE16 ? 16 : 32) + lr;
    b_off[j] = (BM + R) * ROWB;
    b_g[j] = gsw(R + BM);
  }
  auto compute = [&](int buf) {
    const unsigned char *S = smem + buf * STAGE_B;
    if constexpr (SHAPE16) {
      const int cc = lane >> 4;
      bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
        for (int i = 0; i < TM; ++i) av[pc][i] = *reinterpret_cast<const bf16x8 *>(S + a_off[i] + (4 * pc + (cc ^ a_g[i])) * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[pc][j] = *reinterpret_cast<const bf16x8 *>(S + b_off[j] + (4 * pc + (cc ^ b_g[j])) * 16);
      }
#define ONE16(PA, PB)                                                                              \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[PA][i], bv[PB][j], acc[i][j], 0, 0, 0);
      ONE16(0, 2) ONE16(2, 0) ONE16(1, 1) ONE16(0, 1) ONE16(1, 0) ONE16(0, 0)
    } else {
      const int lh = lane >> 5;
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
          for (int i = 0; i < TM; ++i) av[pc][i] = *reinterpret_cast<const bf16x8 *>(S + a_off[i] + (4 * pc + ((2 * kg + lh) ^ a_g[i])) * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j) bv[pc][j] = *reinterpret_cast<const bf16x8 *>(S + b_off[j] + (4 * pc + ((2 * kg + lh) ^ b_g[j])) * 16);
        }
        SPLIT_PRODUCTS(av, bv, acc)
      }
    }
  };
  const int KT = p.KT;
  if (STAGES == 1) {
    for (int kt = 0; kt < KT; ++kt) {
      if (!NODMA || kt == 0) {
        step_delta(kt);
        dma_all(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();
    }
  } else {
    step_delta(0);
    dma_all(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < KT && !NODMA) {
        step_delta(kt + 1);
        dma_all(cur ^ 1);
      }
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  if constexpr (SHAPE16) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = mtile * BM + wm * WTM + i * 16 + (lane >> 4) * 4 + e, cc = ntile * BN + wn * WTN + j * 16 + (lane & 15);
          if (r < p.M && cc < p.N) p.c[(long long)r * p.N + cc] = acc[i][j][e];
        }
  } else {
    store_acc<TM, TN>(p, acc, mtile * BM + wm * WTM, ntile * BN + wn * WTN, lane);
  }
}


// ------------------------------------------------------------------------------------------------------------
// TWO fp16 pieces per fp32 value (h1 = fp16(a), h2 = fp16(a - h1): |a - h1 - h2| <= 2^-24 |a|, half an fp32 ulp),
// v_mfma_f32_16x16x32_f16, NPROD = 3 (h1 g1, h1 g2, h2 g1) or 4 products.  Rows of 128 B = 8 slots, slot of
// (chunk cc, piece pc) in row R = (2 cc + pc) ^ h(R), h(R) = ((R >> 1) & 1) | (((R >> 2) & 1) << 2): conflict-free.
// ------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int BM, int BN, int NW, int WPC, int NPROD, bool NODMA>
__global__ __launch_bounds__(NW * 64, WPC) void k_h2(P p) {
  constexpr int WGN = 2, WGM = NW / 2;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 16, TN = WTN / 16;
  constexpr int ROWB = 128, ROWS = BM + BN, STAGE_B = ROWS * ROWB;
  constexpr int NQ = ROWS * 8 / 64, QPW = (NQ + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = wg % p.ntiles, mtile = wg / p.ntiles;
  auto hsw = [](int R) { return ((R >> 1) & 1) | (((R >> 2) & 1) << 2); };
  unsigned q_base[QPW];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int Q = wave + NW * i;
    const int s = Q * 64 + lane;
    const int row = s >> 3, j = (s & 7) ^ hsw(row);            // j = 2 cc + pc: the memory order of the row's 128-byte span
    const bool isb = row >= BM;
    bool ok = Q < NQ;
    unsigned base;
    if (!isb) {
      const int m = mtile * BM + row;
      ok = ok && m < p.M;
      base = (unsigned)(m + p.pad_rows) * (unsigned)p.C * 4u + 16u * j;
    } else {
      const int n = ntile * BN + (row - BM);
      ok = ok && n < p.N;
      base = (unsigned)n * (unsigned)(p.taps * p.C) * 4u + 16u * j;
    }
    q_base[i] = pred_off(base, ok);
  }
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes), rs_b = make_rsrc(p.b, p.b_bytes);
  const int cblks = p.C / 32;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  const int cc = lane >> 4;
  int a_off[TM][2], b_off[TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int R = wm * WTM + i * 16 + (lane & 15);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) a_off[i][pc] = R * ROWB + (((2 * cc + pc) ^ hsw(R)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int R = BM + wn * WTN + j * 16 + (lane & 15);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) b_off[j][pc] = R * ROWB + (((2 * cc + pc) ^ hsw(R)) << 4);
  }
  const int KT = p.KT;
  for (int kt = 0; kt < KT; ++kt) {
    if (!NODMA || kt == 0) {
      const int tap = __builtin_amdgcn_readfirstlane(kt / cblks), cb = __builtin_amdgcn_readfirstlane(kt - tap * cblks);
      const unsigned d_a = (unsigned)((p.toff[tap] * p.C + cb * 32) * 4), d_b = (unsigned)((tap * p.C + cb * 32) * 4);
#pragma unroll
      for (int i = 0; i < QPW; ++i) {
        const int Q = wave + NW * i;
        if (Q < NQ) {
          if (Q * 64 >= BM * 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_vp)(smem + Q * 1024), 16, (int)(q_base[i] + d_b), 0, 0, 0);
          else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_vp)(smem + Q * 1024), 16, (int)(q_base[i] + d_a), 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
      f16x8 av[2][TM], bv[2][TN];
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
        for (int i = 0; i < TM; ++i) av[pc][i] = *reinterpret_cast<const f16x8 *>(smem + a_off[i][pc]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[pc][j] = *reinterpret_cast<const f16x8 *>(smem + b_off[j][pc]);
      }
#define ONEH(PA, PB)                                                                               \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[PA][i], bv[PB][j], acc[i][j], 0, 0, 0);
      if (NPROD == 4) { ONEH(1, 1) }
      ONEH(0, 1) ONEH(1, 0) ONEH(0, 0)
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = mtile * BM + wm * WTM + i * 16 + (lane >> 4) * 4 + e, ccol = ntile * BN + wn * WTN + j * 16 + (lane & 15);
        if (r < p.M && ccol < p.N) p.c[(long long)r * p.N + ccol] = acc[i][j][e];
      }
}


// two fp16 pieces, TWO stages per workgroup (2 x 32 KB): the next K-step's DMA is issued before this step's MFMAs
template <int BM, int BN, int NW, int WPC, bool EARLY>
__global__ __launch_bounds__(NW * 64, WPC) void k_h2x2(P p) {
  constexpr int WGN = 2, WGM = NW / 2;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 16, TN = WTN / 16;
  constexpr int ROWB = 128, ROWS = BM + BN, STAGE_B = ROWS * ROWB;
  constexpr int NQ = ROWS * 8 / 64, QPW = (NQ + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = wg % p.ntiles, mtile = wg / p.ntiles;
  auto hsw = [](int R) { return ((R >> 1) & 1) | (((R >> 2) & 1) << 2); };
  unsigned q_base[QPW];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int Q = wave + NW * i;
    const int s = Q * 64 + lane;
    const int row = s >> 3, j = (s & 7) ^ hsw(row);
    const bool isb = row >= BM;
    bool ok = Q < NQ;
    unsigned base;
    if (!isb) {
      const int m = mtile * BM + row;
      ok = ok && m < p.M;
      base = (unsigned)(m + p.pad_rows) * (unsigned)p.C * 4u + 16u * j;
    } else {
      const int n = ntile * BN + (row - BM);
      ok = ok && n < p.N;
      base = (unsigned)n * (unsigned)(p.taps * p.C) * 4u + 16u * j;
    }
    q_base[i] = pred_off(base, ok);
  }
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes), rs_b = make_rsrc(p.b, p.b_bytes);
  const int cblks = p.C / 32;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  const int cc = lane >> 4;
  int a_off[TM][2], b_off[TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int R = wm * WTM + i * 16 + (lane & 15);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) a_off[i][pc] = R * ROWB + (((2 * cc + pc) ^ hsw(R)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int R = BM + wn * WTN + j * 16 + (lane & 15);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) b_off[j][pc] = R * ROWB + (((2 * cc + pc) ^ hsw(R)) << 4);
  }
  auto issue = [&](int kt, int buf) {
    const int tap = __builtin_amdgcn_readfirstlane(kt / cblks), cb = __builtin_amdgcn_readfirstlane(kt - tap * cblks);
    const unsigned d_a = (unsigned)((p.toff[tap] * p.C + cb * 32) * 4), d_b = (unsigned)((tap * p.C + cb * 32) * 4);
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
      const int Q = wave + NW * i;
      if (Q < NQ) {
        if (Q * 64 >= BM * 8)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_vp)(smem + buf * STAGE_B + Q * 1024), 16, (int)(q_base[i] + d_b), 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_vp)(smem + buf * STAGE_B + Q * 1024), 16, (int)(q_base[i] + d_a), 0, 0, 0);
      }
    }
  };
  auto compute = [&](int buf) {
    const unsigned char *S = smem + buf * STAGE_B;
    f16x8 av[2][TM], bv[2][TN];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[pc][i] = *reinterpret_cast<const f16x8 *>(S + a_off[i][pc]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[pc][j] = *reinterpret_cast<const f16x8 *>(S + b_off[j][pc]);
    }
    ONEH(0, 1) ONEH(1, 0) ONEH(0, 0)
  };
  const int KT = p.KT;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (EARLY && kt + 1 < KT) issue(kt + 1, cur ^ 1);
    compute(cur);
    if (!EARLY && kt + 1 < KT) issue(kt + 1, cur ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = mtile * BM + wm * WTM + i * 16 + (lane >> 4) * 4 + e, ccol = ntile * BN + wn * WTN + j * 16 + (lane & 15);
        if (r < p.M && ccol < p.N) p.c[(long long)r * p.N + ccol] = acc[i][j][e];
      }
}

// ------------------------------------------------------------------------------------------------------------
static unsigned short f2bf(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// fp32 [rows][C] -> s3
static void to_s3(const std::vector<float> &x, long long rows, int C, std::vector<unsigned short> &out) {
  out.resize((size_t)rows * C * 3);
  for (long long r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      float v = x[(size_t)r * C + c];
      unsigned short h1 = f2bf(v);
      float r1 = v - bf2f(h1);
      unsigned short h2 = f2bf(r1);
      float r2 = r1 - bf2f(h2);
      unsigned short h3 = f2bf(r2);
      size_t base = ((size_t)r * (C / 8) + c / 8) * 3 * 8 + c % 8;
      out[base] = h1;
      out[base + 8] = h2;
      out[base + 16] = h3;
    }
}

// fp32 [rows][C] -> s2h: 8-channel chunks, the two fp16 pieces of a chunk adjacent (4 bytes per element)
static void to_s2h(const std::vector<float> &x, long long rows, int C, std::vector<unsigned short> &out) {
  out.resize((size_t)rows * C * 2);
  for (long long r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      const float v = x[(size_t)r * C + c];
      const _Float16 h1 = (_Float16)v;
      const _Float16 h2 = (_Float16)(v - (float)h1);
      unsigned short u1, u2;
      memcpy(&u1, &h1, 2);
      memcpy(&u2, &h2, 2);
      const size_t base = ((size_t)r * (C / 8) + c / 8) * 2 * 8 + c % 8;
      out[base] = u1;
      out[base + 8] = u2;
    }
}

struct Variant {
  const char *name;
  void (*kern)(P);
  int bm, bn, threads;
  size_t lds;
  int fmt = 0;            // 0: s3 operands, 1: s2h operands
};

int main(int argc, char **argv) {
  int M = argc > 1 ? atoi(argv[1]) : 100352, C = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 256;
  int taps = argc > 4 ? atoi(argv[4]) : 9, reps = argc > 5 ? atoi(argv[5]) : 20;
  const char *only = argc > 6 ? argv[6] : nullptr;
  const int W = 14;
  P p;
  memset(&p, 0, sizeof(p));
  p.M = M; p.C = C; p.N = N; p.taps = taps; p.KT = taps * C / 32;
  p.pad_rows = 32;
  const int t9[9] = {-W - 1, -W, -W + 1, -1, 0, 1, W - 1, W, W + 1};
  for (int t = 0; t < 9; ++t) p.toff[t] = taps == 1 ? 0 : t9[t];
  const long long arows = (long long)M + 2 * p.pad_rows;
  std::vector<float> ha((size_t)arows * C), hb((size_t)N * taps * C);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
  for (auto &v : ha) v = rnd();
  for (auto &v : hb) v = rnd();
  std::vector<unsigned short> sa, sb;
  to_s3(ha, arows, C, sa);
  to_s3(hb, N, taps * C, sb);
  unsigned short *da, *db;
  float *dc;
  p.a_bytes = (long long)sa.size() * 2; p.b_bytes = (long long)sb.size() * 2;
  CHECK(hipMalloc(&da, p.a_bytes)); CHECK(hipMalloc(&db, p.b_bytes)); CHECK(hipMalloc(&dc, (size_t)M * N * 4));
  CHECK(hipMemcpy(da, sa.data(), p.a_bytes, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, sb.data(), p.b_bytes, hipMemcpyHostToDevice));
  p.a = da; p.b = db; p.c = dc;
  std::vector<unsigned short> ha2, hb2;
  to_s2h(ha, arows, C, ha2);
  to_s2h(hb, N, taps * C, hb2);
  unsigned short *da2, *db2;
  CHECK(hipMalloc(&da2, ha2.size() * 2)); CHECK(hipMalloc(&db2, hb2.size() * 2));
  CHECK(hipMemcpy(da2, ha2.data(), ha2.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db2, hb2.data(), hb2.size() * 2, hipMemcpyHostToDevice));
  const long long a_bytes3 = p.a_bytes, b_bytes3 = p.b_bytes;

  std::vector<Variant> vs = {
      {"small1", k_piece<128, 128, 4, 1, 3>, 128, 128, 256, 0},
      {"small1s", k_span<128, 128, 4, 1, 3, 0, false, false>, 128, 128, 256, 0},
      {"small1s_nodma", k_span<128, 128, 4, 1, 3, 0, true, false>, 128, 128, 256, 0},
      {"small1s_nomfma", k_span<128, 128, 4, 1, 3, 0, false, true>, 128, 128, 256, 0},
      {"sw32_small1", k_sw<128, 128, 4, 1, 3, false, false>, 128, 128, 256, 0},
      {"sw16_small1", k_sw<128, 128, 4, 1, 3, true, false>, 128, 128, 256, 0},
      {"sw32_small1_nodma", k_sw<128, 128, 4, 1, 3, false, true>, 128, 128, 256, 0},
      {"sw16_small1_nodma", k_sw<128, 128, 4, 1, 3, true, true>, 128, 128, 256, 0},
      {"sw32_big2", k_sw<256, 128, 8, 2, 1, false, false>, 256, 128, 512, 0},
      {"sw16_big2", k_sw<256, 128, 8, 2, 1, true, false>, 256, 128, 512, 0},
      {"sw32_big2_nodma", k_sw<256, 128, 8, 2, 1, false, true>, 256, 128, 512, 0},
      {"sw16_big2_nodma", k_sw<256, 128, 8, 2, 1, true, true>, 256, 128, 512, 0},
      {"h2_3prod_wpc3", k_h2<128, 128, 4, 3, 3, false>, 128, 128, 256, 0, 1},
      {"h2_4prod_wpc3", k_h2<128, 128, 4, 3, 4, false>, 128, 128, 256, 0, 1},
      {"h2_3prod_wpc4", k_h2<128, 128, 4, 4, 3, false>, 128, 128, 256, 0, 1},
      {"h2_3prod_wpc3_nodma", k_h2<128, 128, 4, 3, 3, true>, 128, 128, 256, 0, 1},
      // larger wave tiles (fewer LDS fragment reads per MFMA: 0.25 instead of 0.33), two workgroups per CU
      {"h2_256x128_4w_wpc2", k_h2<256, 128, 4, 2, 3, false>, 256, 128, 256, 0, 1},
      {"h2_128x256_4w_wpc2", k_h2<128, 256, 4, 2, 3, false>, 128, 256, 256, 0, 1},
      {"h2_256x256_8w_wpc1", k_h2<256, 256, 8, 1, 3, false>, 256, 256, 512, 0, 1},
      {"h2_256x128_4w_wpc2_nodma", k_h2<256, 128, 4, 2, 3, true>, 256, 128, 256, 0, 1},
      {"h2x2_wpc2_early", k_h2x2<128, 128, 4, 2, true>, 128, 128, 256, 0, 1},
      {"h2x2_wpc2_late", k_h2x2<128, 128, 4, 2, false>, 128, 128, 256, 0, 1},
      {"h2x2_256x128_wpc1", k_h2x2<256, 128, 8, 1, true>, 256, 128, 512, 0, 1},
      {"big2", k_piece<256, 128, 8, 2, 1>, 256, 128, 512, 0},
      {"big2s", k_span<256, 128, 8, 2, 1, 0, false, false>, 256, 128, 512, 0},
      {"big2si", k_span<256, 128, 8, 2, 1, 1, false, false>, 256, 128, 512, 0},
      {"big2s_nodma", k_span<256, 128, 8, 2, 1, 0, true, false>, 256, 128, 512, 0},
      {"big2s_nomfma", k_span<256, 128, 8, 2, 1, 0, false, true>, 256, 128, 512, 0},
  };
  // reference on a sample of outputs (fp64 of the fp32 values)
  std::vector<float> hc((size_t)M * N);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double flops = 2.0 * M * N * (double)taps * C;
  printf("M %d C %d N %d taps %d (K %d)  %.1f GFLOP per launch\n", M, C, N, taps, taps * C, flops / 1e9);
  for (auto &v : vs) {
    if (only) {                      // comma-separated substrings of variant names
      bool hit = false;
      std::string o(only);
      size_t pos = 0;
      while (pos <= o.size()) {
        size_t e = o.find(',', pos);
        if (e == std::string::npos) e = o.size();
        const std::string tok = o.substr(pos, e - pos);
        if (!tok.empty() && strstr(v.name, tok.c_str())) hit = true;
        pos = e + 1;
      }
      if (!hit) continue;
    }
    p.mtiles = (M + v.bm - 1) / v.bm; p.ntiles = (N + v.bn - 1) / v.bn;
    p.a = v.fmt ? da2 : da; p.b = v.fmt ? db2 : db;
    p.a_bytes = v.fmt ? (long long)ha2.size() * 2 : a_bytes3; p.b_bytes = v.fmt ? (long long)hb2.size() * 2 : b_bytes3;
    const int grid = p.mtiles * p.ntiles;
    CHECK(hipMemset(dc, 0, (size_t)M * N * 4));
    hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.threads), 0, 0, p);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hc.data(), dc, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int t = 0; t < 400; ++t) {
      const int r = (int)(((uint64_t)t * 2654435761ull) % M), cidx = (int)(((uint64_t)t * 40503ull + 7) % N);
      double ref = 0;
      for (int tp = 0; tp < taps; ++tp)
        for (int k = 0; k < C; ++k)
          ref += (double)ha[(size_t)(r + p.pad_rows + p.toff[tp]) * C + k] * (double)hb[((size_t)cidx * taps + tp) * C + k];
      maxerr = fmax(maxerr, fabs(ref - hc[(size_t)r * N + cidx]));
      maxref = fmax(maxref, fabs(ref));
    }
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.threads), 0, 0, p);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.threads), 0, 0, p);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double stage_bytes = (double)(v.bm + v.bn) * (v.fmt ? 128.0 : 192.0);
    printf("%-16s grid %6d  %8.3f ms  %7.1f TF/s  ingest %6.2f TB/s  err %.2e (max |ref| %.1f)\n", v.name, grid, ms, flops / ms / 1e9,
           stage_bytes * p.KT * grid / ms / 1e9, maxerr / maxref, maxref);
    fflush(stdout);
  }
  return 0;
}
