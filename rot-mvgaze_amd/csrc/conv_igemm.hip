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
#include <stdlib.h>

#include "common.h"

namespace mvg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmParams {
  const float *a;       // gathered operand (fprop: x, dgrad: dy)
  const float *b;       // weights, KRSC
  float *out;
  const float *bias;    // [ncols] or null (fprop)
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
};

// bijective XCD-aware remap of a 1-D grid (cdna_hip_programming.md §5 "XCD swizzle must be
// bijective"): blocks that are adjacent after the remap share an XCD L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const int base = (xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  return base + (orig >> 3);
}

template <int BM, int BN, int BK, int WGM, int WGN, bool DGRAD>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
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
  constexpr int B_ELEMS = DGRAD ? BK * LDB : BN * LDA;
  static_assert(BM % RPP == 0, "tile");
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;

  const int nwg = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int ntile = wg % p.ntiles;
  const int mt_all = wg / p.ntiles;
  const int g = mt_all / p.mtiles_per_group;
  const int mtile = mt_all - g * p.mtiles_per_group;

  // ---- A loader state -------------------------------------------------------------------
  const int a_kv = tid % KV;
  const int a_r0 = tid / KV;
  const float *a_base[A_PASSES];
  int a_y0[A_PASSES], a_x0[A_PASSES];
  bool a_ok[A_PASSES];
  const int ohw = p.out_h * p.out_w;
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const long long m = (long long)mtile * BM + a_r0 + i * RPP;
    a_ok[i] = m < p.rows_per_group;
    const int mm = a_ok[i] ? (int)m : 0;
    const int img = mm / ohw;
    const int rem = mm - img * ohw;
    const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
    if (DGRAD) {
      a_y0[i] = oy + p.pad;
      a_x0[i] = ox + p.pad;
    } else {
      a_y0[i] = oy * p.stride - p.pad;
      a_x0[i] = ox * p.stride - p.pad;
    }
    a_base[i] = p.a + ((long long)g * p.imgs_per_group + img) * p.src_img_stride;
  }

  float4 a_reg[A_PASSES];
  float4 b_reg[DGRAD ? B_PASSES_D : B_PASSES_F];

  auto load_tiles = [&](int kt) {
    // ---- A: one float4 (4 channels of one tap) per row pass
    const int k0 = kt * BK + a_kv * 4;
    int tap = 0, c = k0;
    if (p.rs > 1) {
      tap = k0 >> p.src_c_shift;
      c = k0 - (tap << p.src_c_shift);
    }
    const int fr = tap / p.s, fs = tap - fr * p.s;
    const bool kok = k0 < p.ktotal;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      int iy, ix;
      bool ok = a_ok[i] && kok;
      if (DGRAD) {
        const int ty = a_y0[i] - fr, tx = a_x0[i] - fs;
        ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & (p.stride - 1)) == 0;
        iy = ty >> p.stride_shift;
        ix = tx >> p.stride_shift;
        ok = ok && iy < p.src_h && ix < p.src_w;
      } else {
        iy = a_y0[i] + fr;
        ix = a_x0[i] + fs;
        ok = ok && (unsigned)iy < (unsigned)p.src_h && (unsigned)ix < (unsigned)p.src_w;
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4 *>(a_base[i] + ((long long)(iy * p.src_w + ix) * p.src_c + c));
      a_reg[i] = v;
    }
    // ---- B
    if (!DGRAD) {
#pragma unroll
      for (int i = 0; i < B_PASSES_F; ++i) {
        const int row = a_r0 + i * RPP;
        const int n = ntile * BN + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < BN && n < p.ncols && kok)
          v = *reinterpret_cast<const float4 *>(p.b + (long long)n * p.ktotal + k0);
        b_reg[i] = v;
      }
    } else {
      const int nv = tid % NV;
      const int ncol = ntile * BN + nv * 4;
#pragma unroll
      for (int i = 0; i < B_PASSES_D; ++i) {
        const int krow = tid / NV + i * KRPP;
        const int k = kt * BK + krow;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (krow < BK && k < p.ktotal && ncol < p.ncols) {
          int btap = 0, o = k;
          if (p.rs > 1) {
            btap = k >> p.src_c_shift;
            o = k - (btap << p.src_c_shift);
          }
          v = *reinterpret_cast<const float4 *>(p.b + ((long long)o * p.rs + btap) * p.cin + ncol);
        }
        b_reg[i] = v;
      }
    }
  };

  auto store_tiles = [&](int buf) {
    float *As = smem + buf * (A_ELEMS + B_ELEMS);
    float *Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<float4 *>(As + (a_r0 + i * RPP) * LDA + a_kv * 4) = a_reg[i];
    if (!DGRAD) {
#pragma unroll
      for (int i = 0; i < B_PASSES_F; ++i) {
        const int row = a_r0 + i * RPP;
        if (row < BN) *reinterpret_cast<float4 *>(Bs + row * LDA + a_kv * 4) = b_reg[i];
      }
    } else {
      const int nv = tid % NV;
#pragma unroll
      for (int i = 0; i < B_PASSES_D; ++i) {
        const int krow = tid / NV + i * KRPP;
        if (krow < BK) *reinterpret_cast<float4 *>(Bs + krow * LDB + nv * 4) = b_reg[i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int KT = (p.ktotal + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) load_tiles(kt + 1);
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
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------
  const long long row_base = (long long)mtile * BM + wm * WTM;
  const long long grow0 = (long long)g * p.rows_per_group;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = ntile * BN + wn * WTN + j * 32 + li;
    const bool cok = col < p.ncols;
    float bias = 0.f;
    if (!DGRAD && p.bias && cok) bias = p.bias[col];
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = row_base + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const bool ok = cok && row < p.rows_per_group;
        float v = acc[i][j][e];
        if (ok) {
          const long long off = (grow0 + row) * p.ncols + col;
          if (DGRAD) {
            if (p.mask) v = (p.mask[off] > 0.f) ? v : 0.f;
            if (p.addend) v += p.addend[off];
          } else {
            v += bias;
            if (p.relu) v = fmaxf(v, 0.f);
            csum += v;
          }
          p.out[off] = v;
        }
      }
    }
    if (!DGRAD && p.stats) {
      // per-wave partial: column sum and sum of squares centred on the partial's own mean
      long long cnt_ll = p.rows_per_group - row_base;
      const int cnt = cnt_ll <= 0 ? 0 : (cnt_ll > WTM ? WTM : (int)cnt_ll);
      csum += __shfl_xor(csum, 32, 64);
      const float mean = cnt > 0 ? csum / (float)cnt : 0.f;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long row = row_base + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row < p.rows_per_group) {
            const float dlt = acc[i][j][e] - mean;
            q += dlt * dlt;
          }
        }
      q += __shfl_xor(q, 32, 64);
      if (lh == 0 && cok) {
        const long long P = (long long)p.mtiles_per_group * WGM;
        const long long pi = (long long)mtile * WGM + wm;
        float *st = p.stats + (((long long)g * P + pi) * 2) * p.ncols;
        st[col] = csum;
        st[p.ncols + col] = q;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------
struct WgradParams {
  const float *x;   // [imgs][h][w][cin]
  const float *dy;  // [imgs][ho][wo][cout]
  float *out;       // [splits][cout][rs*cin] (or dw directly when splits == 1)
  int h, w, cin, cout, r, s, stride, pad, ho, wo;
  int ncols;        // rs*cin
  long long pixels; // imgs*ho*wo
  long long pixels_per_split;
  int mtiles, ntiles;
  int accumulate;   // only meaningful when splits == 1
};

template <int BM, int BN, int BK, int WGM, int WGN>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int MV = BM / 4, NVB = BN / 4;
  constexpr int A_KRPP = 256 / MV, B_KRPP = 256 / NVB;
  constexpr int A_PASSES = (BK + A_KRPP - 1) / A_KRPP, B_PASSES = (BK + B_KRPP - 1) / B_KRPP;
  constexpr int A_ELEMS = BK * LDA, B_ELEMS = BK * LDB;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int ntile = blockIdx.x % p.ntiles, mtile = blockIdx.x / p.ntiles;
  const int split = blockIdx.y;
  const long long m_begin = (long long)split * p.pixels_per_split;
  long long m_end = m_begin + p.pixels_per_split;
  if (m_end > p.pixels) m_end = p.pixels;

  // A: dy rows, contiguous along cout
  const int a_mv = tid % MV, a_k0 = tid / MV;
  const int a_col = mtile * BM + a_mv * 4;
  const bool a_cok = a_col < p.cout;
  // B: x gather; this thread's column (tap, c) is fixed
  const int b_nv = tid % NVB, b_k0 = tid / NVB;
  const int b_col = ntile * BN + b_nv * 4;
  const bool b_cok = b_col < p.ncols;
  const int b_tap = b_cok ? b_col / p.cin : 0;
  const int b_c = b_col - b_tap * p.cin;
  const int b_fr = b_tap / p.s, b_fs = b_tap - b_fr * p.s;
  const int ohw = p.ho * p.wo;

  float4 a_reg[A_PASSES], b_reg[B_PASSES];
  auto load_tiles = [&](long long m0) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int kr = a_k0 + i * A_KRPP;
      const long long m = m0 + kr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kr < BK && m < m_end && a_cok) v = *reinterpret_cast<const float4 *>(p.dy + m * p.cout + a_col);
      a_reg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int kr = b_k0 + i * B_KRPP;
      const long long m = m0 + kr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kr < BK && m < m_end && b_cok) {
        const long long img = m / ohw;
        const int rem = (int)(m - img * ohw);
        const int oy = rem / p.wo, ox = rem - oy * p.wo;
        const int iy = oy * p.stride - p.pad + b_fr, ix = ox * p.stride - p.pad + b_fs;
        if ((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w)
          v = *reinterpret_cast<const float4 *>(p.x + ((img * p.h + iy) * p.w + ix) * p.cin + b_c);
      }
      b_reg[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    float *As = smem + buf * (A_ELEMS + B_ELEMS);
    float *Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int kr = a_k0 + i * A_KRPP;
      if (kr < BK) *reinterpret_cast<float4 *>(As + kr * LDA + a_mv * 4) = a_reg[i];
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int kr = b_k0 + i * B_KRPP;
      if (kr < BK) *reinterpret_cast<float4 *>(Bs + kr * LDB + b_nv * 4) = b_reg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int KT = (int)((m_end - m_begin + BK - 1) / BK);
  if (KT > 0) {
    load_tiles(m_begin);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) load_tiles(m_begin + (long long)(kt + 1) * BK);
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
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
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

__global__ void wgrad_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ dw, long long n4, int splits,
                                    int accumulate) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 s = accumulate ? reinterpret_cast<const float4 *>(dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 v = reinterpret_cast<const float4 *>(slabs)[(long long)k * n4 + i];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    reinterpret_cast<float4 *>(dw)[i] = s;
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
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

struct TileChoice {
  int bm, bn;
};

// pick the largest tile that still gives >= 2 workgroups per CU (else the smallest).
static TileChoice choose_tile(long long rows_per_group, int groups, int ncols) {
  static int cus = 0;
  if (cus <= 0) {
    cus = mvg_device_cus();
    if (cus <= 0) cus = 256;
  }
  const TileChoice cand[3] = {{128, 128}, {128, 64}, {64, 64}};
  if (ncols <= 32) return {128, 32};
  for (int i = 0; i < 3; ++i) {
    if (cand[i].bn > 64 && ncols < 128) continue;
    const long long blocks = (long long)groups * ceil_div(rows_per_group, cand[i].bm) * ceil_div(ncols, cand[i].bn);
    if (blocks >= 2LL * cus) return cand[i];
  }
  return (ncols >= 64) ? TileChoice{64, 64} : TileChoice{128, 32};
}

template <bool DGRAD>
static int launch_igemm(IgemmParams &p, TileChoice t, hipStream_t st) {
  p.mtiles_per_group = ceil_div(p.rows_per_group, t.bm);
  p.ntiles = ceil_div(p.ncols, t.bn);
  const long long nblk = (long long)p.groups * p.mtiles_per_group * p.ntiles;
  MVG_REQUIRE(nblk < (1LL << 31), "conv: grid too large");
  dim3 grid((unsigned)nblk), block(256);
  static int bk = 0;
  if (bk == 0) {
    const char *e = getenv("MVG_BK");
    bk = (e && atoi(e) == 16) ? 16 : 32;
  }
  const bool k32 = bk == 32 && p.ktotal >= 64;
  if (t.bm == 128 && t.bn == 128) {
    if (k32) hipLaunchKernelGGL((igemm_kernel<128, 128, 32, 2, 2, DGRAD>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_kernel<128, 128, 16, 2, 2, DGRAD>), grid, block, 0, st, p);
  } else if (t.bm == 128 && t.bn == 64) {
    if (k32) hipLaunchKernelGGL((igemm_kernel<128, 64, 32, 2, 2, DGRAD>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_kernel<128, 64, 16, 2, 2, DGRAD>), grid, block, 0, st, p);
  } else if (t.bm == 64 && t.bn == 64) {
    if (k32) hipLaunchKernelGGL((igemm_kernel<64, 64, 32, 2, 2, DGRAD>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_kernel<64, 64, 16, 2, 2, DGRAD>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((igemm_kernel<128, 32, 16, 4, 1, DGRAD>), grid, block, 0, st, p);
  }
  return check_launch(DGRAD ? "conv_dgrad" : "conv_fprop");
}

static int wave_rows(TileChoice t) { return (t.bn == 32) ? 32 : t.bm / 2; }

}  // namespace mvg

using namespace mvg;

extern "C" {

int mvg_conv_stats_partials(const mvg_conv_desc *d, int32_t *rows_per_partial) {
  if (validate(d)) return -1;
  const long long rows = (long long)d->n * d->ho * d->wo;
  const TileChoice t = choose_tile(rows, d->groups, d->cout);
  const int wr = wave_rows(t);
  if (rows_per_partial) *rows_per_partial = wr;
  return ceil_div(rows, t.bm) * (t.bm / wr);
}

int mvg_conv_fprop(const mvg_conv_desc *d, const float *x, const float *wgt, float *y, const float *bias, int relu,
                   float *stats, void *stream) {
  if (validate(d)) return 2;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.a = x;
  p.b = wgt;
  p.out = y;
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
  p.stride_shift = d->stride == 2 ? 1 : 0;
  p.pad = d->pad;
  p.ktotal = d->r * d->s * d->cin;
  p.cin = d->cin;
  p.rows_per_group = (long long)d->n * d->ho * d->wo;
  p.src_img_stride = (long long)d->h * d->w * d->cin;
  p.imgs_per_group = d->n;
  const TileChoice t = choose_tile(p.rows_per_group, d->groups, d->cout);
  const double flops = 2.0 * d->groups * (double)p.rows_per_group * d->cout * p.ktotal;
  const double bytes = 4.0 * (d->groups * (double)d->n * d->h * d->w * d->cin + (double)d->cout * p.ktotal +
                              d->groups * (double)p.rows_per_group * d->cout);
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  ProfScope ps(lin ? MVG_K_LINEAR_FPROP : MVG_K_CONV_FPROP, (hipStream_t)stream, flops, bytes);
  return launch_igemm<false>(p, t, (hipStream_t)stream);
}

int mvg_conv_dgrad(const mvg_conv_desc *d, const float *dy, const float *wgt, float *dx, const float *mask,
                   const float *addend, void *stream) {
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
  p.rows_per_group = (long long)d->n * d->h * d->w;
  p.src_img_stride = (long long)d->ho * d->wo * d->cout;
  p.imgs_per_group = d->n;
  const TileChoice t = choose_tile(p.rows_per_group, d->groups, d->cin);
  // algorithmic flops: the transposed conv touches each (output pixel, tap) pair of the fprop once
  const double flops = 2.0 * d->groups * (double)d->n * d->ho * d->wo * d->cout * d->r * d->s * d->cin;
  const double bytes = 4.0 * (d->groups * (double)d->n * d->ho * d->wo * d->cout + (double)d->cout * d->r * d->s * d->cin +
                              d->groups * (double)p.rows_per_group * d->cin);
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  ProfScope ps(lin ? MVG_K_LINEAR_DGRAD : MVG_K_CONV_DGRAD, (hipStream_t)stream, flops, bytes);
  return launch_igemm<true>(p, t, (hipStream_t)stream);
}

static TileChoice wgrad_tile(const mvg_conv_desc *d) {
  const int ncols = d->r * d->s * d->cin;
  if (d->cout >= 128 && ncols >= 128) return {128, 128};
  if (d->cout >= 64 && ncols >= 128) return {64, 128};
  if (ncols >= 64 && d->cout >= 64) return {64, 64};
  if (d->cout < 64) return {32, 128};   // skinny cout (e.g. 2-wide head is handled elsewhere)
  return {128, 32};
}

int mvg_conv_wgrad_splits(const mvg_conv_desc *d) {
  if (validate(d)) return -1;
  const TileChoice t = wgrad_tile(d);
  const int ncols = d->r * d->s * d->cin;
  const long long tiles = (long long)ceil_div(d->cout, t.bm) * ceil_div(ncols, t.bn);
  const long long pixels = (long long)d->groups * d->n * d->ho * d->wo;
  int cus = mvg_device_cus();
  if (cus <= 0) cus = 256;
  long long want = (3LL * cus + tiles - 1) / tiles;      // ~3 workgroups per CU
  long long maxs = pixels / 256;                           // at least 256 pixels (16 K-steps) per split
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

int mvg_conv_wgrad(const mvg_conv_desc *d, const float *x, const float *dy, float *dw, float *workspace, int splits,
                   int accumulate, void *stream) {
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
  const TileChoice t = wgrad_tile(d);
  p.mtiles = ceil_div(d->cout, t.bm);
  p.ntiles = ceil_div(p.ncols, t.bn);
  p.out = splits == 1 ? dw : workspace;
  p.accumulate = (splits == 1) ? accumulate : 0;
  hipStream_t st = (hipStream_t)stream;
  const bool lin = d->r == 1 && d->s == 1 && d->h == 1 && d->w == 1;
  {
    const double flops = 2.0 * (double)p.pixels * d->cout * p.ncols;
    const double bytes = 4.0 * ((double)d->groups * d->n * d->h * d->w * d->cin + (double)p.pixels * d->cout +
                                (double)d->cout * p.ncols);
    ProfScope ps(lin ? MVG_K_LINEAR_WGRAD : MVG_K_CONV_WGRAD, st, flops, bytes);
    dim3 grid(p.mtiles * p.ntiles, splits), block(256);
    if (t.bm == 128 && t.bn == 128)
      hipLaunchKernelGGL((wgrad_kernel<128, 128, 16, 2, 2>), grid, block, 0, st, p);
    else if (t.bm == 64 && t.bn == 128)
      hipLaunchKernelGGL((wgrad_kernel<64, 128, 16, 2, 2>), grid, block, 0, st, p);
    else if (t.bm == 64 && t.bn == 64)
      hipLaunchKernelGGL((wgrad_kernel<64, 64, 16, 2, 2>), grid, block, 0, st, p);
    else if (t.bm == 32 && t.bn == 128)
      hipLaunchKernelGGL((wgrad_kernel<32, 128, 16, 1, 4>), grid, block, 0, st, p);
    else
      hipLaunchKernelGGL((wgrad_kernel<128, 32, 16, 4, 1>), grid, block, 0, st, p);
    if (check_launch("conv_wgrad")) return 1;
  }
  if (splits > 1) {
    const long long n = (long long)d->cout * p.ncols;
    MVG_REQUIRE(n % 4 == 0, "wgrad: weight elements %% 4 != 0");
    ProfScope ps(MVG_K_WGRAD_REDUCE, st, 0.0, 4.0 * n * (splits + 1));
    const int blocks = (int)((n / 4 + 255) / 256 > 2048 ? 2048 : (n / 4 + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, dw, n / 4, splits, accumulate);
    if (check_launch("wgrad_reduce")) return 1;
  }
  return 0;
}

}  // extern "C"
