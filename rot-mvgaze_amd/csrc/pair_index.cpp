// Stereo pair index (host, integer-only, bit-exact): the idx_to_kv build of
// /root/reference/dataset/gaze.py:39-73 driven by CPython's global `random` stream
// (random.seed(int) = MT19937 init_by_array; random.choice = getrandbits rejection sampling).
// Sequential by construction (each draw's consumption of the stream depends on the previous
// rejections), so it stays on the host; O(rows) instead of the reference's O(rows^2).
#include <stdint.h>
#include <string.h>

#include "../../include/rotmvgaze.h"

namespace {
const int N = 624, M = 397;

void init_genrand(uint32_t *mt, uint32_t s) {
  mt[0] = s;
  for (int i = 1; i < N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

uint32_t next_u32(uint32_t *st) {
  uint32_t *mt = st;
  uint32_t &idx = st[N];
  if (idx >= (uint32_t)N) {
    for (int k = 0; k < N; ++k) {
      const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % N] & 0x7FFFFFFFu);
      mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
    }
    idx = 0;
  }
  uint32_t y = mt[idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9D2C5680u;
  y ^= (y << 15) & 0xEFC60000u;
  y ^= y >> 18;
  return y;
}

// Random._randbelow_with_getrandbits for n < 2^32
uint32_t randbelow(uint32_t *st, uint32_t n) {
  int k = 0;
  while ((n >> k) != 0) ++k;
  uint32_t r = next_u32(st) >> (32 - k);
  while (r >= n) r = next_u32(st) >> (32 - k);
  return r;
}

bool cam_selected(int tag, int cam) {
  const bool test_cam = (cam % 3) == 2;   // {2,5,...,17}
  return tag == 0 ? true : (tag == 1 ? !test_cam : test_cam);
}
}  // namespace

extern "C" {

int mvg_mt19937_seed(uint32_t *st, uint64_t seed) {
  if (!st) return 2;
  uint32_t key[2] = {(uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32)};
  const int klen = key[1] ? 2 : 1;
  uint32_t *mt = st;
  init_genrand(mt, 19650218u);
  int i = 1, j = 0;
  for (int k = (N > klen ? N : klen); k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    ++i;
    ++j;
    if (i >= N) {
      mt[0] = mt[N - 1];
      i = 1;
    }
    if (j >= klen) j = 0;
  }
  for (int k = N - 1; k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    ++i;
    if (i >= N) {
      mt[0] = mt[N - 1];
      i = 1;
    }
  }
  mt[0] = 0x80000000u;
  st[N] = N;
  return 0;
}

int64_t mvg_pair_index_build(uint32_t *st, const int64_t *file_rows, int n_files, int camera_tag, int64_t *out,
                             int64_t capacity) {
  if (!st || !file_rows || !out || camera_tag < 0 || camera_tag > 2) return -1;
  int64_t cnt = 0;
  for (int f = 0; f < n_files; ++f) {
    const int64_t n = file_rows[f];
    for (int64_t idx = 0; idx < n; ++idx) {
      if (!cam_selected(camera_tag, (int)(idx % 18))) continue;
      const int64_t start = (idx / 18) * 18;
      int64_t cand[18];
      uint32_t nc = 0;
      for (int64_t i = start; i < start + 18 && i < n; ++i)
        if (i != idx && cam_selected(camera_tag, (int)(i % 18))) cand[nc++] = i;
      if (nc == 0) continue;
      const int64_t partner = cand[randbelow(st, nc)];
      if (cnt >= capacity) return -1;
      out[cnt * 3 + 0] = f;
      out[cnt * 3 + 1] = idx;
      out[cnt * 3 + 2] = partner;
      ++cnt;
    }
  }
  return cnt;
}

}  // extern "C"
