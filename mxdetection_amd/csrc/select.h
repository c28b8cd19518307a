// select.h -- workgroup-level "k smallest (key, index)" selection and LDS bitonic sort.
//
// One 1024-thread workgroup (16 wave64s) owns one list (an image, or an image x pyramid level).
// Selection is an 8-bit radix select over 32-bit keys through an LDS histogram (2 or 4 passes),
// followed by one pass in index order that resolves ties on the threshold key by a block-wide
// ballot/popcount prefix (so "ties -> lower index first" is exact, which bf16 scores need: many
// anchors share one bf16 logit). Integer-only, so results are bit-exact against the C oracle.
#pragma once
#include "common.h"

namespace mxdet {

struct SelectSmem {
  unsigned hist[256];
  int scratch[32];
  unsigned prefix;
  int remaining;
  int flag_all;
};

// keyf(i, key&) -> bool : is element i a candidate, and its key (smaller = preferred). Only the top
// `nbits` (16 or 32) bits of keys may be non-zero.
// emitf(i, chosen, key) is called exactly once for every candidate, in a pass that walks i in
// ascending order chunk by chunk.
template <class KeyF, class EmitF>
__device__ inline void block_select_smallest(int n, int k, int nbits, KeyF keyf, EmitF emitf,
                                             SelectSmem& sm) {
  const int tid = threadIdx.x, nt = blockDim.x;
  unsigned prefix = 0, mask = 0;
  int remaining = k < 0 ? 0 : k;
  bool all = false;
  for (int shift = 24; shift >= 32 - nbits; shift -= 8) {
    for (int i = tid; i < 256; i += nt) sm.hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
      unsigned kv;
      if (keyf(i, kv) && (kv & mask) == prefix) atomicAdd(&sm.hist[(kv >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int cum = 0, b = 0;
      for (; b < 256; ++b) {
        int c = (int)sm.hist[b];
        if (cum + c >= remaining) break;
        cum += c;
      }
      if (b == 256) {
        sm.flag_all = 1;  // fewer candidates than requested: take all
      } else {
        sm.flag_all = 0;
        sm.prefix = prefix | ((unsigned)b << shift);
        sm.remaining = remaining - cum;
      }
    }
    __syncthreads();
    if (sm.flag_all) {
      all = true;
      break;
    }
    prefix = sm.prefix;
    remaining = sm.remaining;
    mask |= 255u << shift;
  }
  __syncthreads();
  const unsigned T = prefix;
  const int need_eq = remaining;
  int eq_base = 0;
  for (int base = 0; base < n; base += nt) {
    int i = base + tid;
    unsigned kv = 0;
    bool c = (i < n) && keyf(i, kv);
    bool is_eq = c && !all && kv == T;
    int tot;
    int r = block_excl_count(is_eq, sm.scratch, &tot);
    if (c) {
      bool chosen = all || kv < T || (is_eq && (eq_base + r) < need_eq);
      emitf(i, chosen, kv);
    }
    eq_base += tot;
  }
  __syncthreads();
}

// Sort P (power of two) 64-bit keys in LDS in DESCENDING order with the whole workgroup.
__device__ inline void block_bitonic_sort_desc(unsigned long long* a, int P) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < (P >> 1); t += nt) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool desc = ((lo & size) == 0);
        unsigned long long x = a[lo], y = a[hi];
        if ((x < y) == desc) {
          a[lo] = y;
          a[hi] = x;
        }
      }
    }
  }
  __syncthreads();
}

}  // namespace mxdet
