// select.h -- workgroup-level "k smallest (key, index)" selection and LDS bitonic sort.
//
// One 1024-thread workgroup (16 wave64s) owns one list (an image, or an image x pyramid level).
// Selection is an 8-bit radix select through an LDS histogram: 2 or 4 passes over the 32-bit keys
// find the threshold key T; if the elements equal to T are not all taken (ties -- common for bf16
// logits, where thousands of anchors share one value) 4 more passes over the *index* digits of those
// elements find the index threshold IT. An element is chosen iff
//       key < T  ||  (key == T && index <= IT)
// which is exactly "the k smallest by (key, index)" -- a pure predicate, so the caller applies it in
// parallel (no ordered compaction pass). Integer-only: bit-exact against the C oracle.
#pragma once
#include "common.h"

namespace mxdet {

struct SelectSmem {
  __attribute__((aligned(16))) unsigned hist[256];
  int scratch[32];
  unsigned prefix;
  int remaining;
  int flag_all;
  int bin_count;
};

// hist[bin] += 1 for every lane with ok. Radix digits of detection scores are heavily concentrated (the top byte of
// a logit key takes a handful of values), and same-address LDS atomics serialise 64-fold inside a wave: two rounds
// of ballot aggregation (the first active lane's bin, then the next remaining one) take the dominant bins with one
// atomic each; whatever is left is spread and goes the plain way.
__device__ __forceinline__ void hist_add_agg(unsigned* hist, bool ok, unsigned bin) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    unsigned long long act = __ballot(ok);
    if (act == 0ull) return;
    int leader = __ffsll((long long)act) - 1;
    unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
    bool same = ok && bin == b0;
    unsigned long long sm_ = __ballot(same);
    if (lane == leader) atomicAdd(&hist[b0], (unsigned)__popcll(sm_));
    ok = ok && !same;
  }
  if (ok) atomicAdd(&hist[bin], 1u);
}

// One wave finds the first bin b of a 256-bin histogram with hist[b] > 0 and cum(b-1) + hist[b] >= remaining (the scan
// a single thread did serially: 256 dependent LDS reads per radix pass). Lane l owns bins 4l..4l+3; returns through
// the wave (all 64 lanes must call): bin (256 = none: fewer candidates than requested), candidates before it, its
// count, and the histogram total.
__device__ __forceinline__ void wave_scan_bins(const unsigned* hist, int remaining, int* bin, int* cum_before,
                                               int* bin_count, int* total) {
  const int lane = threadIdx.x & 63;
  const uint4 v = *(const uint4*)(hist + lane * 4);
  const int c[4] = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  const int s = c[0] + c[1] + c[2] + c[3];
  int incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  const int excl = incl - s;
  *total = __shfl(incl, 63);
  // the serial rule `cum + c >= remaining && c > 0`: with remaining <= 0 it is the first non-empty bin
  const int need = remaining > 1 ? remaining : 1;
  const bool here = s > 0 && excl < need && incl >= need;
  const unsigned long long m = __ballot(here);
  if (m == 0ull) { *bin = 256; *cum_before = incl; *bin_count = 0; return; }
  const int src = __ffsll((long long)m) - 1;
  int b = 0, cb = 0, bc = 0;
  if (lane == src) {
    int cum = excl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (bc == 0 && c[j] > 0 && cum + c[j] >= need) { b = lane * 4 + j; cb = cum; bc = c[j]; }
      cum += c[j];
    }
  }
  *bin = __shfl(b, src); *cum_before = __shfl(cb, src); *bin_count = __shfl(bc, src);
}

struct SelectResult {
  unsigned T, IT;
  int mode;    // 0: predicate above, 1: every candidate chosen, 2: none chosen
  int n_cand;  // number of candidates seen
  int remaining, eq_count;  // of the eq_count elements with key == T, the `remaining` smallest indices are chosen
  __device__ __forceinline__ bool chosen(unsigned key, unsigned idx) const {
    return mode == 1 || (mode == 0 && (key < T || (key == T && idx <= IT)));
  }
};

// keyf(i, key&) -> bool : is element i a candidate, and its key (smaller = preferred). Only the top
// `nbits` (16 or 32) bits of keys may be non-zero. All threads must call with the same arguments.
// idxf(i) -> the index used for tie-breaking (defaults to the position i; compacted candidate lists pass the
// original element index so that the result does not depend on the order of the list).
struct IdentityIdx { __device__ __forceinline__ unsigned operator()(int i) const { return (unsigned)i; } };

// Among the elements with key == T, the `remaining`-th smallest index (4 radix passes over the index digits).
template <class KeyF, class IdxF>
__device__ inline unsigned block_tie_threshold(int n, int remaining, unsigned T, KeyF keyf, IdxF idxf, SelectSmem& sm) {
  const int tid = threadIdx.x, nt = blockDim.x;
  unsigned ip = 0, im = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += nt) sm.hist[i] = 0;
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += 4 * nt) {
      unsigned kv[4], ix[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int i = i0 + u * nt;
        ok[u] = (i < n) && keyf(i, kv[u]);
        ix[u] = (i < n) ? idxf(i) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        hist_add_agg(sm.hist, ok[u] && kv[u] == T && (ix[u] & im) == ip, (ix[u] >> shift) & 255u);
    }
    __syncthreads();
    if (tid < 64) {
      int b, cum, bc, tot;
      wave_scan_bins(sm.hist, remaining, &b, &cum, &bc, &tot);
      if (tid == 0) {
        sm.prefix = ip | ((unsigned)b << shift);
        sm.remaining = remaining - cum;
      }
    }
    __syncthreads();
    ip = sm.prefix;
    remaining = sm.remaining;
    im |= 255u << shift;
  }
  __syncthreads();
  return ip;
}

template <class KeyF, class IdxF = IdentityIdx>
__device__ inline SelectResult block_select_threshold(int n, int k, int nbits, KeyF keyf, SelectSmem& sm,
                                                      IdxF idxf = IdxF(), bool resolve_ties = true) {
  const int tid = threadIdx.x, nt = blockDim.x;
  SelectResult res;
  res.T = 0; res.IT = 0xffffffffu; res.mode = 0; res.n_cand = 0; res.remaining = 0; res.eq_count = 0;
  unsigned prefix = 0, mask = 0;
  int remaining = k < 0 ? 0 : k;
  int eq_count = 0;
  bool first = true;
  for (int shift = 24; shift >= 32 - nbits; shift -= 8) {
    for (int i = tid; i < 256; i += nt) sm.hist[i] = 0;
    __syncthreads();
    // 4 independent key fetches in flight per thread before the LDS atomics (the pass is latency bound)
    for (int i0 = tid; i0 < n; i0 += 4 * nt) {
      unsigned kv[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int i = i0 + u * nt;
        ok[u] = (i < n) && keyf(i, kv[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        hist_add_agg(sm.hist, ok[u] && (kv[u] & mask) == prefix, (kv[u] >> shift) & 255u);
    }
    __syncthreads();
    if (tid < 64) {
      int b, cum, bc, total;
      wave_scan_bins(sm.hist, remaining, &b, &cum, &bc, &total);
      if (tid == 0) {
        if (first) sm.scratch[0] = total;
        if (b == 256) {
          sm.flag_all = 1;  // fewer candidates than requested
        } else {
          sm.flag_all = 0;
          sm.prefix = prefix | ((unsigned)b << shift);
          sm.remaining = remaining - cum;
          sm.bin_count = bc;
        }
      }
    }
    __syncthreads();
    if (first) res.n_cand = sm.scratch[0];
    first = false;
    if (sm.flag_all) {
      res.mode = 1;
      __syncthreads();
      return res;
    }
    prefix = sm.prefix;
    remaining = sm.remaining;
    eq_count = sm.bin_count;
    mask |= 255u << shift;
  }
  __syncthreads();
  if (k <= 0) {
    res.mode = 2;
    return res;
  }
  res.T = prefix;
  res.remaining = remaining;
  res.eq_count = eq_count;
  if (remaining >= eq_count || !resolve_ties) return res;  // every element equal to T is taken: IT stays at max
  res.IT = block_tie_threshold(n, remaining, prefix, keyf, idxf, sm);
  return res;
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
