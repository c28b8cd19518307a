// boxes.hip -- IoU, anchor grids, FPN level map and batched bitmask NMS for gfx950.
//
// Slots: core/bbox (/root/reference/README.md:17), core/anchor (README.md:16), ops (README.md:24).
// All arithmetic is mxdet_math.h's contraction-free fp32, so integer outputs (keep indices, levels)
// are bit-exact against oracle/mxdet_oracle.c.
#include "common.h"

namespace mxdet {

// ---------------------------------------------------------------------------------------------
__global__ void box_iou_kernel(const float4* __restrict__ a, int64_t na, const float4* __restrict__ b,
                               int64_t nb, float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= na * nb) return;
  int64_t i = idx / nb, j = idx - i * nb;
  float4 p = a[i], q = b[j];
  out[idx] = mxdet_iou(p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w);
}

__global__ void anchors_kernel(const float* __restrict__ base, int A, int H, int W, int stride,
                               float4* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)H * W * A;
  if (idx >= total) return;
  int a = (int)(idx % A);
  int64_t cell = idx / A;
  int x = (int)(cell % W), y = (int)(cell / W);
  float sx = (float)(x * stride), sy = (float)(y * stride);
  float4 o;
  o.x = base[a * 4 + 0] + sx;
  o.y = base[a * 4 + 1] + sy;
  o.z = base[a * 4 + 2] + sx;
  o.w = base[a * 4 + 3] + sy;
  out[idx] = o;
}

__global__ void fpn_level_kernel(const float* __restrict__ rois, int64_t R, int lvl_min, int lvl_max,
                                 int32_t* __restrict__ levels) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const float* r = rois + i * 5;
  int k = mxdet_fpn_level(r[1], r[2], r[3], r[4]);
  k = k < lvl_min ? lvl_min : (k > lvl_max ? lvl_max : k);
  levels[i] = k;
}

// ---------------------------------------------------------------------------------------------
// NMS, stage 1: suppression bitmask. One wave per (64-row block, 64-column block) pair of one list;
// lane t owns row box t and tests it against the 64 column boxes staged in LDS. Only the upper
// triangle (column block >= row block) is ever read by stage 2, so only that is computed.
__global__ void __launch_bounds__(64)
nms_mask_kernel(const float4* __restrict__ boxes, const int32_t* __restrict__ counts, int n_max,
                int nwords_max, float thresh, unsigned long long* __restrict__ mask) {
  const int col_blk = blockIdx.x, row_blk = blockIdx.y, b = blockIdx.z;
  if (col_blk < row_blk) return;
  int n = counts[b];
  n = n > n_max ? n_max : n;
  if (row_blk * 64 >= n || col_blk * 64 >= n) return;
  __shared__ float4 cb[64];
  const int t = threadIdx.x;
  const float4* bx = boxes + (int64_t)b * n_max;
  int cj = col_blk * 64 + t;
  if (cj < n) cb[t] = bx[cj];
  __syncthreads();
  int i = row_blk * 64 + t;
  if (i >= n) return;
  float4 r = bx[i];
  int ncol = n - col_blk * 64;
  ncol = ncol > 64 ? 64 : ncol;
  unsigned long long bits = 0;
  for (int j = 0; j < ncol; ++j) {
    int c = col_blk * 64 + j;
    float4 q = cb[j];
    float v = mxdet_iou(r.x, r.y, r.z, r.w, q.x, q.y, q.z, q.w);
    if (c > i && v > thresh) bits |= (1ull << j);
  }
  mask[((int64_t)b * n_max + i) * nwords_max + col_blk] = bits;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = (unsigned)__builtin_amdgcn_readlane((int)lo, lane);
  hi = (unsigned)__builtin_amdgcn_readlane((int)hi, lane);
  return ((unsigned long long)hi << 32) | lo;
}

// NMS, stage 2: the sequential keep scan, one wave per list. The `removed` bit vector (<= 64 words) lives in LDS.
// Per 64-box word: the diagonal 64x64 block (one coalesced load, prefetched one word ahead) resolves the word's keep
// bits in registers (v_readlane, no memory in the dependent chain); then every lane that owns a kept box ORs that
// box's mask row into the later words with LDS atomics -- a loop with a uniform trip count whose loads are all
// independent, instead of a data-dependent chain of dependent loads.
__global__ void __launch_bounds__(64)
nms_scan_kernel(const unsigned long long* __restrict__ mask, const int32_t* __restrict__ counts,
                const uint8_t* __restrict__ invalid, int n_max, int nwords_max, int max_keep,
                int32_t* __restrict__ keep_idx, int32_t* __restrict__ num_keep) {
  __shared__ unsigned long long rem[64];
  const int b = blockIdx.x, lane = threadIdx.x;
  int n = counts[b];
  n = n > n_max ? n_max : n;
  n = __builtin_amdgcn_readfirstlane(n);
  const int nw = (n + 63) >> 6;
  const unsigned long long* m = mask + (int64_t)b * n_max * nwords_max;
  // initial removed bits: invalid boxes and the tail beyond n
  for (int w = 0; w < nw; ++w) {
    int idx = w * 64 + lane;
    bool bad = (idx >= n) || (invalid != nullptr && invalid[(int64_t)b * n_max + idx] != 0);
    unsigned long long bm = __ballot(bad);
    if (lane == 0) rem[w] = bm;
  }
  __syncthreads();   // single-wave workgroup: orders the LDS writes above against the reads below
  unsigned long long mykeep = 0;
  unsigned long long diag_next = (nw > 0 && lane < n) ? m[(int64_t)lane * nwords_max] : 0ull;
  for (int w = 0; w < nw; ++w) {
    const unsigned long long diag = diag_next;
    if (w + 1 < nw) {
      int row = (w + 1) * 64 + lane;
      diag_next = (row < n) ? m[(int64_t)row * nwords_max + (w + 1)] : 0ull;
    }
    __syncthreads();   // the LDS atomics of the previous words have landed (one wave: a cheap s_barrier)
    unsigned long long cur = rem[w];
    cur = readlane64(cur, 0);
    unsigned long long keepbits = 0;
    for (int bb = 0; bb < 64; ++bb) {
      unsigned long long d = readlane64(diag, bb);
      if (!((cur >> bb) & 1ull)) {
        keepbits |= (1ull << bb);
        cur |= d;
      }
    }
    if (lane == w) mykeep = keepbits;
    if ((keepbits >> lane) & 1ull) {
      const unsigned long long* row = m + (int64_t)(w * 64 + lane) * nwords_max;
      for (int j = w + 1; j < nw; ++j) {
        unsigned long long v = row[j];
        if (v) atomicOr(&rem[j], v);
      }
    }
  }
  // exclusive scan of popcounts over lanes, then ordered emission
  int cnt = __popcll(mykeep);
  int incl = cnt;
  for (int off = 1; off < 64; off <<= 1) {
    int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  int excl = incl - cnt;
  int total = __shfl(incl, 63);
  unsigned long long kb = mykeep;
  int pos = excl;
  while (kb) {
    int bb = __ffsll((long long)kb) - 1;
    kb &= kb - 1;
    if (pos < max_keep) keep_idx[(int64_t)b * n_max + pos] = lane * 64 + bb;
    ++pos;
  }
  if (lane == 0) num_keep[b] = total < max_keep ? total : max_keep;
}

// Stage 2 for lists of at most NW*64 boxes (the RPN case: 2000 -> 32 words). Same algorithm as nms_scan_kernel with
// the global-memory latency taken off the dependent chain and without atomics:
//   * the mask rows of the 64 boxes of word w sit in LDS ([box][word], padded pitch), those of word w+1 are fetched --
//     all loads independent, issued back to back -- while word w is being resolved;
//   * lane L owns word L of the `removed` set in a register. After the 64-step resolve of word w, the kept boxes' rows
//     are OR-ed in by a wave-uniform loop over the kept bits: every lane reads ITS word of row i (consecutive lanes,
//     consecutive LDS words: conflict-free). The earlier form had each kept lane OR its row into shared words with
//     same-address LDS atomics -- up to 64 serialised operations per word, most of the kernel's 240 us.
template <int NW>
__global__ void __launch_bounds__(64)
nms_scan_rows_kernel(const unsigned long long* __restrict__ mask, const int32_t* __restrict__ counts,
                     const uint8_t* __restrict__ invalid, int n_max, int nwords_max, int max_keep,
                     int32_t* __restrict__ keep_idx, int32_t* __restrict__ num_keep) {
  __shared__ unsigned long long rows[2][64][NW + 1];
  const int b = blockIdx.x, lane = threadIdx.x;
  int n = counts[b];
  n = n > n_max ? n_max : n;
  n = __builtin_amdgcn_readfirstlane(n);
  const int nw = (n + 63) >> 6;
  const unsigned long long* m = mask + (int64_t)b * n_max * NW;   // pitch NW (nwords_max <= NW columns are in use)
  unsigned long long myrem = 0ull;               // lane L: word L of the removed set
  for (int w = 0; w < nw; ++w) {
    int idx = w * 64 + lane;
    bool bad = (idx >= n) || (invalid != nullptr && invalid[(int64_t)b * n_max + idx] != 0);
    unsigned long long bm = __ballot(bad);
    if (lane == w) myrem = bm;
  }
  // The mask pitch is NW words for this kernel (mxdet_nms_batched pads it), so the 64 rows of word w are one
  // contiguous run of 64*NW words: lane l fetches words l, l+64, ... of the run -- 512 contiguous bytes per wave
  // instruction at immediate offsets, no per-load address or predicate arithmetic (one row per lane made every
  // instruction touch 64 separate cache lines; predicated triangle loads cost ~35 instructions each). Entries the mask
  // kernel never wrote (lower triangle, words past the list, rows past n -- the buffer has 64 rows of slack) are
  // read but never used: rows of boxes >= n are never kept, lanes outside (w, nw) drop what they OR.
  unsigned long long nxt[NW];
  auto load_row = [&](unsigned long long (&r)[NW], int w) {
    const unsigned long long* bp = m + (int64_t)(w * 64) * NW + lane;
#pragma unroll
    for (int k = 0; k < NW; ++k) r[k] = bp[k * 64];
  };
  auto stage = [&](const unsigned long long (&r)[NW], int buf, int w) {
    unsigned long long* dst = &rows[buf][lane >> 5][lane & 31];
#pragma unroll
    for (int k = 0; k < NW; ++k) dst[k * 2 * (NW + 1)] = r[k];
  };
  load_row(nxt, 0);
  stage(nxt, 0, 0);
  __syncthreads();
  const int myword = lane < NW ? lane : 0;
  unsigned long long mykeep = 0;
  int buf = 0;
  for (int w = 0; w < nw; ++w) {
#ifndef NMS_ABL_NOLOAD
    if (w + 1 < nw) load_row(nxt, w + 1);
#endif
    const unsigned long long diag = rows[buf][lane][w];
    // Box bb of this word is kept iff bit bb of `c` is clear when its turn comes; a diagonal row only has bits above
    // its own index, so bit bb never changes afterwards and the keep mask is simply ~c at the end: the dependent
    // chain per box is readlane -> select -> or.
    unsigned long long c = readlane64(myrem, w);
#ifndef NMS_ABL_NOCHAIN
#pragma unroll
    for (int bb = 0; bb < 64; ++bb) {
      unsigned long long d = readlane64(diag, bb);
      c |= ((c >> bb) & 1ull) ? 0ull : d;
    }
#else
    c |= readlane64(diag, 0);
#endif
    const unsigned long long keepbits = ~c;
    if (lane == w) mykeep = keepbits;
    // eight rows per trip, all eight LDS reads in flight before the first OR (a loop over the kept bits alone waits out
    // the LDS latency once per box); rows of suppressed boxes are read and dropped
    const unsigned klo = __builtin_amdgcn_readfirstlane((unsigned)keepbits);
    const unsigned khi = __builtin_amdgcn_readfirstlane((unsigned)(keepbits >> 32));
    unsigned long long acc = 0ull;
#ifndef NMS_ABL_NOOR
#pragma unroll
#else
    if (klo == 0x12345u)
#endif
    for (int ch = 0; ch < 8; ++ch) {
      const unsigned bits = ((ch < 4 ? klo : khi) >> (8 * (ch & 3))) & 0xffu;
      if (bits == 0u) continue;
      unsigned long long v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = rows[buf][8 * ch + t][myword];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc |= ((bits >> t) & 1u) ? v[t] : 0ull;
    }
    if (lane > w && lane < nw) myrem |= acc;
    if (w + 1 < nw) stage(nxt, buf ^ 1, w + 1);
    __syncthreads();   // one wave: a cheap s_barrier; the staged rows of word w+1 are visible
    buf ^= 1;
  }
  int cnt = __popcll(mykeep);
  int incl = cnt;
  for (int off = 1; off < 64; off <<= 1) {
    int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  int excl = incl - cnt;
  int total = __shfl(incl, 63);
  unsigned long long kb = mykeep;
  int pos = excl;
  while (kb) {
    int bb = __ffsll((long long)kb) - 1;
    kb &= kb - 1;
    if (pos < max_keep) keep_idx[(int64_t)b * n_max + pos] = lane * 64 + bb;
    ++pos;
  }
  if (lane == 0) num_keep[b] = total < max_keep ? total : max_keep;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" int mxdet_box_iou(const float* boxes_a, int64_t na, const float* boxes_b, int64_t nb,
                             float* out, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(na >= 0 && nb >= 0, MXDET_ESHAPE, "box_iou: negative size");
  if (na == 0 || nb == 0) return MXDET_OK;
  MXDET_REQUIRE(boxes_a && boxes_b && out, MXDET_EINVAL, "box_iou: null pointer");
  int64_t total = na * nb;
  hipLaunchKernelGGL(box_iou_kernel, dim3((unsigned)ceil_div<int64_t>(total, 256)), dim3(256), 0,
                     as_stream(stream), (const float4*)boxes_a, na, (const float4*)boxes_b, nb, out);
  return check_launch("box_iou");
}

extern "C" int mxdet_generate_anchors(const float* base_anchors, int32_t A, int32_t H, int32_t W,
                                      int32_t stride, float* out, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(A > 0 && H >= 0 && W >= 0 && stride > 0, MXDET_ESHAPE, "generate_anchors: bad shape");
  int64_t total = (int64_t)H * W * A;
  if (total == 0) return MXDET_OK;
  MXDET_REQUIRE(base_anchors && out, MXDET_EINVAL, "generate_anchors: null pointer");
  hipLaunchKernelGGL(anchors_kernel, dim3((unsigned)ceil_div<int64_t>(total, 256)), dim3(256), 0,
                     as_stream(stream), base_anchors, A, H, W, stride, (float4*)out);
  return check_launch("generate_anchors");
}

extern "C" int mxdet_fpn_level_map(const float* rois, int64_t R, int32_t lvl_min, int32_t lvl_max,
                                   int32_t* levels, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(R >= 0 && lvl_min <= lvl_max, MXDET_ESHAPE, "fpn_level_map: bad arguments");
  if (R == 0) return MXDET_OK;
  MXDET_REQUIRE(rois && levels, MXDET_EINVAL, "fpn_level_map: null pointer");
  hipLaunchKernelGGL(fpn_level_kernel, dim3((unsigned)ceil_div<int64_t>(R, 256)), dim3(256), 0,
                     as_stream(stream), rois, R, lvl_min, lvl_max, levels);
  return check_launch("fpn_level_map");
}

extern "C" size_t mxdet_nms_batched_workspace_bytes(int32_t B, int32_t n_max) {
  if (B <= 0 || n_max <= 0) return 0;
  size_t nwords = (size_t)(n_max + 63) / 64;
  if (nwords <= 32) nwords = 32;     // the register-row scan reads rows at a fixed 32-word pitch, up to 64 rows past a list
  return ((size_t)B * n_max + 64) * nwords * sizeof(unsigned long long);
}

extern "C" int mxdet_nms_batched(const float* boxes, const int32_t* counts, const uint8_t* invalid,
                                 int32_t B, int32_t n_max, float thresh, int32_t max_keep,
                                 int32_t* keep_idx, int32_t* num_keep, void* workspace,
                                 size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(B >= 0 && n_max >= 0, MXDET_ESHAPE, "nms_batched: negative size");
  if (B == 0) return MXDET_OK;
  MXDET_REQUIRE(n_max <= 4096, MXDET_ESHAPE, "nms_batched: n_max %d > 4096 unsupported", n_max);
  MXDET_REQUIRE(counts && num_keep, MXDET_EINVAL, "nms_batched: null pointer");
  if (n_max == 0) {
    hipError_t e = zero_async(num_keep, sizeof(int32_t) * B, as_stream(stream));
    MXDET_REQUIRE(e == hipSuccess, MXDET_EHIP, "nms_batched: memset failed");
    return MXDET_OK;
  }
  MXDET_REQUIRE(boxes && keep_idx, MXDET_EINVAL, "nms_batched: null pointer");
  size_t need = mxdet_nms_batched_workspace_bytes(B, n_max);
  MXDET_REQUIRE(workspace && workspace_bytes >= need, MXDET_EWORKSPACE,
                "nms_batched: workspace %zu < %zu", workspace_bytes, need);
  int nwords = (n_max + 63) / 64;
  const int pitch = nwords <= 32 ? 32 : nwords;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nwords, nwords, B), dim3(64), 0, as_stream(stream),
                     (const float4*)boxes, counts, n_max, pitch, thresh,
                     (unsigned long long*)workspace);
  if (nwords <= 32)
    hipLaunchKernelGGL(nms_scan_rows_kernel<32>, dim3(B), dim3(64), 0, as_stream(stream),
                       (const unsigned long long*)workspace, counts, invalid, n_max, nwords,
                       max_keep < 0 ? 0 : max_keep, keep_idx, num_keep);
  else
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, as_stream(stream),
                       (const unsigned long long*)workspace, counts, invalid, n_max, nwords,
                       max_keep < 0 ? 0 : max_keep, keep_idx, num_keep);
  return check_launch("nms_batched");
}
