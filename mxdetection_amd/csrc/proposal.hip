// proposal.hip -- pyramid RPN proposal generation on gfx950.
//
// Slot: rpn_heads + ops (/root/reference/README.md:28, :24); MXNet role contrib.Proposal /
// MultiProposal / the lineage's pyramid-proposal CustomOp (README.md:37). Pipeline, all on device:
//   (1) per (image, level): chip-wide radix-select of the top-k logits -> LDS bitonic sort by (score desc, index asc)
//   (2) decode + clip + min-size flag for the selected anchors
//   (3) batched bitmask NMS (boxes.hip)
//   (4) per image: rank-merge of the per-level kept lists (binary searches in LDS), cut to top-N
// Indices are integer-exact against the oracle; boxes are bit-exact (mxdet_math.h decode).
#include "common.h"
#include "select.h"

namespace mxdet {

constexpr int kMaxPre = 4096;  // pre_nms_top_n upper bound (LDS sort buffer = 32 KiB)

struct PyramidDev {
  int num_levels, A, dtype;
  int H[8], W[8], stride[8];
  const void* cls[8];
  const void* reg[8];
  long long cls_sn[8], cls_sy[8], cls_sx[8], cls_sa[8];
  long long reg_sn[8], reg_sy[8], reg_sx[8], reg_sc[8];
  const float* base[8];
  long long level_offset[8];  // global anchor index of the level's first anchor
  int classes;                // >= 1: (anchor, class) pairs share the anchor's deltas and base box
};

__device__ __forceinline__ float pyr_score(const PyramidDev& p, int l, int n, int local) {
  int A = p.A;
  int a = local % A;
  int cell = local / A;
  int x = cell % p.W[l], y = cell / p.W[l];
  long long off = n * p.cls_sn[l] + y * p.cls_sy[l] + x * p.cls_sx[l] + a * p.cls_sa[l];
  return load_as_f32(p.cls[l], off, p.dtype);
}

// ---- (0)+(1): chip-wide radix selection of the pre-NMS top-k of every (image, level) -------------------------------
// One workgroup per list (the earlier form) spends 200+ us streaming the 201,600 P2 keys through a single CU, seven
// times. Here every pass is a grid over 4096-key chunks of all lists, a few microseconds each:
//   gather  : logits (arbitrary strides) -> dense order-preserving keys [N][A_total]; zeroes the selection state
//   hist p  : 8-bit digit histogram of pass p (2 passes for bf16 logits, 4 for f32), LDS per workgroup, then one
//             global atomic per non-empty bin. The digit prefix found by the earlier passes is re-derived by every
//             workgroup from the global histograms (one wave, 256 bins) instead of a kernel of its own.
//   select  : keys above the threshold T go to the list's unsorted output; the INDICES of keys equal to T (ties --
//             thousands of anchors share one bf16 logit) go to a tie list
//   finish  : one workgroup per list: the `remaining` smallest tie indices join the output (radix passes over the tie
//             list only), LDS bitonic sort by (score desc, index asc), write keys + count
// Selection is integer-only and equals "the k smallest by (~key, index)": bit-exact against the oracle.
constexpr int kSelChunk = 4096;    // keys per workgroup (256 threads x 16)
constexpr int kMaxPasses = 4;

struct SelDev {
  unsigned* hist;     // [kMaxPasses][B][256]
  int* cnt;           // [B][2]: n_sel, n_tie
  unsigned* ties;     // [N][A_total], a list's ties at its level offset
  int npass;          // 2 (bf16) or 4 (f32)
  unsigned keymask;
  int B, pre_n;
};

__global__ void proposal_gather_kernel(PyramidDev p, int N, long long A_total, unsigned* __restrict__ fkeys,
                                       SelDev sd) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long long)kMaxPasses * sd.B * 256) sd.hist[idx] = 0u;
  if (idx < (long long)sd.B * 2) sd.cnt[idx] = 0;
  if (idx >= (long long)N * A_total) return;
  int n = (int)(idx / A_total);
  long long g = idx - (long long)n * A_total;
  int l = 0;
  while (l + 1 < p.num_levels && g >= p.level_offset[l + 1]) ++l;
  fkeys[idx] = mxdet_float_key(pyr_score(p, l, n, (int)(g - p.level_offset[l])));
}

// Threshold state after `npass_done` histogram passes, derived by one wave from the global histograms.
struct SelState {
  unsigned prefix, mask;
  int remaining, eq_count, all;   // all: fewer candidates than requested -> everything is chosen
};

__device__ inline SelState sel_resolve(const SelDev& sd, int b, int npass_done, SelState* sh) {
  if (threadIdx.x < 64) {
    SelState st;
    st.prefix = 0; st.mask = 0; st.remaining = sd.pre_n; st.eq_count = 0; st.all = 0;
    for (int ps = 0; ps < npass_done && !st.all; ++ps) {
      const int shift = 24 - 8 * ps;
      int bin, cum, bc, tot;
      wave_scan_bins(sd.hist + ((size_t)ps * sd.B + b) * 256, st.remaining, &bin, &cum, &bc, &tot);
      if (bin == 256) { st.all = 1; break; }
      st.prefix |= (unsigned)bin << shift;
      st.mask |= 255u << shift;
      st.remaining -= cum;
      st.eq_count = bc;
    }
    if (threadIdx.x == 0) *sh = st;
  }
  __syncthreads();
  return *sh;
}

__global__ void __launch_bounds__(256)
proposal_hist_kernel(PyramidDev p, long long A_total, const unsigned* __restrict__ fkeys, SelDev sd, int pass) {
  __shared__ unsigned h[256];
  __shared__ SelState sh;
  const int l = blockIdx.y, n = blockIdx.z;
  const int nl = p.H[l] * p.W[l] * p.A;
  const int c0 = blockIdx.x * kSelChunk;
  if (c0 >= nl) return;
  const int b = n * p.num_levels + l;
  h[threadIdx.x] = 0u;
  SelState st = sel_resolve(sd, b, pass, &sh);   // contains the barrier that publishes h[] = 0
  if (st.all) return;
  const unsigned* fk_l = fkeys + (long long)n * A_total + p.level_offset[l];
  const int shift = 24 - 8 * pass;
  const int c1 = c0 + kSelChunk < nl ? c0 + kSelChunk : nl;
  for (int i0 = c0 + threadIdx.x; i0 < c1; i0 += 4 * 256) {
    unsigned kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = i0 + u * 256;
      kv[u] = i < c1 ? (~fk_l[i]) & sd.keymask : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      hist_add_agg(h, (i0 + u * 256 < c1) && (kv[u] & st.mask) == st.prefix, (kv[u] >> shift) & 255u);
  }
  __syncthreads();
  unsigned v = h[threadIdx.x];
  if (v) atomicAdd(&sd.hist[((size_t)pass * sd.B + b) * 256 + threadIdx.x], v);
}

__global__ void __launch_bounds__(256)
proposal_select_kernel(PyramidDev p, long long A_total, const unsigned* __restrict__ fkeys, SelDev sd,
                       unsigned long long* __restrict__ keys) {
  __shared__ SelState sh;
  const int l = blockIdx.y, n = blockIdx.z;
  const int nl = p.H[l] * p.W[l] * p.A;
  const int c0 = blockIdx.x * kSelChunk;
  if (c0 >= nl) return;
  const int b = n * p.num_levels + l;
  const SelState st = sel_resolve(sd, b, sd.npass, &sh);
  const bool split_ties = !st.all && st.remaining < st.eq_count;
  const unsigned goff = (unsigned)p.level_offset[l];
  const unsigned* fk_l = fkeys + (long long)n * A_total + p.level_offset[l];
  unsigned* ties = sd.ties + (long long)n * A_total + p.level_offset[l];
  unsigned long long* out = keys + (long long)b * sd.pre_n;
  const int c1 = c0 + kSelChunk < nl ? c0 + kSelChunk : nl;
  // Chosen keys / tie indices of this chunk are first collected in LDS (LDS atomics), then ONE global atomic per
  // workgroup and list reserves the output run: same-address global atomics from every wave of the grid serialise
  // at the L2 and were most of this kernel's time.
  __shared__ unsigned long long s_take[kSelChunk];
  __shared__ unsigned s_tie[kSelChunk];
  __shared__ int s_nt, s_ni, s_bt, s_bi;
  if (threadIdx.x == 0) { s_nt = 0; s_ni = 0; }
  __syncthreads();
  for (int i = c0 + (int)threadIdx.x; i < c1; i += 256) {
    const unsigned fk = fk_l[i];
    const unsigned kv = (~fk) & sd.keymask;
    if (st.all || kv < st.prefix || (kv == st.prefix && !split_ties))
      s_take[atomicAdd(&s_nt, 1)] = ((unsigned long long)fk << 32) | (unsigned long long)(0xffffffffu - (goff + (unsigned)i));
    else if (split_ties && kv == st.prefix)
      s_tie[atomicAdd(&s_ni, 1)] = (unsigned)i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s_bt = s_nt ? atomicAdd(&sd.cnt[b * 2 + 0], s_nt) : 0;
    s_bi = s_ni ? atomicAdd(&sd.cnt[b * 2 + 1], s_ni) : 0;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < s_nt; j += 256) out[s_bt + j] = s_take[j];
  for (int j = threadIdx.x; j < s_ni; j += 256) ties[s_bi + j] = s_tie[j];
}

__global__ void __launch_bounds__(1024)
proposal_finish_kernel(PyramidDev p, long long A_total, const unsigned* __restrict__ fkeys, SelDev sd, int Kpad,
                       unsigned long long* __restrict__ keys, int32_t* __restrict__ counts) {
  __shared__ SelectSmem sm;
  __shared__ SelState sh;
  __shared__ unsigned long long list[kMaxPre];
  __shared__ int n_sel;
  const int l = blockIdx.x, n = blockIdx.y;
  const int b = n * p.num_levels + l;
  const SelState st = sel_resolve(sd, b, sd.npass, &sh);
  const bool split_ties = !st.all && st.remaining < st.eq_count;
  const int got = sd.cnt[b * 2 + 0], n_tie = sd.cnt[b * 2 + 1];
  unsigned long long* out = keys + (long long)b * sd.pre_n;
  for (int i = threadIdx.x; i < Kpad; i += blockDim.x) list[i] = i < got ? out[i] : 0ull;
  if (threadIdx.x == 0) n_sel = got;
  __syncthreads();
  if (split_ties) {
    const unsigned goff = (unsigned)p.level_offset[l];
    const unsigned* fk_l = fkeys + (long long)n * A_total + p.level_offset[l];
    const unsigned* ties = sd.ties + (long long)n * A_total + p.level_offset[l];
    const unsigned T = st.prefix;
    auto tkey = [&](int, unsigned& kv) -> bool { kv = T; return true; };
    auto tidx = [&](int i) -> unsigned { return ties[i]; };
    const unsigned IT = block_tie_threshold(n_tie, st.remaining, T, tkey, tidx, sm);
    for (int j = threadIdx.x; j < n_tie; j += blockDim.x) {
      unsigned i = ties[j];
      if (i <= IT) {
        int pos = atomicAdd(&n_sel, 1);
        list[pos] = ((unsigned long long)fk_l[i] << 32) | (unsigned long long)(0xffffffffu - (goff + i));
      }
    }
    __syncthreads();
  }
  block_bitonic_sort_desc(list, Kpad);
  const int cnt = n_sel;
  for (int i = threadIdx.x; i < sd.pre_n; i += blockDim.x) out[i] = list[i];
  if (threadIdx.x == 0) counts[b] = cnt;
}

// (2) decode the selected anchors
__global__ void proposal_decode_kernel(PyramidDev p, int N, int pre_n, const float* __restrict__ im_info,
                                       float min_size, const unsigned long long* __restrict__ keys,
                                       const int32_t* __restrict__ counts, float4* __restrict__ boxes,
                                       uint8_t* __restrict__ invalid) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int B = N * p.num_levels;
  if (idx >= B * pre_n) return;
  int b = idx / pre_n, j = idx - b * pre_n;
  int n = b / p.num_levels, l = b - n * p.num_levels;
  if (j >= counts[b]) {
    boxes[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    invalid[idx] = 1;
    return;
  }
  unsigned long long key = keys[idx];
  unsigned gidx = 0xffffffffu - (unsigned)(key & 0xffffffffull);
  int local = (int)(gidx - (unsigned)p.level_offset[l]);
  int A = p.A;
  int a = local % A;
  int cell = local / A;
  int x = cell % p.W[l], y = cell / p.W[l];
  a /= p.classes;
  const float* base = p.base[l] + a * 4;
  float sx = (float)(x * p.stride[l]), sy = (float)(y * p.stride[l]);
  float ax1 = base[0] + sx, ay1 = base[1] + sy, ax2 = base[2] + sx, ay2 = base[3] + sy;
  long long off = n * p.reg_sn[l] + y * p.reg_sy[l] + x * p.reg_sx[l];
  float d0 = load_as_f32(p.reg[l], off + (a * 4 + 0) * p.reg_sc[l], p.dtype);
  float d1 = load_as_f32(p.reg[l], off + (a * 4 + 1) * p.reg_sc[l], p.dtype);
  float d2 = load_as_f32(p.reg[l], off + (a * 4 + 2) * p.reg_sc[l], p.dtype);
  float d3 = load_as_f32(p.reg[l], off + (a * 4 + 3) * p.reg_sc[l], p.dtype);
  float o[4];
  mxdet_decode_clip(ax1, ay1, ax2, ay2, d0, d1, d2, d3, im_info[n * 3 + 0], im_info[n * 3 + 1], o);
  float ms = min_size * im_info[n * 3 + 2];
  float w = o[2] - o[0] + 1.0f, h = o[3] - o[1] + 1.0f;
  boxes[idx] = make_float4(o[0], o[1], o[2], o[3]);
  invalid[idx] = (w < ms || h < ms) ? 1 : 0;
}

// (4) merge levels of one image. Kept keys of every level are staged in LDS (each list is already
// sorted descending); an element's final rank is its own position plus, for every other level, the
// number of that level's keys greater than it (keys are unique: they embed the anchor index).
__global__ void __launch_bounds__(1024)
proposal_merge_kernel(int L, int pre_n, int lvl_cap, int post_n,
                      const unsigned long long* __restrict__ keys,
                      const float4* __restrict__ boxes, const int32_t* __restrict__ keep_idx,
                      const int32_t* __restrict__ num_keep, float* __restrict__ rois,
                      float* __restrict__ roi_scores, int32_t* __restrict__ roi_anchor,
                      int32_t* __restrict__ num_rois) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* lk = (unsigned long long*)smem_raw;  // [L][lvl_cap]
  __shared__ int nk[8];
  // grid (L, N): every workgroup stages all of the image's lists (the binary searches need them) and ranks the
  // elements of ONE level; a single workgroup per image left the other 254 CUs idle for 90-180 us on the critical path
  const int n = blockIdx.y, l_own = blockIdx.x;
  if (threadIdx.x < L) {
    int v = num_keep[n * L + threadIdx.x];
    nk[threadIdx.x] = v > lvl_cap ? lvl_cap : v;
  }
  __syncthreads();
  int total = 0;
  for (int l = 0; l < L; ++l) total += nk[l];
  for (int l = 0; l < L; ++l) {
    int b = n * L + l;
    for (int j = threadIdx.x; j < nk[l]; j += blockDim.x)
      lk[l * lvl_cap + j] = keys[(long long)b * pre_n + keep_idx[(long long)b * pre_n + j]];
  }
  __syncthreads();
  {
    const int l = l_own;
    int b = n * L + l;
    for (int j = threadIdx.x; j < nk[l]; j += blockDim.x) {
      unsigned long long e = lk[l * lvl_cap + j];
      int rank = j;
      for (int l2 = 0; l2 < L; ++l2) {
        if (l2 == l) continue;
        // count of keys in list l2 that are > e (list sorted descending)
        int lo = 0, hi = nk[l2];
        const unsigned long long* q = lk + l2 * lvl_cap;
        while (lo < hi) {
          int mid = (lo + hi) >> 1;
          if (q[mid] > e) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
      if (rank < post_n) {
        int pos = keep_idx[(long long)b * pre_n + j];
        float4 bx = boxes[(long long)b * pre_n + pos];
        float* r = rois + ((long long)n * post_n + rank) * 5;
        r[0] = (float)n; r[1] = bx.x; r[2] = bx.y; r[3] = bx.z; r[4] = bx.w;
        unsigned fk = (unsigned)(e >> 32);
        unsigned u = (fk & 0x80000000u) ? (fk & 0x7fffffffu) : ~fk;
        roi_scores[(long long)n * post_n + rank] = __uint_as_float(u);
        roi_anchor[(long long)n * post_n + rank] = (int32_t)(0xffffffffu - (unsigned)(e & 0xffffffffull));
      }
    }
  }
  if (l_own != 0) return;
  int nout = total < post_n ? total : post_n;
  for (int j = nout + threadIdx.x; j < post_n; j += blockDim.x) {
    float* r = rois + ((long long)n * post_n + j) * 5;
    r[0] = (float)n; r[1] = 0.f; r[2] = 0.f; r[3] = 0.f; r[4] = 0.f;
    roi_scores[(long long)n * post_n + j] = 0.f;
    roi_anchor[(long long)n * post_n + j] = -1;
  }
  if (threadIdx.x == 0) num_rois[n] = nout;
}

struct ProposalWs {
  unsigned* fkeys;
  unsigned* hist;
  int* cnt;
  unsigned* ties;
  unsigned long long* keys;
  int32_t* counts;
  float4* boxes;
  uint8_t* invalid;
  int32_t* keep_idx;
  int32_t* num_keep;
  void* nms_ws;
  size_t nms_bytes;
  size_t total;
};

static ProposalWs carve(void* base, int B, int pre_n, int N, long long A_total) {
  ProposalWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  char* p = (char*)base;
  w.fkeys = (unsigned*)(p + take((size_t)N * A_total * 4));
  w.hist = (unsigned*)(p + take((size_t)kMaxPasses * B * 256 * 4));
  w.cnt = (int*)(p + take((size_t)B * 2 * 4));
  w.ties = (unsigned*)(p + take((size_t)N * A_total * 4));
  w.keys = (unsigned long long*)(p + take((size_t)B * pre_n * 8));
  w.counts = (int32_t*)(p + take((size_t)B * 4));
  w.boxes = (float4*)(p + take((size_t)B * pre_n * 16));
  w.invalid = (uint8_t*)(p + take((size_t)B * pre_n));
  w.keep_idx = (int32_t*)(p + take((size_t)B * pre_n * 4));
  w.num_keep = (int32_t*)(p + take((size_t)B * 4));
  w.nms_bytes = mxdet_nms_batched_workspace_bytes(B, pre_n);
  w.nms_ws = (void*)(p + take(w.nms_bytes));
  w.total = off;
  return w;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" size_t mxdet_proposal_workspace_bytes(const mxdet_pyramid_t* p, int32_t N,
                                                 int32_t pre_nms_top_n) {
  if (!p || N <= 0 || pre_nms_top_n <= 0 || p->num_levels <= 0) return 0;
  long long At = 0;
  for (int l = 0; l < p->num_levels; ++l) At += (long long)p->H[l] * p->W[l] * p->A;
  return carve(nullptr, N * p->num_levels, pre_nms_top_n, N, At).total;
}

extern "C" int mxdet_proposal(const mxdet_pyramid_t* p, int32_t N, const float* im_info,
                              int32_t pre_nms_top_n, int32_t post_nms_top_n, float nms_thresh,
                              float min_size, float* rois, float* roi_scores, int32_t* roi_anchor,
                              int32_t* num_rois, void* workspace, size_t workspace_bytes,
                              mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(p != nullptr, MXDET_EINVAL, "proposal: null pyramid");
  MXDET_REQUIRE(N > 0 && p->num_levels > 0 && p->num_levels <= 8 && p->A > 0, MXDET_ESHAPE,
                "proposal: bad N/levels/A");
  MXDET_REQUIRE(pre_nms_top_n > 0 && pre_nms_top_n <= kMaxPre, MXDET_ESHAPE,
                "proposal: pre_nms_top_n %d outside (0,%d]", pre_nms_top_n, kMaxPre);
  MXDET_REQUIRE(post_nms_top_n > 0 && post_nms_top_n <= pre_nms_top_n * p->num_levels, MXDET_ESHAPE,
                "proposal: bad post_nms_top_n %d", post_nms_top_n);
  MXDET_REQUIRE(p->dtype == MXDET_DTYPE_F32 || p->dtype == MXDET_DTYPE_BF16, MXDET_EINVAL,
                "proposal: bad dtype");
  MXDET_REQUIRE(im_info && rois && roi_scores && roi_anchor && num_rois, MXDET_EINVAL,
                "proposal: null pointer");
  const int L = p->num_levels, B = N * L;
  long long A_total = 0;
  for (int l = 0; l < L; ++l) A_total += (long long)p->H[l] * p->W[l] * p->A;
  ProposalWs w = carve(workspace, B, pre_nms_top_n, N, A_total);
  MXDET_REQUIRE(workspace && workspace_bytes >= w.total, MXDET_EWORKSPACE,
                "proposal: workspace %zu < %zu", workspace_bytes, w.total);
  int per_level_post = post_nms_top_n < pre_nms_top_n ? post_nms_top_n : pre_nms_top_n;
  size_t merge_lds = (size_t)L * per_level_post * 8;
  MXDET_REQUIRE(merge_lds <= 150 * 1024, MXDET_ESHAPE, "proposal: levels*post_nms_top_n too large");
  PyramidDev d;
  memset(&d, 0, sizeof(d));
  d.num_levels = L; d.A = p->A; d.dtype = p->dtype;
  d.classes = p->classes > 1 ? p->classes : 1;
  MXDET_REQUIRE(p->A % d.classes == 0, MXDET_ESHAPE, "proposal: A=%d is not a multiple of classes=%d", p->A, d.classes);
  long long off = 0;
  for (int l = 0; l < L; ++l) {
    MXDET_REQUIRE(p->H[l] > 0 && p->W[l] > 0 && p->stride[l] > 0 && p->cls[l] && p->reg[l] &&
                      p->base_anchors[l],
                  MXDET_EINVAL, "proposal: level %d incomplete", l);
    MXDET_REQUIRE((long long)p->H[l] * p->W[l] * p->A < (1ll << 30), MXDET_ESHAPE,
                  "proposal: level %d too large", l);
    d.H[l] = p->H[l]; d.W[l] = p->W[l]; d.stride[l] = p->stride[l];
    d.cls[l] = p->cls[l]; d.reg[l] = p->reg[l];
    d.cls_sn[l] = p->cls_sn[l]; d.cls_sy[l] = p->cls_sy[l]; d.cls_sx[l] = p->cls_sx[l]; d.cls_sa[l] = p->cls_sa[l];
    d.reg_sn[l] = p->reg_sn[l]; d.reg_sy[l] = p->reg_sy[l]; d.reg_sx[l] = p->reg_sx[l]; d.reg_sc[l] = p->reg_sc[l];
    d.base[l] = p->base_anchors[l];
    d.level_offset[l] = off;
    off += (long long)p->H[l] * p->W[l] * p->A;
  }
  int Kpad = 1;
  while (Kpad < pre_nms_top_n) Kpad <<= 1;
  hipStream_t s = as_stream(stream);
  SelDev sd;
  sd.hist = w.hist; sd.cnt = w.cnt; sd.ties = w.ties;
  sd.npass = (p->dtype == MXDET_DTYPE_BF16) ? 2 : 4;
  sd.keymask = (p->dtype == MXDET_DTYPE_BF16) ? 0xffff0000u : 0xffffffffu;
  sd.B = B; sd.pre_n = pre_nms_top_n;
  int max_nl = 0;
  for (int l = 0; l < L; ++l) max_nl = max_nl > p->H[l] * p->W[l] * p->A ? max_nl : p->H[l] * p->W[l] * p->A;
  long long gthreads = (long long)N * A_total;
  if (gthreads < (long long)kMaxPasses * B * 256) gthreads = (long long)kMaxPasses * B * 256;
  hipLaunchKernelGGL(proposal_gather_kernel, dim3((unsigned)ceil_div<long long>(gthreads, 256)), dim3(256), 0, s, d,
                     N, A_total, w.fkeys, sd);
  const dim3 sel_grid(ceil_div(max_nl, kSelChunk), L, N);
  for (int ps = 0; ps < sd.npass; ++ps)
    hipLaunchKernelGGL(proposal_hist_kernel, sel_grid, dim3(256), 0, s, d, A_total, (const unsigned*)w.fkeys, sd, ps);
  hipLaunchKernelGGL(proposal_select_kernel, sel_grid, dim3(256), 0, s, d, A_total, (const unsigned*)w.fkeys, sd,
                     w.keys);
  hipLaunchKernelGGL(proposal_finish_kernel, dim3(L, N), dim3(1024), 0, s, d, A_total, (const unsigned*)w.fkeys, sd,
                     Kpad, w.keys, w.counts);
  int tot = B * pre_nms_top_n;
  hipLaunchKernelGGL(proposal_decode_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, s, d, N,
                     pre_nms_top_n, im_info, min_size, w.keys, w.counts, w.boxes, w.invalid);
  int rc = check_launch("proposal(topk/decode)");
  if (rc) return rc;
  rc = mxdet_nms_batched((const float*)w.boxes, w.counts, w.invalid, B, pre_nms_top_n, nms_thresh,
                         per_level_post, w.keep_idx, w.num_keep, w.nms_ws, w.nms_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(proposal_merge_kernel, dim3(L, N), dim3(1024), merge_lds, s, L, pre_nms_top_n,
                     per_level_post, post_nms_top_n, w.keys, w.boxes, w.keep_idx, w.num_keep, rois, roi_scores,
                     roi_anchor, num_rois);
  return check_launch("proposal(merge)");
}
