// targets.hip -- anchor<->GT assignment (RPN / RetinaNet targets) and proposal-target (RoI sampling).
//
// Slots: core/anchor (/root/reference/README.md:16) and core/bbox (README.md:17); in the MXNet-1.3.0
// lineage these are numpy/Cython CustomOps (AnchorLoader / proposal_target, README.md:37,41-44) that
// run on the host and stall the GPU; here they stay on device. Sampling uses counter-based Philox
// keys (mxdet_math.h), so the sampled index sets are a pure function of (seed, step, image) and are
// bit-exact against oracle/mxdet_oracle.c.
#include "common.h"
#include "select.h"

namespace mxdet {

// ---------------------------------------------------------------------------------------------
// anchor target, stage 1: per-anchor max/argmax IoU and per-GT maximum (uint atomicMax on the
// non-negative float bits is order independent, hence deterministic).
__global__ void __launch_bounds__(256)
anchor_iou_kernel(const float4* __restrict__ anchors, long long A_total, const float* __restrict__ gt,
                  int G_max, const float* __restrict__ im_info, float allowed_border,
                  float* __restrict__ max_iou, int32_t* __restrict__ argmax,
                  unsigned* __restrict__ gt_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sg = (float*)smem_raw;                    // [G_max][5]
  unsigned* sgm = (unsigned*)(sg + G_max * 5);     // [G_max]
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < G_max * 5; i += blockDim.x) sg[i] = gt[(long long)n * G_max * 5 + i];
  for (int i = threadIdx.x; i < G_max; i += blockDim.x) sgm[i] = 0u;
  __syncthreads();
  long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a < A_total) {
    float4 b = anchors[a];
    float im_h = im_info[n * 3 + 0], im_w = im_info[n * 3 + 1];
    bool inside = (b.x >= -allowed_border) && (b.y >= -allowed_border) &&
                  (b.z < im_w + allowed_border) && (b.w < im_h + allowed_border);
    float best = -1.0f;
    int bi = -1;
    if (inside) {
      for (int g = 0; g < G_max; ++g) {
        if (sg[g * 5 + 4] < 0.0f) continue;
        float v = mxdet_iou(b.x, b.y, b.z, b.w, sg[g * 5], sg[g * 5 + 1], sg[g * 5 + 2], sg[g * 5 + 3]);
        if (v > best) { best = v; bi = g; }
        if (v > 0.0f) atomicMax(&sgm[g], __float_as_uint(v));
      }
    }
    max_iou[(long long)n * A_total + a] = best;
    argmax[(long long)n * A_total + a] = bi;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G_max; g += blockDim.x)
    if (sgm[g] != 0u) atomicMax(&gt_max[(long long)n * G_max + g], sgm[g]);
}

// Candidate compaction for the sampler: every foreground anchor and every background anchor whose Philox key is
// below kBgKeyCut (expected ~3 % of them) is appended to a per-image list, in parallel. The sampler then selects
// among a few thousand entries instead of scanning 268k anchors 8 times from one CU; if a list overflows or the
// filtered background list is shorter than the number wanted, it falls back to the full scan (same result).
constexpr int kCandCap = 16384;
constexpr unsigned kBgKeyCut = 0x08000000u;   // 2^32 / 32
struct CandLists {
  unsigned* key;   // [N][2][kCandCap]
  int* idx;        // [N][2][kCandCap]
  int* count;      // [N][4]: appended fg, appended bg, total fg, total bg
};
// Workgroup-aggregated append: ONE global atomic per workgroup and stream (every wave of the grid hitting the same
// counter costs ~5 ns each at the L2: 17k waves made this kernel 90 us). Waves count with a ballot, the counts
// meet in LDS, the first thread reserves the workgroup's run, lanes take consecutive slots. Must be called by
// every thread of a 256-thread workgroup (lab < 0 = nothing to append); contains two barriers. count[n*4+0] ends up
// as the number of ALL foreground anchors, count[n*4+1] as the number of background anchors below the key cut (both
// may exceed kCandCap: then the list is incomplete). The order of a list does not matter to the sampler.
__device__ __forceinline__ void cand_append(const CandLists& c, int n, int lab, unsigned key, int a) {
  __shared__ int s_cnt[2][4];
  __shared__ int s_base[2];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  bool want[2];
  unsigned long long m[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    want[st] = (st == 0) ? (lab == 1) : (lab == 0 && key < kBgKeyCut);
    m[st] = __ballot(want[st]);
    if (lane == 0) s_cnt[st][wv] = __popcll(m[st]);
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int st = threadIdx.x;
    const int tot = s_cnt[st][0] + s_cnt[st][1] + s_cnt[st][2] + s_cnt[st][3];
    s_base[st] = tot ? atomicAdd(&c.count[n * 4 + st], tot) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    if (!want[st]) continue;
    int base = s_base[st];
    for (int k = 0; k < wv; ++k) base += s_cnt[st][k];
    const unsigned long long below = (lane == 0) ? 0ull : (m[st] & (~0ull >> (64 - lane)));
    const int pos = base + __popcll(below);
    if (pos < kCandCap) {
      c.key[((long long)n * 2 + st) * kCandCap + pos] = key;
      c.idx[((long long)n * 2 + st) * kCandCap + pos] = a;
    }
  }
}

// stage 2: labels before sampling
__global__ void __launch_bounds__(256)
anchor_label_kernel(const float4* __restrict__ anchors, long long A_total, const float* __restrict__ gt,
                    int G_max, const float* __restrict__ max_iou, const unsigned* __restrict__ gt_max,
                    float fg_thresh, float bg_thresh, unsigned seed, unsigned step,
                    const unsigned* __restrict__ step_dev, unsigned image_offset,
                    int32_t* __restrict__ labels, unsigned* __restrict__ keys, CandLists cand) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sg = (float*)smem_raw;
  unsigned* sgm = (unsigned*)(sg + G_max * 5);
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < G_max * 5; i += blockDim.x) sg[i] = gt[(long long)n * G_max * 5 + i];
  for (int i = threadIdx.x; i < G_max; i += blockDim.x) sgm[i] = gt_max[(long long)n * G_max + i];
  __syncthreads();
  long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = a < A_total;
  float m = in_range ? max_iou[(long long)n * A_total + a] : -1.0f;
  int lab = -1;
  if (m >= 0.0f) {  // inside the image and at least one valid GT (m = -1 otherwise)
    if (m < bg_thresh) lab = 0;
    if (m >= fg_thresh) lab = 1;
    if (lab != 1 && m > 0.0f) {
      float4 b = anchors[a];
      for (int g = 0; g < G_max; ++g) {
        if (sg[g * 5 + 4] < 0.0f || sgm[g] == 0u) continue;
        float v = mxdet_iou(b.x, b.y, b.z, b.w, sg[g * 5], sg[g * 5 + 1], sg[g * 5 + 2], sg[g * 5 + 3]);
        if (__float_as_uint(v) == sgm[g]) { lab = 1; break; }
      }
    }
  }
  // sampling key of this anchor in its stream (0 = fg, 1 = bg), computed once, in parallel
  if (step_dev) step = *step_dev;
  unsigned key = lab >= 0 ? mxdet_sample_key(seed, step, image_offset + (unsigned)n, lab == 1 ? 0u : 1u, (unsigned)a) : 0u;
  if (in_range) {
    labels[(long long)n * A_total + a] = lab;
    keys[(long long)n * A_total + a] = key;
  }
  if (cand.count != nullptr) cand_append(cand, n, lab, key, (int)a);   // workgroup-uniform call
}

// An inside anchor of an image with no valid GT has max_iou = -1 above; the lineage labels those
// background. Handled here: stage 2b turns them into label 0 when the image has no GT at all.
__global__ void __launch_bounds__(256)
anchor_nogt_kernel(const float4* __restrict__ anchors, long long A_total, const float* __restrict__ gt,
                   int G_max, const float* __restrict__ im_info, float allowed_border, unsigned seed,
                   unsigned step, const unsigned* __restrict__ step_dev, unsigned image_offset,
                   int32_t* __restrict__ labels, float* __restrict__ max_iou,
                   unsigned* __restrict__ keys, CandLists cand) {
  const int n = blockIdx.y;
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  for (int g = threadIdx.x; g < G_max; g += blockDim.x)
    if (gt[((long long)n * G_max + g) * 5 + 4] >= 0.0f) any = 1;
  __syncthreads();
  if (any) return;   // block-uniform
  long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool inside = false;
  if (a < A_total) {
    float4 b = anchors[a];
    float im_h = im_info[n * 3 + 0], im_w = im_info[n * 3 + 1];
    inside = (b.x >= -allowed_border) && (b.y >= -allowed_border) && (b.z < im_w + allowed_border) &&
             (b.w < im_h + allowed_border);
  }
  if (step_dev) step = *step_dev;
  unsigned key = 0u;
  if (inside) {
    key = mxdet_sample_key(seed, step, image_offset + (unsigned)n, 1u, (unsigned)a);
    labels[(long long)n * A_total + a] = 0;
    max_iou[(long long)n * A_total + a] = 0.0f;
    keys[(long long)n * A_total + a] = key;
  }
  if (cand.count != nullptr) cand_append(cand, n, inside ? 0 : -1, key, (int)a);
}

// stage 3: per-image selection thresholds (one 1024-thread workgroup per image). The k smallest
// (key, index) of each stream are described by a SelectResult; stage 4 applies them in parallel.
struct AnchorSel { unsigned T[2], IT[2]; int mode[2]; };

__global__ void __launch_bounds__(1024)
anchor_sample_kernel(long long A_total, int batch_size, int max_fg, const int32_t* __restrict__ labels,
                     const unsigned* __restrict__ keys, CandLists cand, AnchorSel* __restrict__ sel) {
  __shared__ SelectSmem sm;
  const int n = blockIdx.x;
  const int32_t* lab = labels + (long long)n * A_total;
  const unsigned* key = keys + (long long)n * A_total;
  const int A = (int)A_total;
  const int n_fg_list = cand.count[n * 4 + 0], n_bg_list = cand.count[n * 4 + 1];
  const int n_fg_all = n_fg_list;   // every foreground anchor is appended (count keeps counting past the cap)
  const unsigned* fk = cand.key + ((long long)n * 2 + 0) * kCandCap;
  const unsigned* bk = cand.key + ((long long)n * 2 + 1) * kCandCap;
  const int* fi = cand.idx + ((long long)n * 2 + 0) * kCandCap;
  const int* bi = cand.idx + ((long long)n * 2 + 1) * kCandCap;
  SelectResult rf, rb;
  if (n_fg_list <= kCandCap) {   // complete list of foreground anchors
    auto kf = [&](int i, unsigned& kv) -> bool { kv = fk[i]; return true; };
    auto xf = [&](int i) -> unsigned { return (unsigned)fi[i]; };
    rf = block_select_threshold(n_fg_list, max_fg, 32, kf, sm, xf);
  } else {
    auto kf = [&](int i, unsigned& kv) -> bool { kv = key[i]; return lab[i] == 1; };
    rf = block_select_threshold(A, max_fg, 32, kf, sm);
  }
  const int nfg = n_fg_all < max_fg ? n_fg_all : max_fg;
  const int want_bg = batch_size - nfg;
  // the filtered list holds exactly the background anchors with key < kBgKeyCut: if it has at least want_bg
  // entries, the want_bg smallest keys overall are all inside it
  if (n_bg_list <= kCandCap && n_bg_list >= want_bg) {
    auto kb = [&](int i, unsigned& kv) -> bool { kv = bk[i]; return true; };
    auto xb = [&](int i) -> unsigned { return (unsigned)bi[i]; };
    rb = block_select_threshold(n_bg_list, want_bg, 32, kb, sm, xb);
    if (rb.mode == 1) { rb.mode = 0; rb.T = kBgKeyCut - 1u; rb.IT = 0xffffffffu; }   // "all of the list", not all bg
  } else {
    auto kb = [&](int i, unsigned& kv) -> bool { kv = key[i]; return lab[i] == 0; };
    rb = block_select_threshold(A, want_bg, 32, kb, sm);
  }
  if (threadIdx.x == 0) {
    AnchorSel o;
    o.T[0] = rf.T; o.IT[0] = rf.IT; o.mode[0] = rf.mode;
    o.T[1] = rb.T; o.IT[1] = rb.IT; o.mode[1] = rb.mode;
    sel[n] = o;
  }
}

// stage 4: apply the sampling predicate and write the regression targets
__global__ void __launch_bounds__(256)
anchor_encode_kernel(const float4* __restrict__ anchors, long long A_total, const float* __restrict__ gt,
                     int G_max, int32_t* __restrict__ labels, const int32_t* __restrict__ argmax,
                     const unsigned* __restrict__ keys, const AnchorSel* __restrict__ sel,
                     float4* __restrict__ targets) {
  const int n = blockIdx.y;
  long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A_total) return;
  long long idx = (long long)n * A_total + a;
  int lab = labels[idx];
  if (sel != nullptr && lab >= 0) {
    const AnchorSel s = sel[n];
    const int st = lab == 1 ? 0 : 1;
    SelectResult r;
    r.T = s.T[st]; r.IT = s.IT[st]; r.mode = s.mode[st]; r.n_cand = 0;
    if (!r.chosen(keys[idx], (unsigned)a)) {
      lab = -1;
      labels[idx] = -1;
    }
  }
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lab == 1) {
    int g = argmax[idx];
    float4 b = anchors[a];
    const float* q = gt + ((long long)n * G_max + g) * 5;
    float o[4];
    mxdet_encode(b.x, b.y, b.z, b.w, q[0], q[1], q[2], q[3], o);
    t = make_float4(o[0], o[1], o[2], o[3]);
  }
  targets[idx] = t;
}

// ---------------------------------------------------------------------------------------------
// proposal target: one workgroup per image.
__global__ void __launch_bounds__(1024)
proposal_target_kernel(const float* __restrict__ rois, const int32_t* __restrict__ num_rois,
                       int rois_stride, const float* __restrict__ gt, int G_max, int R, int max_fg,
                       float fg_thresh, float bg_hi, float bg_lo, int num_classes, int class_agnostic,
                       float4 means, float4 stds, unsigned seed, unsigned step,
                       const unsigned* __restrict__ step_dev, unsigned image_offset,
                       float* __restrict__ out_rois, int32_t* __restrict__ labels,
                       float* __restrict__ bbox_targets, float* __restrict__ bbox_weights,
                       int32_t* __restrict__ matched_gt, int32_t* __restrict__ num_fg_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // layout: gt[G_max*5] f32 | valid_gt[G_max] i32 | cstate[ncap] i8 (0 none,1 fg cand,2 bg cand,
  //         3 fg chosen, 4 bg chosen) | cgt[ncap] i16
  const int ncap = rois_stride + G_max;
  float* sg = (float*)smem_raw;
  int* vg = (int*)(sg + G_max * 5);
  short* cgt = (short*)(vg + G_max);
  signed char* cstate = (signed char*)(cgt + ((ncap + 7) & ~7));
  __shared__ SelectSmem sm;
  __shared__ int n_valid_gt, cnt_bg;
  const int n = blockIdx.x;
  const unsigned image = image_offset + (unsigned)n;
  if (step_dev) step = *step_dev;
  for (int i = threadIdx.x; i < G_max * 5; i += blockDim.x) sg[i] = gt[(long long)n * G_max * 5 + i];
  if (threadIdx.x == 0) cnt_bg = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int g = 0; g < G_max; ++g)
      if (sg[g * 5 + 4] >= 0.0f) vg[c++] = g;
    n_valid_gt = c;
  }
  __syncthreads();
  int nr = num_rois[n];
  nr = nr > rois_stride ? rois_stride : (nr < 0 ? 0 : nr);
  const int nv = n_valid_gt;
  const int nc = nr + nv;
  const float* rb = rois + (long long)n * rois_stride * 5;
  auto cand_box = [&](int i, float* b) {
    if (i < nr) {
      const float* r = rb + (long long)i * 5;
      b[0] = r[1]; b[1] = r[2]; b[2] = r[3]; b[3] = r[4];
    } else {
      const float* q = sg + vg[i - nr] * 5;
      b[0] = q[0]; b[1] = q[1]; b[2] = q[2]; b[3] = q[3];
    }
  };
  for (int i = threadIdx.x; i < nc; i += blockDim.x) {
    float b[4];
    cand_box(i, b);
    float best = -1.0f;
    int bi = -1;
    for (int k = 0; k < nv; ++k) {
      const float* q = sg + vg[k] * 5;
      float v = mxdet_iou(b[0], b[1], b[2], b[3], q[0], q[1], q[2], q[3]);
      if (v > best) { best = v; bi = vg[k]; }
    }
    if (nv == 0) best = 0.0f;
    signed char st = 0;
    if (best >= fg_thresh && bi >= 0) st = 1;
    else if (best < bg_hi && best >= bg_lo) st = 2;
    cstate[i] = st;
    cgt[i] = (short)bi;
  }
  __syncthreads();
  auto kfg = [&](int i, unsigned& kv) -> bool {
    kv = mxdet_sample_key(seed, step, image, 2u, (unsigned)i);
    return cstate[i] == 1;
  };
  auto kbg = [&](int i, unsigned& kv) -> bool {
    kv = mxdet_sample_key(seed, step, image, 3u, (unsigned)i);
    return cstate[i] == 2;
  };
  SelectResult sel_f = block_select_threshold(nc, max_fg, 32, kfg, sm);
  const int nfg = sel_f.n_cand < max_fg ? sel_f.n_cand : max_fg;
  SelectResult sel_b = block_select_threshold(nc, R - nfg, 32, kbg, sm);
  const int want_bg = (R - nfg) > 0 ? (R - nfg) : 0;
  if (threadIdx.x == 0) cnt_bg = sel_b.n_cand < want_bg ? sel_b.n_cand : want_bg;
  for (int i = threadIdx.x; i < nc; i += blockDim.x) {
    unsigned kv;
    if (cstate[i] == 1) { kfg(i, kv); if (sel_f.chosen(kv, (unsigned)i)) cstate[i] = 3; }
    else if (cstate[i] == 2) { kbg(i, kv); if (sel_b.chosen(kv, (unsigned)i)) cstate[i] = 4; }
  }
  __syncthreads();
  const int nbg = cnt_bg;
  const int reg_dim = class_agnostic ? 4 : 4 * num_classes;
  float* orois = out_rois + (long long)n * R * 5;
  int32_t* olab = labels + (long long)n * R;
  float* otgt = bbox_targets + (long long)n * R * reg_dim;
  float* owgt = bbox_weights + (long long)n * R * reg_dim;
  int32_t* omg = matched_gt + (long long)n * R;
  // targets / weights were zero-filled by a chip-wide kernel before this launch (1.3 MB per image is far too much
  // for one workgroup); here only the padding slots
  for (int s = nfg + nbg + threadIdx.x; s < R; s += blockDim.x) {
    float* r = orois + (long long)s * 5;
    r[0] = (float)n; r[1] = 0.f; r[2] = 0.f; r[3] = 0.f; r[4] = 0.f;
    olab[s] = -1;
    omg[s] = -1;
  }
  __syncthreads();
  // ordered compaction: chosen fg first, then chosen bg, each in ascending candidate order
  int base_fg = 0, base_bg = 0;
  for (int base = 0; base < nc; base += blockDim.x) {
    int i = base + threadIdx.x;
    signed char st = (i < nc) ? cstate[i] : 0;
    int tf, tb;
    int rf = block_excl_count(st == 3, sm.scratch, &tf);
    int rbk = block_excl_count(st == 4, sm.scratch, &tb);
    if (st == 3 || st == 4) {
      int slot = (st == 3) ? (base_fg + rf) : (nfg + base_bg + rbk);
      float b[4];
      cand_box(i, b);
      float* r = orois + (long long)slot * 5;
      r[0] = (float)n; r[1] = b[0]; r[2] = b[1]; r[3] = b[2]; r[4] = b[3];
      int g = cgt[i];
      omg[slot] = g;
      if (st == 3) {
        const float* q = sg + g * 5;
        int cls = (int)q[4];
        olab[slot] = cls;
        float o[4];
        mxdet_encode(b[0], b[1], b[2], b[3], q[0], q[1], q[2], q[3], o);
        int c0 = class_agnostic ? 0 : 4 * cls;
        float* t = otgt + (long long)slot * reg_dim + c0;
        float* w = owgt + (long long)slot * reg_dim + c0;
        t[0] = (o[0] - means.x) / stds.x;
        t[1] = (o[1] - means.y) / stds.y;
        t[2] = (o[2] - means.z) / stds.z;
        t[3] = (o[3] - means.w) / stds.w;
        w[0] = 1.0f; w[1] = 1.0f; w[2] = 1.0f; w[3] = 1.0f;
      } else {
        olab[slot] = 0;
      }
    }
    base_fg += tf;
    base_bg += tb;
  }
  if (threadIdx.x == 0) num_fg_out[n] = nfg;
}

struct AnchorWs {
  int32_t* argmax;
  unsigned* gt_max;
  float* max_iou;
  unsigned* keys;
  AnchorSel* sel;
  CandLists cand;
  size_t cand_count_bytes;
  size_t total;
};
static AnchorWs carve_anchor(void* base, int N, long long A_total, int G_max) {
  AnchorWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  char* p = (char*)base;
  w.argmax = (int32_t*)(p + take((size_t)N * A_total * 4));
  w.gt_max = (unsigned*)(p + take((size_t)N * G_max * 4));
  w.max_iou = (float*)(p + take((size_t)N * A_total * 4));
  w.keys = (unsigned*)(p + take((size_t)N * A_total * 4));
  w.sel = (AnchorSel*)(p + take((size_t)N * sizeof(AnchorSel)));
  w.cand.key = (unsigned*)(p + take((size_t)N * 2 * kCandCap * 4));
  w.cand.idx = (int*)(p + take((size_t)N * 2 * kCandCap * 4));
  w.cand_count_bytes = align_up((size_t)N * 4 * sizeof(int), 16);
  w.cand.count = (int*)(p + take(w.cand_count_bytes));
  w.total = off;
  return w;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" size_t mxdet_anchor_target_workspace_bytes(int32_t N, int64_t A_total, int32_t G_max) {
  if (N <= 0 || A_total <= 0 || G_max <= 0) return 0;
  return carve_anchor(nullptr, N, A_total, G_max).total;
}

extern "C" int mxdet_anchor_target(const float* anchors, int64_t A_total, const float* gt_boxes,
                                   int32_t N, int32_t G_max, const float* im_info, float fg_thresh,
                                   float bg_thresh, float allowed_border, int32_t batch_size,
                                   float fg_fraction, uint32_t seed, uint32_t step,
                                   const uint32_t* step_dev, uint32_t image_offset, int32_t* labels,
                                   int32_t* matched_gt,
                                   float* bbox_targets, float* max_iou, void* workspace,
                                   size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && A_total > 0 && G_max > 0, MXDET_ESHAPE, "anchor_target: bad sizes");
  MXDET_REQUIRE(A_total < (1ll << 30) && G_max <= 1024, MXDET_ESHAPE, "anchor_target: too large");
  MXDET_REQUIRE(anchors && gt_boxes && im_info && labels && bbox_targets, MXDET_EINVAL,
                "anchor_target: null pointer");
  AnchorWs w = carve_anchor(workspace, N, A_total, G_max);
  MXDET_REQUIRE(workspace && workspace_bytes >= w.total, MXDET_EWORKSPACE,
                "anchor_target: workspace %zu < %zu", workspace_bytes, w.total);
  hipStream_t s = as_stream(stream);
  int32_t* amax = matched_gt ? matched_gt : w.argmax;
  float* miou = max_iou ? max_iou : w.max_iou;
  hipError_t e = zero_async(w.gt_max, (size_t)N * G_max * 4, s);
  MXDET_REQUIRE(e == hipSuccess, MXDET_EHIP, "anchor_target: memset failed");
  e = zero_async(w.cand.count, w.cand_count_bytes, s);
  MXDET_REQUIRE(e == hipSuccess, MXDET_EHIP, "anchor_target: memset failed");
  CandLists cl = w.cand;
  if (batch_size <= 0) cl.count = nullptr;   // no sampling: no candidate lists
  dim3 grid((unsigned)ceil_div<long long>(A_total, 256), N);
  size_t lds = (size_t)G_max * 6 * 4;
  hipLaunchKernelGGL(anchor_iou_kernel, grid, dim3(256), lds, s, (const float4*)anchors,
                     (long long)A_total, gt_boxes, G_max, im_info, allowed_border, miou, amax,
                     w.gt_max);
  hipLaunchKernelGGL(anchor_label_kernel, grid, dim3(256), lds, s, (const float4*)anchors,
                     (long long)A_total, gt_boxes, G_max, miou, w.gt_max, fg_thresh, bg_thresh, seed,
                     step, step_dev, image_offset, labels, w.keys, cl);
  hipLaunchKernelGGL(anchor_nogt_kernel, grid, dim3(256), 0, s, (const float4*)anchors,
                     (long long)A_total, gt_boxes, G_max, im_info, allowed_border, seed, step, step_dev,
                     image_offset, labels, miou, w.keys, cl);
  if (batch_size > 0) {
    int max_fg = (int)(fg_fraction * (float)batch_size);
    hipLaunchKernelGGL(anchor_sample_kernel, dim3(N), dim3(1024), 0, s, (long long)A_total,
                       batch_size, max_fg, (const int32_t*)labels, (const unsigned*)w.keys, cl, w.sel);
  }
  hipLaunchKernelGGL(anchor_encode_kernel, grid, dim3(256), 0, s, (const float4*)anchors,
                     (long long)A_total, gt_boxes, G_max, labels, amax, (const unsigned*)w.keys,
                     batch_size > 0 ? (const AnchorSel*)w.sel : (const AnchorSel*)nullptr,
                     (float4*)bbox_targets);
  return check_launch("anchor_target");
}

extern "C" int mxdet_proposal_target(const float* rois, const int32_t* num_rois, int32_t rois_stride,
                                     const float* gt_boxes, int32_t N, int32_t G_max,
                                     int32_t rois_per_image, float fg_fraction, float fg_thresh,
                                     float bg_hi, float bg_lo, int32_t num_classes,
                                     int32_t class_agnostic, const float* means, const float* stds,
                                     uint32_t seed, uint32_t step, const uint32_t* step_dev,
                                     uint32_t image_offset, float* out_rois, int32_t* labels,
                                     float* bbox_targets,
                                     float* bbox_weights, int32_t* matched_gt, int32_t* num_fg,
                                     mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && rois_stride > 0 && G_max > 0 && rois_per_image > 0 && num_classes > 0,
                MXDET_ESHAPE, "proposal_target: bad sizes");
  MXDET_REQUIRE(rois && num_rois && gt_boxes && means && stds && out_rois && labels && bbox_targets &&
                    bbox_weights && matched_gt && num_fg,
                MXDET_EINVAL, "proposal_target: null pointer");
  MXDET_REQUIRE(G_max <= 1024 && rois_stride <= 16384, MXDET_ESHAPE, "proposal_target: too large");
  int ncap = rois_stride + G_max;
  size_t lds = (size_t)G_max * 5 * 4 + (size_t)G_max * 4 + (size_t)((ncap + 7) & ~7) * 2 +
               (size_t)((ncap + 15) & ~15);
  MXDET_REQUIRE(lds <= 120 * 1024, MXDET_ESHAPE, "proposal_target: LDS budget exceeded");
  int max_fg = (int)(fg_fraction * (float)rois_per_image);
  // means/stds are tiny host arrays by contract (passed by value into the kernel)
  float4 m = make_float4(means[0], means[1], means[2], means[3]);
  float4 sd = make_float4(stds[0], stds[1], stds[2], stds[3]);
  {
    const size_t reg_dim = class_agnostic ? 4 : 4 * (size_t)num_classes;
    const size_t bytes = (size_t)N * rois_per_image * reg_dim * sizeof(float);
    hipError_t e = zero_async(bbox_targets, bytes, as_stream(stream));
    if (e == hipSuccess) e = zero_async(bbox_weights, bytes, as_stream(stream));
    MXDET_REQUIRE(e == hipSuccess, MXDET_EHIP, "proposal_target: zero fill failed");
  }
  hipLaunchKernelGGL(proposal_target_kernel, dim3(N), dim3(1024), lds, as_stream(stream), rois,
                     num_rois, rois_stride, gt_boxes, G_max, rois_per_image, max_fg, fg_thresh, bg_hi,
                     bg_lo, num_classes, class_agnostic, m, sd, seed, step, step_dev, image_offset, out_rois,
                     labels, bbox_targets, bbox_weights, matched_gt, num_fg);
  return check_launch("proposal_target");
}
