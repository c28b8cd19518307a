// postprocess.hip -- test-time detection post-processing on gfx950.
//
// Slot: core/evaluation (/root/reference/README.md:20) + ops (README.md:24); MXNet-lineage role: the per-class
// `im_detect` -> threshold -> nms -> max_per_image loop that py-faster-rcnn / mx-rcnn run in numpy on the host.
// Here the whole thing stays on the device:
//   (1) det_score_box_kernel : softmax over the classes and class-specific box decoding (deltas * stds + means at the
//                              roi, clipped to the image), one thread per roi
//   (2) det_class_sort_kernel: one workgroup per (image, foreground class): rois with score > thresh -> LDS bitonic
//                              sort by (score desc, roi index asc) -> sorted boxes / keys / count
//   (3) mxdet_nms_batched    : the bitmask NMS of boxes.hip over the N*(C-1) lists
//   (4) det_merge_kernel     : per image, rank-merge of the per-class kept lists (binary searches in LDS) and cut
//                              to max_per_image by (score desc, roi index asc, class asc)
// Integer-exact (kept indices, classes) and bit-exact (scores, boxes) against oracle/mxdet_oracle.c.
#include "common.h"
#include "select.h"

namespace mxdet {

constexpr int kDetMaxRois = 4096;   // rois per image (LDS sort buffer = 32 KiB)

__global__ void det_score_box_kernel(const void* __restrict__ cls, const void* __restrict__ reg, int dtype, int ld_cls,
                                     int ld_reg, const float* __restrict__ rois, const int32_t* __restrict__ num_rois,
                                     const float* __restrict__ im_info, int N, int Rpi, int C, float4 means,
                                     float4 stds, float* __restrict__ scores, float4* __restrict__ boxes) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N * Rpi) return;
  const int n = r / Rpi, i = r - n * Rpi;
  float* so = scores + (size_t)r * C;
  float4* bo = boxes + (size_t)r * C;
  if (i >= num_rois[n]) {
    for (int c = 0; c < C; ++c) { so[c] = 0.0f; bo[c] = make_float4(0.f, 0.f, 0.f, 0.f); }
    return;
  }
  // softmax with a fixed evaluation order: m = max, e_c = exp(x_c - m), s = sum in class order, p_c = e_c / s
  float m = load_as_f32(cls, (int64_t)r * ld_cls, dtype);
  for (int c = 1; c < C; ++c) {
    float v = load_as_f32(cls, (int64_t)r * ld_cls + c, dtype);
    m = v > m ? v : m;
  }
  float s = 0.0f;
  for (int c = 0; c < C; ++c) s = s + mxdet_expf(load_as_f32(cls, (int64_t)r * ld_cls + c, dtype) - m);
  const float* rb = rois + (size_t)r * 5;
  const float x1 = rb[1], y1 = rb[2], x2 = rb[3], y2 = rb[4];
  const float im_h = im_info[n * 3 + 0], im_w = im_info[n * 3 + 1];
  for (int c = 0; c < C; ++c) {
    float e = mxdet_expf(load_as_f32(cls, (int64_t)r * ld_cls + c, dtype) - m);
    so[c] = e / s;
    float d0 = load_as_f32(reg, (int64_t)r * ld_reg + 4 * c + 0, dtype) * stds.x + means.x;
    float d1 = load_as_f32(reg, (int64_t)r * ld_reg + 4 * c + 1, dtype) * stds.y + means.y;
    float d2 = load_as_f32(reg, (int64_t)r * ld_reg + 4 * c + 2, dtype) * stds.z + means.z;
    float d3 = load_as_f32(reg, (int64_t)r * ld_reg + 4 * c + 3, dtype) * stds.w + means.w;
    float o[4];
    mxdet_decode_clip(x1, y1, x2, y2, d0, d1, d2, d3, im_h, im_w, o);
    bo[c] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// one workgroup per (foreground class, image): candidates above the threshold, sorted
__global__ void __launch_bounds__(1024)
det_class_sort_kernel(const float* __restrict__ scores, const float4* __restrict__ boxes,
                      const int32_t* __restrict__ sparse_cls,   // null: dense [roi][class] inputs; else one class per roi
                      const int32_t* __restrict__ num_rois, int Rpi, int C, int Kpad, float thresh,
                      float4* __restrict__ sboxes, unsigned long long* __restrict__ skeys,
                      int32_t* __restrict__ counts) {
  __shared__ unsigned long long list[kDetMaxRois];
  __shared__ int n_sel;
  const int c = blockIdx.x + 1, n = blockIdx.y;
  const int b = n * (C - 1) + (c - 1);
  if (threadIdx.x == 0) n_sel = 0;
  for (int i = threadIdx.x; i < Kpad; i += blockDim.x) list[i] = 0ull;
  __syncthreads();
  int nr = num_rois[n];
  nr = nr > Rpi ? Rpi : (nr < 0 ? 0 : nr);
  for (int i = threadIdx.x; i < nr; i += blockDim.x) {
    float s;
    if (sparse_cls) s = sparse_cls[(size_t)n * Rpi + i] == c ? scores[(size_t)n * Rpi + i] : 0.0f;
    else s = scores[((size_t)n * Rpi + i) * C + c];
    if (s > thresh) {
      int pos = atomicAdd(&n_sel, 1);
      list[pos] = ((unsigned long long)mxdet_float_key(s) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
    }
  }
  __syncthreads();
  block_bitonic_sort_desc(list, Kpad);
  const int cnt = n_sel;
  for (int j = threadIdx.x; j < Rpi; j += blockDim.x) {
    unsigned long long k = list[j];
    skeys[(size_t)b * Rpi + j] = k;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < cnt) {
      unsigned i = 0xffffffffu - (unsigned)(k & 0xffffffffull);
      bx = sparse_cls ? boxes[(size_t)n * Rpi + i] : boxes[((size_t)n * Rpi + i) * C + c];
    }
    sboxes[(size_t)b * Rpi + j] = bx;
  }
  if (threadIdx.x == 0) counts[b] = cnt;
}

// per image: merge the kept lists of all classes, keep the max_det best by (score desc, roi asc, class asc)
__global__ void __launch_bounds__(1024)
det_merge_kernel(int C, int Rpi, int cap, int max_det, const unsigned long long* __restrict__ skeys,
                 const float4* __restrict__ sboxes, const int32_t* __restrict__ keep_idx,
                 const int32_t* __restrict__ num_keep, float* __restrict__ dets, int32_t* __restrict__ num_dets) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* lk = (unsigned long long*)smem_raw;      // [C-1][cap] merged keys
  int* nk = (int*)(lk + (size_t)(C - 1) * cap);                // [C-1]
  const int n = blockIdx.x, L = C - 1;
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    int v = num_keep[n * L + l];
    nk[l] = v > cap ? cap : v;
  }
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    const int b = n * L + l;
    for (int j = threadIdx.x; j < nk[l]; j += blockDim.x) {
      unsigned long long k = skeys[(size_t)b * Rpi + keep_idx[(size_t)b * Rpi + j]];
      unsigned i = 0xffffffffu - (unsigned)(k & 0xffffffffull);
      // same score key, low word orders by (roi index, class): unique over the whole image
      lk[(size_t)l * cap + j] = (k & 0xffffffff00000000ull) | (unsigned long long)(0xffffffffu - (i * (unsigned)C + (unsigned)(l + 1)));
    }
  }
  __syncthreads();
  int total = 0;
  for (int l = 0; l < L; ++l) total += nk[l];
  for (int l = 0; l < L; ++l) {
    const int b = n * L + l;
    for (int j = threadIdx.x; j < nk[l]; j += blockDim.x) {
      const unsigned long long e = lk[(size_t)l * cap + j];
      int rank = j;
      for (int l2 = 0; l2 < L; ++l2) {
        if (l2 == l || nk[l2] == 0) continue;
        int lo = 0, hi = nk[l2];
        const unsigned long long* q = lk + (size_t)l2 * cap;
        while (lo < hi) {
          int mid = (lo + hi) >> 1;
          if (q[mid] > e) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
      if (rank < max_det) {
        const int pos = keep_idx[(size_t)b * Rpi + j];
        const float4 bx = sboxes[(size_t)b * Rpi + pos];
        const unsigned fk = (unsigned)(e >> 32);
        const unsigned u = (fk & 0x80000000u) ? (fk & 0x7fffffffu) : ~fk;
        float* d = dets + ((size_t)n * max_det + rank) * 6;
        d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w; d[4] = __uint_as_float(u); d[5] = (float)(l + 1);
      }
    }
  }
  const int nout = total < max_det ? total : max_det;
  for (int j = nout + threadIdx.x; j < max_det; j += blockDim.x) {
    float* d = dets + ((size_t)n * max_det + j) * 6;
    d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f; d[4] = 0.f; d[5] = -1.0f;
  }
  if (threadIdx.x == 0) num_dets[n] = nout;
}

struct DetWs {
  float* scores; float4* boxes; float4* sboxes; unsigned long long* skeys;
  int32_t* counts; int32_t* keep_idx; int32_t* num_keep; void* nms_ws;
  size_t nms_bytes, total;
};

static DetWs det_carve(void* base, int N, int Rpi, int C) {
  DetWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  char* p = (char*)base;
  const size_t R = (size_t)N * Rpi, B = (size_t)N * (C - 1);
  w.scores = (float*)(p + take(R * C * 4));
  w.boxes = (float4*)(p + take(R * C * 16));
  w.sboxes = (float4*)(p + take(B * Rpi * 16));
  w.skeys = (unsigned long long*)(p + take(B * Rpi * 8));
  w.counts = (int32_t*)(p + take(B * 4));
  w.keep_idx = (int32_t*)(p + take(B * Rpi * 4));
  w.num_keep = (int32_t*)(p + take(B * 4));
  w.nms_bytes = mxdet_nms_batched_workspace_bytes((int32_t)B, Rpi);
  w.nms_ws = (void*)(p + take(w.nms_bytes));
  w.total = off;
  return w;
}

// RetinaNet: merged per-level top-k (mxdet_proposal outputs) -> sparse candidates: probability, class, box
__global__ void retina_candidates_kernel(const float* __restrict__ rois, const float* __restrict__ logit,
                                         const int32_t* __restrict__ gidx, const int32_t* __restrict__ num, int N, int R,
                                         int C, float* __restrict__ score, int32_t* __restrict__ cls,
                                         float4* __restrict__ boxes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * R) return;
  const int n = i / R, j = i - n * R;
  if (j >= num[n]) {
    score[i] = 0.0f; cls[i] = 0; boxes[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float z = logit[i];
  score[i] = z >= 0.0f ? 1.0f / (1.0f + mxdet_expf(-z)) : mxdet_expf(z) / (1.0f + mxdet_expf(z));
  cls[i] = gidx[i] % C + 1;               // level offsets are multiples of A*C
  const float* r = rois + (size_t)i * 5;
  boxes[i] = make_float4(r[1], r[2], r[3], r[4]);
}

struct RetinaWs {
  float* rois; float* logit; int32_t* gidx; int32_t* num; float* score; int32_t* cls; float4* boxes;
  float4* sboxes; unsigned long long* skeys; int32_t* counts; int32_t* keep_idx; int32_t* num_keep;
  void* nms_ws; size_t nms_bytes; void* prop_ws; size_t prop_bytes; size_t total;
};

static RetinaWs retina_carve(void* base, const mxdet_pyramid_t* p, int N, int pre_n, int R, int C) {
  RetinaWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  char* q = (char*)base;
  const size_t NR = (size_t)N * R, B = (size_t)N * C;
  w.rois = (float*)(q + take(NR * 20));
  w.logit = (float*)(q + take(NR * 4));
  w.gidx = (int32_t*)(q + take(NR * 4));
  w.num = (int32_t*)(q + take((size_t)N * 4));
  w.score = (float*)(q + take(NR * 4));
  w.cls = (int32_t*)(q + take(NR * 4));
  w.boxes = (float4*)(q + take(NR * 16));
  w.sboxes = (float4*)(q + take(B * R * 16));
  w.skeys = (unsigned long long*)(q + take(B * R * 8));
  w.counts = (int32_t*)(q + take(B * 4));
  w.keep_idx = (int32_t*)(q + take(B * R * 4));
  w.num_keep = (int32_t*)(q + take(B * 4));
  w.nms_bytes = mxdet_nms_batched_workspace_bytes((int32_t)B, R);
  w.nms_ws = (void*)(q + take(w.nms_bytes));
  w.prop_bytes = mxdet_proposal_workspace_bytes(p, N, pre_n);
  w.prop_ws = (void*)(q + take(w.prop_bytes));
  w.total = off;
  return w;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" size_t mxdet_detection_postprocess_workspace_bytes(int32_t N, int32_t rois_per_image,
                                                              int32_t num_classes) {
  if (N <= 0 || rois_per_image <= 0 || num_classes <= 1) return 0;
  return det_carve(nullptr, N, rois_per_image, num_classes).total;
}

extern "C" int mxdet_detection_postprocess(const void* cls_logits, const void* bbox_pred, int32_t dtype,
                                           int32_t ld_cls, int32_t ld_reg, const float* rois,
                                           const int32_t* num_rois, const float* im_info, int32_t N,
                                           int32_t rois_per_image, int32_t num_classes, const float* means,
                                           const float* stds, float score_thresh, float nms_thresh,
                                           int32_t max_per_image, float* dets, int32_t* num_dets,
                                           void* workspace, size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && rois_per_image > 0 && num_classes > 1, MXDET_ESHAPE, "detection_postprocess: bad sizes");
  MXDET_REQUIRE(rois_per_image <= kDetMaxRois, MXDET_ESHAPE, "detection_postprocess: rois_per_image %d > %d",
                rois_per_image, kDetMaxRois);
  MXDET_REQUIRE(max_per_image > 0 && max_per_image <= rois_per_image, MXDET_ESHAPE,
                "detection_postprocess: bad max_per_image %d", max_per_image);
  MXDET_REQUIRE(dtype == MXDET_DTYPE_F32 || dtype == MXDET_DTYPE_BF16, MXDET_EINVAL, "detection_postprocess: bad dtype");
  MXDET_REQUIRE(ld_cls >= num_classes && ld_reg >= 4 * num_classes, MXDET_ESHAPE,
                "detection_postprocess: leading dimensions too small");
  MXDET_REQUIRE(cls_logits && bbox_pred && rois && num_rois && im_info && means && stds && dets && num_dets,
                MXDET_EINVAL, "detection_postprocess: null pointer");
  const int C = num_classes, Rpi = rois_per_image, B = N * (C - 1);
  DetWs w = det_carve(workspace, N, Rpi, C);
  MXDET_REQUIRE(workspace && workspace_bytes >= w.total, MXDET_EWORKSPACE,
                "detection_postprocess: workspace %zu < %zu", workspace_bytes, w.total);
  const size_t merge_lds = (size_t)(C - 1) * max_per_image * 8 + (size_t)(C - 1) * 4;
  MXDET_REQUIRE(merge_lds <= 150 * 1024, MXDET_ESHAPE, "detection_postprocess: classes*max_per_image too large");
  hipStream_t s = as_stream(stream);
  const float4 m4 = make_float4(means[0], means[1], means[2], means[3]);
  const float4 s4 = make_float4(stds[0], stds[1], stds[2], stds[3]);
  hipLaunchKernelGGL(det_score_box_kernel, dim3(ceil_div(N * Rpi, 64)), dim3(64), 0, s, cls_logits, bbox_pred,
                     dtype, ld_cls, ld_reg, rois, num_rois, im_info, N, Rpi, C, m4, s4, w.scores, w.boxes);
  int Kpad = 1;
  while (Kpad < Rpi) Kpad <<= 1;
  hipLaunchKernelGGL(det_class_sort_kernel, dim3(C - 1, N), dim3(1024), 0, s, (const float*)w.scores,
                     (const float4*)w.boxes, (const int32_t*)nullptr, num_rois, Rpi, C, Kpad, score_thresh, w.sboxes, w.skeys,
                     w.counts);
  int rc = check_launch("detection_postprocess(score/sort)");
  if (rc) return rc;
  rc = mxdet_nms_batched((const float*)w.sboxes, w.counts, nullptr, B, Rpi, nms_thresh, max_per_image, w.keep_idx,
                         w.num_keep, w.nms_ws, w.nms_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(det_merge_kernel, dim3(N), dim3(1024), merge_lds, s, C, Rpi, max_per_image, max_per_image,
                     (const unsigned long long*)w.skeys, (const float4*)w.sboxes, (const int32_t*)w.keep_idx,
                     (const int32_t*)w.num_keep, dets, num_dets);
  return check_launch("detection_postprocess(merge)");
}

static int retina_candidates_cap(const mxdet_pyramid_t* p, int pre_n) {
  long long r = (long long)p->num_levels * pre_n;
  return (int)(r < kDetMaxRois ? r : kDetMaxRois);
}

extern "C" size_t mxdet_retina_detect_workspace_bytes(const mxdet_pyramid_t* p, int32_t N, int32_t pre_nms_top_n) {
  if (!p || N <= 0 || pre_nms_top_n <= 0 || p->num_levels <= 0 || p->classes <= 0) return 0;
  return retina_carve(nullptr, p, N, pre_nms_top_n, retina_candidates_cap(p, pre_nms_top_n), p->classes).total;
}

extern "C" int mxdet_retina_detect(const mxdet_pyramid_t* p, int32_t N, const float* im_info, int32_t pre_nms_top_n,
                                   float score_thresh, float nms_thresh, int32_t max_per_image, float* dets,
                                   int32_t* num_dets, void* workspace, size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(p && im_info && dets && num_dets, MXDET_EINVAL, "retina_detect: null pointer");
  MXDET_REQUIRE(N > 0 && p->num_levels > 0 && p->classes >= 1 && pre_nms_top_n > 0, MXDET_ESHAPE, "retina_detect: bad sizes");
  const int C = p->classes, R = retina_candidates_cap(p, pre_nms_top_n), B = N * C;
  MXDET_REQUIRE(max_per_image > 0 && max_per_image <= R, MXDET_ESHAPE, "retina_detect: bad max_per_image %d", max_per_image);
  const size_t merge_lds = (size_t)C * max_per_image * 8 + (size_t)C * 4;
  MXDET_REQUIRE(merge_lds <= 150 * 1024, MXDET_ESHAPE, "retina_detect: classes*max_per_image too large");
  RetinaWs w = retina_carve(workspace, p, N, pre_nms_top_n, R, C);
  MXDET_REQUIRE(workspace && workspace_bytes >= w.total, MXDET_EWORKSPACE, "retina_detect: workspace %zu < %zu",
                workspace_bytes, w.total);
  // per-level top-k + decode + merge: the proposal path with suppression switched off (no IoU exceeds 2)
  int rc = mxdet_proposal(p, N, im_info, pre_nms_top_n, R, 2.0f, 0.0f, w.rois, w.logit, w.gidx, w.num, w.prop_ws,
                          w.prop_bytes, stream);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(retina_candidates_kernel, dim3(ceil_div(N * R, 256)), dim3(256), 0, s, (const float*)w.rois,
                     (const float*)w.logit, (const int32_t*)w.gidx, (const int32_t*)w.num, N, R, C, w.score, w.cls, w.boxes);
  int Kpad = 1;
  while (Kpad < R) Kpad <<= 1;
  hipLaunchKernelGGL(det_class_sort_kernel, dim3(C, N), dim3(1024), 0, s, (const float*)w.score, (const float4*)w.boxes,
                     (const int32_t*)w.cls, (const int32_t*)w.num, R, C + 1, Kpad, score_thresh, w.sboxes, w.skeys, w.counts);
  rc = check_launch("retina_detect(candidates/sort)");
  if (rc) return rc;
  rc = mxdet_nms_batched((const float*)w.sboxes, w.counts, nullptr, B, R, nms_thresh, max_per_image, w.keep_idx,
                         w.num_keep, w.nms_ws, w.nms_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(det_merge_kernel, dim3(N), dim3(1024), merge_lds, s, C + 1, R, max_per_image, max_per_image,
                     (const unsigned long long*)w.skeys, (const float4*)w.sboxes, (const int32_t*)w.keep_idx,
                     (const int32_t*)w.num_keep, dets, num_dets);
  return check_launch("retina_detect(merge)");
}
