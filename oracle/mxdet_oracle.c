/*
 * mxdet_oracle.c -- CPU restatement of the two-stage-detector hot path. TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference snapshot (/root/reference) holds only README.md + LICENSE -- no
 * code, tests, golden vectors or fixtures -- and its arithmetic lives in MXNet 1.3.0
 * (/root/reference/README.md:37), which is not installed and cannot be fetched. Each function below
 * therefore restates the *published* algorithm of the op that the declared slot
 * (README.md:15-19,24,27-32) would have bound, with the conventions frozen in DESIGN.md section 3
 * ("convention chosen", py-faster-rcnn / mx-rcnn / Detectron lineage). It is pinned only by the
 * hand-computed known-answer vectors in tests/golden/ and cross-checks against numpy/torch-CPU.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (mxdetection_amd/) never does.
 *
 * Written as plain sequential loops, deliberately unlike the HIP kernels (sort instead of radix
 * select, sequential greedy NMS instead of bitmasks, ...). The only code shared with the product
 * is include/mxdet_math.h's expf/logf/Philox/bf16 helpers, which is what makes bit-exact
 * comparison possible at all; those helpers are pinned separately in tests/test_math.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mxdet_math.h"

#define API __attribute__((visibility("default")))

/* ---- scalar box math, restated (legacy +1 convention) ---------------------------------------- */
static float o_iou(const float* a, const float* b) {
  float ix1 = a[0] > b[0] ? a[0] : b[0];
  float iy1 = a[1] > b[1] ? a[1] : b[1];
  float ix2 = a[2] < b[2] ? a[2] : b[2];
  float iy2 = a[3] < b[3] ? a[3] : b[3];
  float iw = ix2 - ix1 + 1.0f, ih = iy2 - iy1 + 1.0f;
  if (iw <= 0.0f || ih <= 0.0f) return 0.0f;
  float inter = iw * ih;
  float area_a = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
  float area_b = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
  float uni = area_a + area_b;
  uni = uni - inter;
  return inter / uni;
}

static float o_clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void o_decode_clip(const float* box, const float* d, float im_h, float im_w, float* out) {
  float w = box[2] - box[0] + 1.0f, h = box[3] - box[1] + 1.0f;
  float cx = box[0] + 0.5f * (w - 1.0f), cy = box[1] + 0.5f * (h - 1.0f);
  float dw = d[2], dh = d[3];
  const float clipv = 4.135166556742356f;
  if (dw > clipv) dw = clipv;
  if (dh > clipv) dh = clipv;
  float pcx = d[0] * w;
  pcx = pcx + cx;
  float pcy = d[1] * h;
  pcy = pcy + cy;
  float pw = mxdet_expf(dw) * w, ph = mxdet_expf(dh) * h;
  float hw = 0.5f * (pw - 1.0f), hh = 0.5f * (ph - 1.0f);
  out[0] = o_clipf(pcx - hw, 0.0f, im_w - 1.0f);
  out[1] = o_clipf(pcy - hh, 0.0f, im_h - 1.0f);
  out[2] = o_clipf(pcx + hw, 0.0f, im_w - 1.0f);
  out[3] = o_clipf(pcy + hh, 0.0f, im_h - 1.0f);
}

static void o_encode(const float* ex, const float* gt, float* out) {
  float ew = ex[2] - ex[0] + 1.0f, eh = ex[3] - ex[1] + 1.0f;
  float ecx = ex[0] + 0.5f * (ew - 1.0f), ecy = ex[1] + 0.5f * (eh - 1.0f);
  float gw = gt[2] - gt[0] + 1.0f, gh = gt[3] - gt[1] + 1.0f;
  float gcx = gt[0] + 0.5f * (gw - 1.0f), gcy = gt[1] + 0.5f * (gh - 1.0f);
  out[0] = (gcx - ecx) / ew;
  out[1] = (gcy - ecy) / eh;
  out[2] = mxdet_logf(gw / ew);
  out[3] = mxdet_logf(gh / eh);
}

static uint32_t o_float_key(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

API void oracle_box_iou(const float* a, int64_t na, const float* b, int64_t nb, float* out) {
  for (int64_t i = 0; i < na; ++i)
    for (int64_t j = 0; j < nb; ++j) out[i * nb + j] = o_iou(a + 4 * i, b + 4 * j);
}

/* py-faster-rcnn generate_anchors: base_size = stride, ratios then scales, numpy round-half-even */
API void oracle_base_anchors(int stride, const double* ratios, int nr, const double* scales, int ns,
                             float* out /* [nr*ns,4] */) {
  double w = stride, h = stride, xc = 0.5 * (w - 1.0), yc = 0.5 * (h - 1.0);
  double size = w * h;
  int k = 0;
  for (int r = 0; r < nr; ++r) {
    double ws = rint(sqrt(size / ratios[r]));
    double hs = rint(ws * ratios[r]);
    for (int s = 0; s < ns; ++s) {
      double wss = ws * scales[s], hss = hs * scales[s];
      out[k * 4 + 0] = (float)(xc - 0.5 * (wss - 1.0));
      out[k * 4 + 1] = (float)(yc - 0.5 * (hss - 1.0));
      out[k * 4 + 2] = (float)(xc + 0.5 * (wss - 1.0));
      out[k * 4 + 3] = (float)(yc + 0.5 * (hss - 1.0));
      ++k;
    }
  }
}

API void oracle_grid_anchors(const float* base, int A, int H, int W, int stride, float* out) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int a = 0; a < A; ++a) {
        float* o = out + (((int64_t)y * W + x) * A + a) * 4;
        float sx = (float)(x * stride), sy = (float)(y * stride);
        o[0] = base[a * 4 + 0] + sx;
        o[1] = base[a * 4 + 1] + sy;
        o[2] = base[a * 4 + 2] + sx;
        o[3] = base[a * 4 + 3] + sy;
      }
}

API int oracle_fpn_level(const float* roi5, int lvl_min, int lvl_max) {
  /* k = floor(4 + log2(sqrt(w*h)/224)) evaluated in double, clamped */
  double w = (double)(float)(roi5[3] - roi5[1] + 1.0f), h = (double)(float)(roi5[4] - roi5[2] + 1.0f);
  float area = (float)w * (float)h; /* same fp32 product the kernel forms */
  double s = sqrt((double)area);
  int k;
  if (!(s > 0.0)) k = lvl_min;
  else k = (int)floor(4.0 + log2(s / 224.0));
  if (k < lvl_min) k = lvl_min;
  if (k > lvl_max) k = lvl_max;
  return k;
}

/* ---- greedy NMS on one score-sorted list ------------------------------------------------------ */
API int oracle_nms(const float* boxes, int n, const uint8_t* invalid, float thresh, int max_keep,
                   int32_t* keep) {
  uint8_t* dead = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    if (dead[i] || (invalid && invalid[i])) continue;
    if (nk < max_keep) keep[nk] = i;
    ++nk;
    for (int j = i + 1; j < n; ++j)
      if (!dead[j] && o_iou(boxes + 4 * i, boxes + 4 * j) > thresh) dead[j] = 1;
  }
  free(dead);
  return nk < max_keep ? nk : max_keep;
}

/* ---- pyramid proposal --------------------------------------------------------------------------- */
typedef struct { uint32_t key; uint32_t gidx; } o_cand;
static int o_cmp_cand(const void* a, const void* b) {
  const o_cand* x = (const o_cand*)a; const o_cand* y = (const o_cand*)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;      /* score descending (bit-pattern order) */
  if (x->gidx != y->gidx) return x->gidx < y->gidx ? -1 : 1;  /* then anchor index ascending */
  return 0;
}

/* scores[l]: [N][n_l] fp32, deltas[l]: [N][n_l][4] fp32, canonical (y,x,a) order inside a level */
API void oracle_proposal(int L, int A, const int32_t* H, const int32_t* W, const int32_t* stride,
                         const float* const* scores, const float* const* deltas,
                         const float* const* base, int N, const float* im_info, int pre_n, int post_n,
                         float thresh, float min_size, float* rois, float* roi_scores,
                         int32_t* roi_anchor, int32_t* num_rois) {
  int64_t off[9];
  off[0] = 0;
  for (int l = 0; l < L; ++l) off[l + 1] = off[l] + (int64_t)H[l] * W[l] * A;
  int lvl_cap = post_n < pre_n ? post_n : pre_n;
  for (int n = 0; n < N; ++n) {
    o_cand* merged = (o_cand*)malloc(sizeof(o_cand) * (size_t)L * lvl_cap + 16);
    float* mboxes = (float*)malloc(sizeof(float) * 4 * (size_t)L * lvl_cap + 16);
    int nm = 0;
    for (int l = 0; l < L; ++l) {
      int nl = H[l] * W[l] * A;
      o_cand* c = (o_cand*)malloc(sizeof(o_cand) * (size_t)nl + 16);
      for (int i = 0; i < nl; ++i) {
        c[i].key = o_float_key(scores[l][(int64_t)n * nl + i]);
        c[i].gidx = (uint32_t)(off[l] + i);
      }
      qsort(c, (size_t)nl, sizeof(o_cand), o_cmp_cand);
      int k = nl < pre_n ? nl : pre_n;
      float* bx = (float*)malloc(sizeof(float) * 4 * (size_t)k + 16);
      uint8_t* inv = (uint8_t*)malloc((size_t)k + 16);
      for (int j = 0; j < k; ++j) {
        int local = (int)(c[j].gidx - off[l]);
        int a = local % A, cell = local / A, x = cell % W[l], y = cell / W[l];
        float anc[4];
        float sx = (float)(x * stride[l]), sy = (float)(y * stride[l]);
        anc[0] = base[l][a * 4 + 0] + sx; anc[1] = base[l][a * 4 + 1] + sy;
        anc[2] = base[l][a * 4 + 2] + sx; anc[3] = base[l][a * 4 + 3] + sy;
        o_decode_clip(anc, deltas[l] + ((int64_t)n * nl + local) * 4, im_info[n * 3], im_info[n * 3 + 1],
                      bx + 4 * j);
        float ms = min_size * im_info[n * 3 + 2];
        float w = bx[4 * j + 2] - bx[4 * j] + 1.0f, h = bx[4 * j + 3] - bx[4 * j + 1] + 1.0f;
        inv[j] = (w < ms || h < ms) ? 1 : 0;
      }
      int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(k > 0 ? k : 1));
      int nk = oracle_nms(bx, k, inv, thresh, lvl_cap, keep);
      for (int j = 0; j < nk; ++j) {
        merged[nm] = c[keep[j]];
        memcpy(mboxes + 4 * nm, bx + 4 * keep[j], 16);
        ++nm;
      }
      free(keep); free(inv); free(bx); free(c);
    }
    /* merge: stable selection by (score desc, gidx asc) over the union */
    int* order = (int*)malloc(sizeof(int) * (size_t)(nm > 0 ? nm : 1));
    for (int i = 0; i < nm; ++i) order[i] = i;
    /* insertion-free: simple O(n log n) via qsort on an index array needs context; do a merge by
       repeatedly building (cand,index) pairs instead */
    typedef struct { o_cand c; int idx; } pair_t;
    pair_t* pr = (pair_t*)malloc(sizeof(pair_t) * (size_t)(nm > 0 ? nm : 1));
    for (int i = 0; i < nm; ++i) { pr[i].c = merged[i]; pr[i].idx = i; }
    qsort(pr, (size_t)nm, sizeof(pair_t), o_cmp_cand); /* o_cand is the first member */
    int nout = nm < post_n ? nm : post_n;
    for (int j = 0; j < post_n; ++j) {
      float* r = rois + ((int64_t)n * post_n + j) * 5;
      r[0] = (float)n;
      if (j < nout) {
        memcpy(r + 1, mboxes + 4 * pr[j].idx, 16);
        uint32_t fk = pr[j].c.key, u = (fk & 0x80000000u) ? (fk & 0x7fffffffu) : ~fk;
        float s; memcpy(&s, &u, 4);
        roi_scores[(int64_t)n * post_n + j] = s;
        roi_anchor[(int64_t)n * post_n + j] = (int32_t)pr[j].c.gidx;
      } else {
        r[1] = r[2] = r[3] = r[4] = 0.0f;
        roi_scores[(int64_t)n * post_n + j] = 0.0f;
        roi_anchor[(int64_t)n * post_n + j] = -1;
      }
    }
    num_rois[n] = nout;
    free(pr); free(order); free(mboxes); free(merged);
  }
}

/* ---- sampling: the k candidates with the smallest (philox key, index) ---------------------------- */
typedef struct { uint32_t key; int32_t idx; } o_samp;
static int o_cmp_samp(const void* a, const void* b) {
  const o_samp* x = (const o_samp*)a; const o_samp* y = (const o_samp*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
/* marks chosen[idx] = 1 for the selected candidates; returns how many were chosen */
static int o_sample(const int32_t* cand, int ncand, int k, uint32_t seed, uint32_t step,
                    uint32_t image, uint32_t stream, uint8_t* chosen) {
  if (k < 0) k = 0;
  o_samp* s = (o_samp*)malloc(sizeof(o_samp) * (size_t)(ncand > 0 ? ncand : 1));
  for (int i = 0; i < ncand; ++i) {
    s[i].idx = cand[i];
    s[i].key = mxdet_sample_key(seed, step, image, stream, (uint32_t)cand[i]);
  }
  qsort(s, (size_t)ncand, sizeof(o_samp), o_cmp_samp);
  int take = ncand < k ? ncand : k;
  for (int i = 0; i < take; ++i) chosen[s[i].idx] = 1;
  free(s);
  return take;
}

/* ---- anchor target ------------------------------------------------------------------------------ */
API void oracle_anchor_target(const float* anchors, int64_t A_total, const float* gt, int N, int G,
                              const float* im_info, float fg_thresh, float bg_thresh, float border,
                              int batch_size, float fg_fraction, uint32_t seed, uint32_t step,
                              uint32_t image_offset, int32_t* labels, int32_t* matched,
                              float* targets, float* max_iou_out) {
  for (int n = 0; n < N; ++n) {
    const float* g = gt + (int64_t)n * G * 5;
    float im_h = im_info[n * 3], im_w = im_info[n * 3 + 1];
    int32_t* lab = labels + (int64_t)n * A_total;
    int32_t* mg = matched + (int64_t)n * A_total;
    float* miou = max_iou_out + (int64_t)n * A_total;
    float* gmax = (float*)calloc((size_t)G, sizeof(float));
    uint8_t* inside = (uint8_t*)calloc((size_t)A_total, 1);
    int any_gt = 0;
    for (int k = 0; k < G; ++k) if (g[k * 5 + 4] >= 0.0f) any_gt = 1;
    for (int64_t a = 0; a < A_total; ++a) {
      const float* b = anchors + 4 * a;
      inside[a] = (b[0] >= -border && b[1] >= -border && b[2] < im_w + border && b[3] < im_h + border);
      float best = -1.0f; int bi = -1;
      if (inside[a])
        for (int k = 0; k < G; ++k) {
          if (g[k * 5 + 4] < 0.0f) continue;
          float v = o_iou(b, g + k * 5);
          if (v > best) { best = v; bi = k; }
          if (v > gmax[k]) gmax[k] = v;
        }
      miou[a] = best; mg[a] = bi;
    }
    for (int64_t a = 0; a < A_total; ++a) {
      int l = -1;
      float m = miou[a];
      if (inside[a] && !any_gt) { l = 0; miou[a] = 0.0f; }
      else if (m >= 0.0f) {
        if (m < bg_thresh) l = 0;
        if (m >= fg_thresh) l = 1;
        if (l != 1 && m > 0.0f)
          for (int k = 0; k < G; ++k) {
            if (g[k * 5 + 4] < 0.0f || !(gmax[k] > 0.0f)) continue;
            if (o_iou(anchors + 4 * a, g + k * 5) == gmax[k]) { l = 1; break; }
          }
      }
      lab[a] = l;
    }
    if (batch_size > 0) {
      int max_fg = (int)(fg_fraction * (float)batch_size);
      int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)A_total);
      uint8_t* chosen = (uint8_t*)calloc((size_t)A_total, 1);
      int nc = 0;
      for (int64_t a = 0; a < A_total; ++a) if (lab[a] == 1) cand[nc++] = (int32_t)a;
      int nfg = o_sample(cand, nc, max_fg, seed, step, image_offset + (uint32_t)n, 0u, chosen);
      for (int i = 0; i < nc; ++i) if (!chosen[cand[i]]) lab[cand[i]] = -1;
      memset(chosen, 0, (size_t)A_total);
      nc = 0;
      for (int64_t a = 0; a < A_total; ++a) if (lab[a] == 0) cand[nc++] = (int32_t)a;
      o_sample(cand, nc, batch_size - nfg, seed, step, image_offset + (uint32_t)n, 1u, chosen);
      for (int i = 0; i < nc; ++i) if (!chosen[cand[i]]) lab[cand[i]] = -1;
      free(cand); free(chosen);
    }
    for (int64_t a = 0; a < A_total; ++a) {
      float* t = targets + ((int64_t)n * A_total + a) * 4;
      t[0] = t[1] = t[2] = t[3] = 0.0f;
      if (lab[a] == 1) o_encode(anchors + 4 * a, g + mg[a] * 5, t);
    }
    free(gmax); free(inside);
  }
}

/* ---- proposal target ---------------------------------------------------------------------------- */
API void oracle_proposal_target(const float* rois, const int32_t* num_rois, int rois_stride,
                                const float* gt, int N, int G, int R, float fg_fraction,
                                float fg_thresh, float bg_hi, float bg_lo, int num_classes,
                                int class_agnostic, const float* means, const float* stds,
                                uint32_t seed, uint32_t step, uint32_t image_offset, float* out_rois,
                                int32_t* labels, float* tgt, float* wgt, int32_t* matched,
                                int32_t* num_fg) {
  int reg_dim = class_agnostic ? 4 : 4 * num_classes;
  int max_fg = (int)(fg_fraction * (float)R);
  for (int n = 0; n < N; ++n) {
    const float* g = gt + (int64_t)n * G * 5;
    int nr = num_rois[n];
    if (nr > rois_stride) nr = rois_stride;
    if (nr < 0) nr = 0;
    int nv = 0;
    int* vg = (int*)malloc(sizeof(int) * (size_t)(G > 0 ? G : 1));
    for (int k = 0; k < G; ++k) if (g[k * 5 + 4] >= 0.0f) vg[nv++] = k;
    int nc = nr + nv;
    float* cb = (float*)malloc(sizeof(float) * 4 * (size_t)(nc > 0 ? nc : 1));
    for (int i = 0; i < nr; ++i) memcpy(cb + 4 * i, rois + ((int64_t)n * rois_stride + i) * 5 + 1, 16);
    for (int i = 0; i < nv; ++i) memcpy(cb + 4 * (nr + i), g + vg[i] * 5, 16);
    int32_t* fgc = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1));
    int32_t* bgc = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1));
    int32_t* am = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1));
    int nfc = 0, nbc = 0;
    for (int i = 0; i < nc; ++i) {
      float best = -1.0f; int bi = -1;
      for (int k = 0; k < nv; ++k) {
        float v = o_iou(cb + 4 * i, g + vg[k] * 5);
        if (v > best) { best = v; bi = vg[k]; }
      }
      if (nv == 0) best = 0.0f;
      am[i] = bi;
      if (best >= fg_thresh && bi >= 0) fgc[nfc++] = i;
      else if (best < bg_hi && best >= bg_lo) bgc[nbc++] = i;
    }
    uint8_t* chf = (uint8_t*)calloc((size_t)(nc > 0 ? nc : 1), 1);
    uint8_t* chb = (uint8_t*)calloc((size_t)(nc > 0 ? nc : 1), 1);
    int nfg = o_sample(fgc, nfc, max_fg, seed, step, image_offset + (uint32_t)n, 2u, chf);
    int nbg = o_sample(bgc, nbc, R - nfg, seed, step, image_offset + (uint32_t)n, 3u, chb);
    float* orois = out_rois + (int64_t)n * R * 5;
    int32_t* olab = labels + (int64_t)n * R;
    float* ot = tgt + (int64_t)n * R * reg_dim;
    float* ow = wgt + (int64_t)n * R * reg_dim;
    int32_t* om = matched + (int64_t)n * R;
    memset(ot, 0, sizeof(float) * (size_t)R * reg_dim);
    memset(ow, 0, sizeof(float) * (size_t)R * reg_dim);
    int slot = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int i = 0; i < nc; ++i) {
        if (!(pass == 0 ? chf[i] : chb[i])) continue;
        float* r = orois + (int64_t)slot * 5;
        r[0] = (float)n; memcpy(r + 1, cb + 4 * i, 16);
        om[slot] = am[i];
        if (pass == 0) {
          const float* q = g + am[i] * 5;
          int cls = (int)q[4];
          olab[slot] = cls;
          float e[4];
          o_encode(cb + 4 * i, q, e);
          int c0 = class_agnostic ? 0 : 4 * cls;
          for (int k = 0; k < 4; ++k) {
            ot[(int64_t)slot * reg_dim + c0 + k] = (e[k] - means[k]) / stds[k];
            ow[(int64_t)slot * reg_dim + c0 + k] = 1.0f;
          }
        } else {
          olab[slot] = 0;
        }
        ++slot;
      }
    for (; slot < R; ++slot) {
      float* r = orois + (int64_t)slot * 5;
      r[0] = (float)n; r[1] = r[2] = r[3] = r[4] = 0.0f;
      olab[slot] = -1; om[slot] = -1;
    }
    num_fg[n] = nfg;
    (void)nbg;
    free(chf); free(chb); free(am); free(bgc); free(fgc); free(cb); free(vg);
  }
}

/* ---- RoIAlign (Detectron aligned=False / MXNet-1.3.0 contrib.ROIAlign semantics) ----------------- */
static void o_taps(float y, float x, int H, int W, int* yl, int* yh, int* xl, int* xh, float* w,
                   int* valid) {
  *valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.0f) y = 0.0f;
  if (x <= 0.0f) x = 0.0f;
  *yl = (int)y; *xl = (int)x;
  if (*yl >= H - 1) { *yh = *yl = H - 1; y = (float)*yl; } else *yh = *yl + 1;
  if (*xl >= W - 1) { *xh = *xl = W - 1; x = (float)*xl; } else *xh = *xl + 1;
  float ly = y - (float)*yl, lx = x - (float)*xl, hy = 1.0f - ly, hx = 1.0f - lx;
  w[0] = hy * hx; w[1] = hy * lx; w[2] = ly * hx; w[3] = ly * lx;
}

/* feats[l]: bf16 bits [N,H,W,C]; out: bf16 bits [R,PH,PW,C]; mode 0 fwd. For bwd (mode 1) `gout` is
 * the bf16 upstream gradient and dfeat[l] (fp32 [N,H,W,C], pre-zeroed by caller) is accumulated. */
API void oracle_roi_align(int L, int lvl_min, const int32_t* H, const int32_t* W, const float* scale,
                          const uint16_t* const* feats, float* const* dfeat, int N, int C,
                          const float* rois, const int32_t* levels, int64_t R, int PH, int PW,
                          int sampling_ratio, uint16_t* out, const uint16_t* gout, int mode) {
  for (int64_t r = 0; r < R; ++r) {
    const float* q = rois + r * 5;
    int l = levels[r] - lvl_min;
    if (l < 0) l = 0;
    if (l >= L) l = L - 1;
    int b = (int)q[0];
    if (b < 0) b = 0;
    if (b >= N) b = N - 1;
    float s = scale[l];
    float sw = q[1] * s, sh = q[2] * s, ew = q[3] * s, eh = q[4] * s;
    float rw = ew - sw, rh = eh - sh;
    if (!(rw > 1.0f)) rw = 1.0f;
    if (!(rh > 1.0f)) rh = 1.0f;
    float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    float count = (float)(gh * gw);
    int Hl = H[l], Wl = W[l];
    for (int ph = 0; ph < PH; ++ph)
      for (int pw = 0; pw < PW; ++pw)
        for (int c = 0; c < C; ++c) {
          float acc = 0.0f;
          float go = 0.0f;
          if (mode == 1) go = mxdet_bf16_to_f32(gout[((r * PH + ph) * PW + pw) * C + c]) / count;
          for (int iy = 0; iy < gh; ++iy) {
            float y = sh + (float)ph * bin_h;
            y = y + (((float)iy + 0.5f) * bin_h) / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = sw + (float)pw * bin_w;
              x = x + (((float)ix + 0.5f) * bin_w) / (float)gw;
              int yl, yh, xl, xh, valid; float w[4];
              o_taps(y, x, Hl, Wl, &yl, &yh, &xl, &xh, w, &valid);
              if (!valid) continue;
              int64_t base = (int64_t)b * Hl * Wl;
              int64_t i1 = ((base + (int64_t)yl * Wl + xl) * C) + c, i2 = ((base + (int64_t)yl * Wl + xh) * C) + c;
              int64_t i3 = ((base + (int64_t)yh * Wl + xl) * C) + c, i4 = ((base + (int64_t)yh * Wl + xh) * C) + c;
              if (mode == 0) {
                const uint16_t* f = feats[l];
                float v = w[0] * mxdet_bf16_to_f32(f[i1]);
                v = v + w[1] * mxdet_bf16_to_f32(f[i2]);
                v = v + w[2] * mxdet_bf16_to_f32(f[i3]);
                v = v + w[3] * mxdet_bf16_to_f32(f[i4]);
                acc = acc + v;
              } else {
                float* d = dfeat[l];
                d[i1] += w[0] * go; d[i2] += w[1] * go; d[i3] += w[2] * go; d[i4] += w[3] * go;
              }
            }
          }
          if (mode == 0) out[((r * PH + ph) * PW + pw) * C + c] = mxdet_f32_to_bf16(acc / count);
        }
  }
}

/* ---- losses (per-element math in fp32 exactly as specified, sums in double) ---------------------- */
static float o_softplus_neg_abs(float z) { float az = z < 0 ? -z : z; return mxdet_logf(1.0f + mxdet_expf(-az)); }
static float o_sigmoid(float z) {
  if (z >= 0.0f) return 1.0f / (1.0f + mxdet_expf(-z));
  float e = mxdet_expf(z); return e / (1.0f + e);
}
static float o_sl1(float x, float s2) {
  float ax = x < 0 ? -x : x, inv = 1.0f / s2;
  if (ax < inv) { float t = 0.5f * s2; t = t * x; return t * x; }
  return ax - 0.5f * inv;
}
static float o_sl1g(float x, float s2) {
  float ax = x < 0 ? -x : x, inv = 1.0f / s2;
  if (ax < inv) return s2 * x;
  return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f);
}

API void oracle_smooth_l1(const float* p, const float* t, const float* w, int64_t n, float sigma,
                          float* out, float* grad) {
  float s2 = sigma * sigma;
  for (int64_t i = 0; i < n; ++i) {
    float ww = w ? w[i] : 1.0f;
    float d = (p[i] - t[i]) * ww;
    if (out) out[i] = o_sl1(d, s2);
    if (grad) grad[i] = o_sl1g(d, s2) * ww;
  }
}

/* head: fp32 copy of the bf16 head output [N,H,W,Cpad]; outputs loss[2] (double sums * norm) and
 * grad [N,H,W,Cpad] fp32 (before bf16 rounding) */
API void oracle_rpn_loss_level(const float* head, int N, int H, int W, int A, int Cpad,
                               const int32_t* labels, const float* targets, int64_t A_total,
                               int64_t level_offset, float sigma, float norm, float loss_scale,
                               float* grad, double* loss) {
  float s2 = sigma * sigma;
  double lc = 0.0, lr = 0.0;
  for (int n = 0; n < N; ++n)
    for (int64_t cell = 0; cell < (int64_t)H * W; ++cell) {
      const float* h = head + ((int64_t)n * H * W + cell) * Cpad;
      float* g = grad + ((int64_t)n * H * W + cell) * Cpad;
      for (int c = 0; c < Cpad; ++c) g[c] = 0.0f;
      for (int a = 0; a < A; ++a) {
        int64_t gi = (int64_t)n * A_total + level_offset + cell * A + a;
        int lab = labels[gi];
        if (lab >= 0) {
          float z = h[a];
          float l = (z > 0 ? z : 0.0f) - z * (float)lab + o_softplus_neg_abs(z);
          lc += (double)l;
          g[a] = (o_sigmoid(z) - (float)lab) * norm * loss_scale;
        }
        if (lab == 1)
          for (int k = 0; k < 4; ++k) {
            float d = h[A + 4 * a + k] - targets[gi * 4 + k];
            lr += (double)o_sl1(d, s2);
            g[A + 4 * a + k] = o_sl1g(d, s2) * norm * loss_scale;
          }
      }
    }
  loss[0] = lc * (double)norm;
  loss[1] = lr * (double)norm;
}

API void oracle_rcnn_loss(const float* cls, const float* reg, int ld_cls, int ld_reg,
                          const int32_t* labels, const float* tgt, const float* wgt, int64_t R,
                          int num_classes, int reg_dim, float sigma, float norm, float loss_scale,
                          float* gcls, float* greg, double* loss) {
  float s2 = sigma * sigma;
  double lc = 0.0, lr = 0.0;
  for (int64_t r = 0; r < R; ++r) {
    int lab = labels[r];
    const float* z = cls + r * ld_cls;
    float mx = -3.0e38f;
    for (int c = 0; c < num_classes; ++c) if (z[c] > mx) mx = z[c];
    double se = 0.0;
    for (int c = 0; c < num_classes; ++c) se += (double)mxdet_expf(z[c] - mx);
    for (int c = 0; c < num_classes; ++c) {
      float g = 0.0f;
      if (lab >= 0) {
        float p = (float)((double)mxdet_expf(z[c] - mx) / se);
        g = (p - (c == lab ? 1.0f : 0.0f)) * norm * loss_scale;
      }
      gcls[r * ld_cls + c] = g;
    }
    if (lab >= 0) lc += -((double)(z[lab] - mx) - log(se));
    for (int c = 0; c < reg_dim; ++c) {
      float w = wgt[r * reg_dim + c], g = 0.0f;
      if (w != 0.0f && lab > 0) {
        float d = (reg[r * ld_reg + c] - tgt[r * reg_dim + c]) * w;
        lr += (double)o_sl1(d, s2);
        g = o_sl1g(d, s2) * w * norm * loss_scale;
      }
      greg[r * ld_reg + c] = g;
    }
  }
  loss[0] = lc * (double)norm;
  loss[1] = lr * (double)norm;
}

API void oracle_focal_loss(const float* logits, const int32_t* labels, int64_t n, int C, float alpha,
                           float gamma, float grad_scale, float* grad, double* loss) {
  int64_t nfg = 0;
  for (int64_t i = 0; i < n; ++i) if (labels[i] > 0) ++nfg;
  double inv = 1.0 / (double)(nfg > 1 ? nfg : 1);
  double acc = 0.0;
  for (int64_t r = 0; r < n; ++r)
    for (int c = 0; c < C; ++c) {
      double z = (double)logits[r * C + c];
      double g = 0.0;
      int lab = labels[r];
      if (lab >= 0) {
        double p = 1.0 / (1.0 + exp(-z));
        double logp = -log1p(exp(-z)), log1mp = -log1p(exp(z));
        if (z < -30) logp = z;  /* avoid overflow of exp(-z) */
        if (z > 30) log1mp = -z;
        if (lab == c + 1) {
          acc += -alpha * pow(1.0 - p, gamma) * logp;
          g = -alpha * pow(1.0 - p, gamma) * ((1.0 - p) - gamma * p * logp);
        } else {
          acc += -(1.0 - alpha) * pow(p, gamma) * log1mp;
          g = (1.0 - alpha) * pow(p, gamma) * (p - gamma * (1.0 - p) * log1mp);
        }
      }
      grad[r * C + c] = (float)(g * inv * (double)grad_scale);
    }
  loss[0] = acc * inv;
}

/* ---- dense ops: naive channels-last convolution (fp32 inputs already bf16-rounded by the caller,
 *      double accumulation) -- the reference against which the MFMA kernels are tolerance-checked */
API void oracle_conv2d_fwd(const float* x, const float* w, const float* bias, const float* residual,
                           int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                           int Ho, int Wo, int relu, int res_upsample, float* y) {
  for (int n = 0; n < N; ++n)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo)
        for (int co = 0; co < Cout; ++co) {
          double acc = 0.0;
          for (int kh = 0; kh < KH; ++kh) {
            int hi = ho * stride + kh - pad;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < KW; ++kw) {
              int wi = wo * stride + kw - pad;
              if (wi < 0 || wi >= W) continue;
              const float* xp = x + (((int64_t)n * H + hi) * W + wi) * Cin;
              const float* wp = w + (((int64_t)co * KH + kh) * KW + kw) * Cin;
              for (int ci = 0; ci < Cin; ++ci) acc += (double)xp[ci] * (double)wp[ci];
            }
          }
          float v = (float)acc;
          if (bias) v += bias[co];
          if (residual) {
            if (res_upsample) {
              int Hc = (Ho + 1) / 2, Wc = (Wo + 1) / 2;
              v += residual[(((int64_t)n * Hc + ho / 2) * Wc + wo / 2) * Cout + co];
            } else {
              v += residual[(((int64_t)n * Ho + ho) * Wo + wo) * Cout + co];
            }
          }
          if (relu && v < 0.0f) v = 0.0f;
          y[(((int64_t)n * Ho + ho) * Wo + wo) * Cout + co] = v;
        }
}

API void oracle_conv2d_dgrad(const float* dy, const float* w, int N, int H, int W, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, int Ho, int Wo, float* dx) {
  int64_t nx = (int64_t)N * H * W * Cin;
  double* acc = (double*)calloc((size_t)nx, sizeof(double));
  for (int n = 0; n < N; ++n)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo)
        for (int co = 0; co < Cout; ++co) {
          double g = (double)dy[(((int64_t)n * Ho + ho) * Wo + wo) * Cout + co];
          if (g == 0.0) continue;
          for (int kh = 0; kh < KH; ++kh) {
            int hi = ho * stride + kh - pad;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < KW; ++kw) {
              int wi = wo * stride + kw - pad;
              if (wi < 0 || wi >= W) continue;
              double* ap = acc + (((int64_t)n * H + hi) * W + wi) * Cin;
              const float* wp = w + (((int64_t)co * KH + kh) * KW + kw) * Cin;
              for (int ci = 0; ci < Cin; ++ci) ap[ci] += g * (double)wp[ci];
            }
          }
        }
  for (int64_t i = 0; i < nx; ++i) dx[i] = (float)acc[i];
  free(acc);
}

API void oracle_conv2d_wgrad(const float* x, const float* dy, int N, int H, int W, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, int Ho, int Wo, float* dw, float* db) {
  int64_t nw = (int64_t)Cout * KH * KW * Cin;
  double* acc = (double*)calloc((size_t)nw, sizeof(double));
  double* accb = (double*)calloc((size_t)Cout, sizeof(double));
  for (int n = 0; n < N; ++n)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo)
        for (int co = 0; co < Cout; ++co) {
          double g = (double)dy[(((int64_t)n * Ho + ho) * Wo + wo) * Cout + co];
          accb[co] += g;
          if (g == 0.0) continue;
          for (int kh = 0; kh < KH; ++kh) {
            int hi = ho * stride + kh - pad;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < KW; ++kw) {
              int wi = wo * stride + kw - pad;
              if (wi < 0 || wi >= W) continue;
              const float* xp = x + (((int64_t)n * H + hi) * W + wi) * Cin;
              double* ap = acc + (((int64_t)co * KH + kh) * KW + kw) * Cin;
              for (int ci = 0; ci < Cin; ++ci) ap[ci] += g * (double)xp[ci];
            }
          }
        }
  for (int64_t i = 0; i < nw; ++i) dw[i] = (float)acc[i];
  if (db) for (int c = 0; c < Cout; ++c) db[c] = (float)accb[c];
  free(acc); free(accb);
}

API void oracle_maxpool3x3s2(const float* x, int N, int H, int W, int C, float* y) {
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  for (int n = 0; n < N; ++n)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo)
        for (int c = 0; c < C; ++c) {
          float m = -3.0e38f;
          for (int kh = 0; kh < 3; ++kh) {
            int hi = ho * 2 + kh - 1;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
              int wi = wo * 2 + kw - 1;
              if (wi < 0 || wi >= W) continue;
              float v = x[(((int64_t)n * H + hi) * W + wi) * C + c];
              if (v > m) m = v;
            }
          }
          y[(((int64_t)n * Ho + ho) * Wo + wo) * C + c] = m;
        }
}

/* ---- Mask R-CNN: mask targets and mask loss ------------------------------------------------------------- */
API void oracle_mask_target(const float* rois, const int32_t* matched, const int32_t* labels,
                            const uint8_t* masks, int64_t R, int G, int H, int W, int S, uint8_t* targets,
                            int32_t* cls_out) {
  for (int64_t r = 0; r < R; ++r) {
    const float* q = rois + r * 5;
    int n = (int)q[0], g = matched[r], lab = labels[r];
    int fg = lab > 0 && g >= 0 && g < G;
    cls_out[r] = fg ? lab : -1;
    float rw = q[3] - q[1], rh = q[4] - q[2];
    if (!(rw > 1.0f)) rw = 1.0f;
    if (!(rh > 1.0f)) rh = 1.0f;
    float bw = rw / (float)S, bh = rh / (float)S;
    const uint8_t* m = masks + ((int64_t)n * G + (fg ? g : 0)) * H * W;
    for (int py = 0; py < S; ++py)
      for (int px = 0; px < S; ++px) {
        uint8_t t = 0;
        if (fg) {
          float y = q[2] + ((float)py + 0.5f) * bh, x = q[1] + ((float)px + 0.5f) * bw;
          int yl, yh, xl, xh, valid; float w[4];
          o_taps(y, x, H, W, &yl, &yh, &xl, &xh, w, &valid);
          if (valid) {
            float v = w[0] * (float)m[(int64_t)yl * W + xl];
            v = v + w[1] * (float)m[(int64_t)yl * W + xh];
            v = v + w[2] * (float)m[(int64_t)yh * W + xl];
            v = v + w[3] * (float)m[(int64_t)yh * W + xh];
            t = v >= 0.5f ? 1 : 0;
          }
        }
        targets[(r * S + py) * S + px] = t;
      }
  }
}

/* logits fp32 [R,S,S,Cpad]; returns loss (double) and grad fp32 (before bf16 rounding) */
API void oracle_mask_loss(const float* logits, const int32_t* cls, const uint8_t* targets, int64_t R, int S,
                          int Cpad, float loss_scale, float* grad, double* loss) {
  int64_t nfg = 0;
  for (int64_t r = 0; r < R; ++r) if (cls[r] > 0) ++nfg;
  float norm = 1.0f / (float)((nfg > 0 ? nfg : 1) * S * S);
  double acc = 0.0;
  for (int64_t i = 0; i < R * S * S * Cpad; ++i) grad[i] = 0.0f;
  for (int64_t r = 0; r < R; ++r) {
    int c = cls[r];
    if (!(c > 0 && c <= Cpad)) continue;
    for (int i = 0; i < S * S; ++i) {
      int64_t pix = r * S * S + i;
      float z = logits[pix * Cpad + (c - 1)], t = (float)targets[pix];
      float l = (z > 0 ? z : 0.0f) - z * t + o_softplus_neg_abs(z);
      acc += (double)l;
      grad[pix * Cpad + (c - 1)] = (o_sigmoid(z) - t) * norm * loss_scale;
    }
  }
  loss[0] = acc * (double)norm;
}

/* expose the shared scalar helpers so tests can pin them against numpy float64 */
API float oracle_expf(float x) { return mxdet_expf(x); }
API float oracle_logf(float x) { return mxdet_logf(x); }
API uint32_t oracle_sample_key(uint32_t seed, uint32_t step, uint32_t image, uint32_t stream, uint32_t idx) {
  return mxdet_sample_key(seed, step, image, stream, idx);
}
API void oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                       uint32_t* out) {
  mxdet_u32x4 r = mxdet_philox4x32_10(c0, c1, c2, c3, k0, k1);
  for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
API uint16_t oracle_f32_to_bf16(float f) { return mxdet_f32_to_bf16(f); }
API void oracle_decode_clip(const float* box, const float* d, float im_h, float im_w, float* out) {
  o_decode_clip(box, d, im_h, im_w, out);
}
API void oracle_encode(const float* ex, const float* gt, float* out) { o_encode(ex, gt, out); }

/* ---- test-time detection post-processing (follows the contract in include/mxdet.h:
 * "core/evaluation ... mxdet_detection_postprocess"; MXNet-lineage role: im_detect + per-class nms + max_per_image) --- */
typedef struct { uint32_t key; uint32_t roi; uint32_t cls; float box[4]; float score; } o_det;
static int o_cmp_det_cls(const void* a, const void* b) {      /* inside one class: score desc, roi asc */
  const o_det* x = (const o_det*)a; const o_det* y = (const o_det*)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  if (x->roi != y->roi) return x->roi < y->roi ? -1 : 1;
  return 0;
}
static int o_cmp_det_all(const void* a, const void* b) {      /* whole image: score desc, roi asc, class asc */
  const o_det* x = (const o_det*)a; const o_det* y = (const o_det*)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  if (x->roi != y->roi) return x->roi < y->roi ? -1 : 1;
  if (x->cls != y->cls) return x->cls < y->cls ? -1 : 1;
  return 0;
}
/* cls [N*R][C] f32, reg [N*R][4C] f32, rois [N*R][5]; dets [N][max_det][6]; scores_out/boxes_out optional */
API void oracle_detection_postprocess(const float* cls, const float* reg, const float* rois, const int32_t* num_rois,
                                      const float* im_info, int N, int R, int C, const float* means,
                                      const float* stds, float score_thresh, float nms_thresh, int max_det,
                                      float* dets, int32_t* num_dets, float* scores_out, float* boxes_out) {
  o_det* cand = (o_det*)malloc(sizeof(o_det) * (size_t)(R > 0 ? R : 1));
  o_det* kept = (o_det*)malloc(sizeof(o_det) * (size_t)(R > 0 ? R : 1) * (size_t)C);
  float* bx = (float*)malloc(sizeof(float) * 4 * (size_t)(R > 0 ? R : 1));
  int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
  float* sc = (float*)malloc(sizeof(float) * (size_t)R * C);
  float* bb = (float*)malloc(sizeof(float) * (size_t)R * C * 4);
  for (int n = 0; n < N; ++n) {
    int nr = num_rois[n]; nr = nr > R ? R : (nr < 0 ? 0 : nr);
    for (int i = 0; i < R; ++i) {
      const size_t r = (size_t)n * R + i;
      for (int c = 0; c < C; ++c) { sc[(size_t)i * C + c] = 0.0f; for (int k = 0; k < 4; ++k) bb[((size_t)i * C + c) * 4 + k] = 0.0f; }
      if (i >= nr) continue;
      float m = cls[r * C];
      for (int c = 1; c < C; ++c) m = cls[r * C + c] > m ? cls[r * C + c] : m;
      float s = 0.0f;
      for (int c = 0; c < C; ++c) s = s + mxdet_expf(cls[r * C + c] - m);
      for (int c = 0; c < C; ++c) {
        float e = mxdet_expf(cls[r * C + c] - m);
        sc[(size_t)i * C + c] = e / s;
        float d[4];
        for (int k = 0; k < 4; ++k) d[k] = reg[r * 4 * C + 4 * c + k] * stds[k] + means[k];
        o_decode_clip(rois + r * 5 + 1, d, im_info[n * 3], im_info[n * 3 + 1], bb + ((size_t)i * C + c) * 4);
      }
    }
    if (scores_out) memcpy(scores_out + (size_t)n * R * C, sc, sizeof(float) * (size_t)R * C);
    if (boxes_out) memcpy(boxes_out + (size_t)n * R * C * 4, bb, sizeof(float) * (size_t)R * C * 4);
    int nkept = 0;
    for (int c = 1; c < C; ++c) {
      int nc = 0;
      for (int i = 0; i < nr; ++i) {
        float s = sc[(size_t)i * C + c];
        if (s > score_thresh) {
          cand[nc].key = o_float_key(s); cand[nc].roi = (uint32_t)i; cand[nc].cls = (uint32_t)c; cand[nc].score = s;
          memcpy(cand[nc].box, bb + ((size_t)i * C + c) * 4, 16);
          ++nc;
        }
      }
      qsort(cand, (size_t)nc, sizeof(o_det), o_cmp_det_cls);
      for (int j = 0; j < nc; ++j) memcpy(bx + 4 * j, cand[j].box, 16);
      int nk = oracle_nms(bx, nc, NULL, nms_thresh, max_det, keep);
      for (int j = 0; j < nk; ++j) kept[nkept++] = cand[keep[j]];
    }
    qsort(kept, (size_t)nkept, sizeof(o_det), o_cmp_det_all);
    int nout = nkept < max_det ? nkept : max_det;
    for (int j = 0; j < max_det; ++j) {
      float* d = dets + ((size_t)n * max_det + j) * 6;
      if (j < nout) { memcpy(d, kept[j].box, 16); d[4] = kept[j].score; d[5] = (float)kept[j].cls; }
      else { d[0] = d[1] = d[2] = d[3] = d[4] = 0.0f; d[5] = -1.0f; }
    }
    num_dets[n] = nout;
  }
  free(cand); free(kept); free(bx); free(keep); free(sc); free(bb);
}

/* ------------------------------------------------------------------------------------------------
 * process_data (README.md:23): flip -> 8-bit bilinear resize -> channel swap, (v - mean) / std -> bf16 NCHW, zero pad.
 * The resize restates OpenCV's 8-bit INTER_LINEAR path (imgproc/resize.cpp: resizeGeneric_ coefficient set-up with
 * INTER_RESIZE_COEF_BITS = 11, HResizeLinear, VResizeLinear's FixedPtCast<int, uchar, 22>); cv2 is the resizer the
 * reference names (README.md:53-56). Parity with a cv2 build is UNPINNED: cv2 is not in this image. */
static void o_resize_coef(int d, double inv, int slen, int* s0, int* s1, int* c0, int* c1) {
  float f = (float)(((double)d + 0.5) * inv - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.0f; s = 0; }
  if (s >= slen - 1) { f = 0.0f; s = slen - 1; }
  *s0 = s;
  *s1 = s + 1 < slen ? s + 1 : slen - 1;
  *c0 = (int)lrintf((1.0f - f) * 2048.0f);
  *c1 = (int)lrintf(f * 2048.0f);
}
/* one image: src [sh][sw][3] u8 -> out planes [3][Hp][Wp] bf16 bits (out points at image n's first plane).
 * resized_u8, if not NULL, receives the [dh][dw][3] 8-bit resize result (before normalisation, after flip + swap). */
API void oracle_image_preprocess(const uint8_t* src, int sh, int sw, int dh, int dw, int flip, double inv_scale,
                                 int Hp, int Wp, const float* mean3, const float* std3, int swap_rb, uint16_t* out,
                                 uint8_t* resized_u8) {
  for (size_t i = 0; i < (size_t)3 * Hp * Wp; ++i) out[i] = 0;
  for (int y = 0; y < dh; ++y) {
    int sy0, sy1, b0, b1;
    o_resize_coef(y, inv_scale, sh, &sy0, &sy1, &b0, &b1);
    for (int x = 0; x < dw; ++x) {
      int sx0, sx1, a0, a1;
      o_resize_coef(x, inv_scale, sw, &sx0, &sx1, &a0, &a1);
      if (flip) { sx0 = sw - 1 - sx0; sx1 = sw - 1 - sx1; }
      for (int c = 0; c < 3; ++c) {
        const int sc = swap_rb ? 2 - c : c;
        const int t0 = (int)src[((size_t)sy0 * sw + sx0) * 3 + sc] * a0 + (int)src[((size_t)sy0 * sw + sx1) * 3 + sc] * a1;
        const int t1 = (int)src[((size_t)sy1 * sw + sx0) * 3 + sc] * a0 + (int)src[((size_t)sy1 * sw + sx1) * 3 + sc] * a1;
        const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
        if (resized_u8) resized_u8[((size_t)y * dw + x) * 3 + c] = (uint8_t)v;
        const float f = ((float)v - mean3[c]) / std3[c];
        out[((size_t)c * Hp + y) * Wp + x] = mxdet_f32_to_bf16(f);
      }
    }
  }
}

/* datasets: polygon lists -> instance masks [NG][H][W] u8; pixel-centre even-odd fill per polygon, union per instance */
API void oracle_polygon_masks(const float* verts, const int32_t* poly_start, const int32_t* inst_first, int NG, int H,
                              int W, uint8_t* masks) {
  for (int inst = 0; inst < NG; ++inst)
    for (int y = 0; y < H; ++y) {
      const float py = (float)y + 0.5f;
      for (int x = 0; x < W; ++x) {
        const float px = (float)x + 0.5f;
        int inside = 0;
        for (int p = inst_first[inst]; p < inst_first[inst + 1]; ++p) {
          const int vb = poly_start[p], ve = poly_start[p + 1];
          if (ve - vb < 3) continue;
          int par = 0;
          float ax = verts[2 * (ve - 1)], ay = verts[2 * (ve - 1) + 1];
          for (int v = vb; v < ve; ++v) {
            const float bx = verts[2 * v], by = verts[2 * v + 1];
            if ((ay <= py) != (by <= py)) {
              const float xi = ax + ((py - ay) * (bx - ax)) / (by - ay);
              if (px < xi) par ^= 1;
            }
            ax = bx; ay = by;
          }
          inside |= par;
        }
        masks[((size_t)inst * H + y) * W + x] = (uint8_t)inside;
      }
    }
}

/* core/mask inference paste-back: see include/mxdet.h mxdet_mask_paste. logits [R][S][S][Cpad] f32 (bf16-valued). */
static void o_paste_coef(int d, int len, int S, int* s0, int* s1, float* f) {
  float x = (float)(((double)d + 0.5) * ((double)S / (double)len) - 0.5);
  int s = (int)floorf(x);
  x -= (float)s;
  if (s < 0) { x = 0.0f; s = 0; }
  if (s >= S - 1) { x = 0.0f; s = S - 1; }
  *s0 = s; *s1 = s + 1 < S ? s + 1 : S - 1; *f = x;
}
API void oracle_mask_paste(const float* logits, const float* dets, int R, int S, int Cpad, int H, int W, float thresh,
                           uint8_t* masks) {
  float* p = (float*)malloc(sizeof(float) * (size_t)S * S);
  for (int r = 0; r < R; ++r) {
    uint8_t* out = masks + (size_t)r * H * W;
    memset(out, 0, (size_t)H * W);
    const float* d = dets + (size_t)r * 6;
    const int c = (int)d[5];
    if (!(c > 0 && c <= Cpad)) continue;
    for (int i = 0; i < S * S; ++i) p[i] = o_sigmoid(logits[((size_t)r * S * S + i) * Cpad + (c - 1)]);
    const int bx1 = (int)rintf(d[0]), by1 = (int)rintf(d[1]), bx2 = (int)rintf(d[2]), by2 = (int)rintf(d[3]);
    const int bw = bx2 - bx1 + 1, bh = by2 - by1 + 1;
    if (bw <= 0 || bh <= 0) continue;
    for (int y = by1 < 0 ? 0 : by1; y <= by2 && y < H; ++y) {
      int sy0, sy1; float fy;
      o_paste_coef(y - by1, bh, S, &sy0, &sy1, &fy);
      for (int x = bx1 < 0 ? 0 : bx1; x <= bx2 && x < W; ++x) {
        int sx0, sx1; float fx;
        o_paste_coef(x - bx1, bw, S, &sx0, &sx1, &fx);
        const float top = p[sy0 * S + sx0] * (1.0f - fx) + p[sy0 * S + sx1] * fx;
        const float bot = p[sy1 * S + sx0] * (1.0f - fx) + p[sy1 * S + sx1] * fx;
        const float v = top * (1.0f - fy) + bot * fy;
        out[(size_t)y * W + x] = v > thresh ? 1 : 0;
      }
    }
  }
  free(p);
}

/* One-stage test-time detection, second half (see include/mxdet.h mxdet_retina_detect): sparse candidates -- one class
 * per candidate, cls 0 = none -- through per-class threshold / sort / greedy NMS and the per-image cut. */
API float oracle_sigmoid(float z) { return o_sigmoid(z); }
API void oracle_class_nms_topk(const float* boxes, const float* scores, const int32_t* cls, const int32_t* num, int N,
                               int R, int C, float score_thresh, float nms_thresh, int max_det, float* dets,
                               int32_t* num_dets) {
  o_det* cand = (o_det*)malloc(sizeof(o_det) * (size_t)(R > 0 ? R : 1));
  o_det* kept = (o_det*)malloc(sizeof(o_det) * (size_t)(R > 0 ? R : 1) * (size_t)(C > 0 ? C : 1));
  float* bx = (float*)malloc(sizeof(float) * 4 * (size_t)(R > 0 ? R : 1));
  int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
  for (int n = 0; n < N; ++n) {
    int nr = num[n]; nr = nr > R ? R : (nr < 0 ? 0 : nr);
    int nkept = 0;
    for (int c = 1; c <= C; ++c) {
      int nc = 0;
      for (int i = 0; i < nr; ++i) {
        const size_t r = (size_t)n * R + i;
        if (cls[r] == c && scores[r] > score_thresh) {
          cand[nc].key = o_float_key(scores[r]); cand[nc].roi = (uint32_t)i; cand[nc].cls = (uint32_t)c; cand[nc].score = scores[r];
          memcpy(cand[nc].box, boxes + r * 4, 16);
          ++nc;
        }
      }
      qsort(cand, (size_t)nc, sizeof(o_det), o_cmp_det_cls);
      for (int j = 0; j < nc; ++j) memcpy(bx + 4 * j, cand[j].box, 16);
      int nk = oracle_nms(bx, nc, NULL, nms_thresh, max_det, keep);
      for (int j = 0; j < nk; ++j) kept[nkept++] = cand[keep[j]];
    }
    qsort(kept, (size_t)nkept, sizeof(o_det), o_cmp_det_all);
    int nout = nkept < max_det ? nkept : max_det;
    for (int j = 0; j < max_det; ++j) {
      float* d = dets + ((size_t)n * max_det + j) * 6;
      if (j < nout) { memcpy(d, kept[j].box, 16); d[4] = kept[j].score; d[5] = (float)kept[j].cls; }
      else { d[0] = d[1] = d[2] = d[3] = d[4] = 0.0f; d[5] = -1.0f; }
    }
    num_dets[n] = nout;
  }
  free(cand); free(kept); free(bx); free(keep);
}
