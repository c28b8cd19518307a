/*
 * mxdet_math.h -- deterministic scalar math shared by the HIP kernels and the C oracle.
 *
 * Everything here is built from IEEE-754 binary32 add / mul / div / floor and integer ops only,
 * written in one fixed operation order. Compiled with -ffp-contract=off on both sides (gcc for the
 * oracle, hipcc for gfx950) every function returns the same bits on host and device, which is what
 * lets box decode / target encode / sampling be compared bit-exact (SURVEY.md section 7 "Hard parts").
 *
 * The reference (/root/reference/README.md:37) delegates this arithmetic to MXNet 1.3.0, which is
 * not available; the conventions below are "convention chosen", see DESIGN.md section 3.
 */
#ifndef MXDET_MATH_H_
#define MXDET_MATH_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MXDET_HD __host__ __device__ __forceinline__
#else
#define MXDET_HD static inline
#endif

/* ---- bit casts ---------------------------------------------------------------------------- */
MXDET_HD uint32_t mxdet_f32_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
MXDET_HD float mxdet_bits_f32(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* ---- bf16 <-> f32 (round to nearest even; NaN stays NaN) ------------------------------------ */
MXDET_HD uint16_t mxdet_f32_to_bf16(float f) {
  uint32_t u = mxdet_f32_bits(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
MXDET_HD float mxdet_bf16_to_f32(uint16_t h) { return mxdet_bits_f32(((uint32_t)h) << 16); }

/* ---- order-preserving key for floats: larger float -> larger key (total order, -0 < +0) ----- */
MXDET_HD uint32_t mxdet_float_key(float f) {
  uint32_t u = mxdet_f32_bits(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

/* ---- expf: |rel err| ~ 2e-7, identical bits on host and device ------------------------------ */
MXDET_HD float mxdet_expf(float x) {
  if (x != x) return x;
  if (x > 88.0f) x = 88.0f;
  if (x < -87.0f) x = -87.0f;
  const float log2e = 1.44269504088896341f;
  const float ln2_hi = 0.693359375f;          /* 355/512, exact in 10 bits */
  const float ln2_lo = -2.12194440e-4f;
  float t = x * log2e;
  float n = t + 0.5f;
  /* floor without libm so host and device cannot disagree */
  float nf = (float)(int32_t)n;
  if (nf > n) nf = nf - 1.0f;
  float r = x - nf * ln2_hi;
  r = r - nf * ln2_lo;
  /* exp(r) on [-0.35, 0.35], degree-6 Taylor/minimax (cephes expf coefficients) */
  float p = 1.9875691500e-4f;
  p = p * r + 1.3981999507e-3f;
  p = p * r + 8.3334519073e-3f;
  p = p * r + 4.1665795894e-2f;
  p = p * r + 1.6666665459e-1f;
  p = p * r + 5.0000001201e-1f;
  float r2 = r * r;
  float e = p * r2;
  e = e + r;
  e = e + 1.0f;
  int32_t ni = (int32_t)nf;
  float scale = mxdet_bits_f32((uint32_t)(ni + 127) << 23);
  return e * scale;
}

/* ---- logf for x > 0 (cephes-style), identical bits on host and device ------------------------ */
MXDET_HD float mxdet_logf(float x) {
  if (x != x) return x;
  if (x <= 0.0f) return (x == 0.0f) ? -3.402823466e+38f : mxdet_bits_f32(0x7fc00000u);
  uint32_t u = mxdet_f32_bits(x);
  int32_t e = 0;
  if (u < 0x00800000u) { /* subnormal: scale up by 2^23 */
    x = x * 8388608.0f;
    u = mxdet_f32_bits(x);
    e = -23;
  }
  e += (int32_t)(u >> 23) - 126;
  float m = mxdet_bits_f32((u & 0x007fffffu) | 0x3f000000u); /* m in [0.5, 1) */
  if (m < 0.707106781186547524f) {
    e = e - 1;
    m = m + m;
  }
  float f = m - 1.0f;
  float z = f * f;
  float y = 7.0376836292e-2f;
  y = y * f + -1.1514610310e-1f;
  y = y * f + 1.1676998740e-1f;
  y = y * f + -1.2420140846e-1f;
  y = y * f + 1.4249322787e-1f;
  y = y * f + -1.6668057665e-1f;
  y = y * f + 2.0000714765e-1f;
  y = y * f + -2.4999993993e-1f;
  y = y * f + 3.3333331174e-1f;
  y = y * f;
  y = y * z;
  float fe = (float)e;
  y = y + fe * -2.12194440e-4f;
  y = y - 0.5f * z;
  float res = f + y;
  res = res + fe * 0.693359375f;
  return res;
}

/* ---- Philox4x32-10 counter-based RNG (Salmon et al. 2011) ------------------------------------ */
typedef struct { uint32_t v[4]; } mxdet_u32x4;

MXDET_HD uint32_t mxdet_mulhi32(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

MXDET_HD mxdet_u32x4 mxdet_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                        uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = mxdet_mulhi32(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = mxdet_mulhi32(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  mxdet_u32x4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

/* Sampling key of element `idx` of stream `stream` (0 = RPN fg, 1 = RPN bg, 2 = RCNN fg, 3 = RCNN bg)
 * for (seed, step, image). The elements with the smallest (key, idx) are the ones sampled. */
MXDET_HD uint32_t mxdet_sample_key(uint32_t seed, uint32_t step, uint32_t image, uint32_t stream,
                                  uint32_t idx) {
  mxdet_u32x4 r = mxdet_philox4x32_10(idx, stream, image, step, seed, 0x6d786474u /* "mxdt" */);
  return r.v[0];
}

/* ---- box helpers (legacy "+1" pixel convention of the py-faster-rcnn / mx-rcnn lineage) ------- */
MXDET_HD float mxdet_iou(float ax1, float ay1, float ax2, float ay2, float bx1, float by1,
                         float bx2, float by2) {
  float ix1 = ax1 > bx1 ? ax1 : bx1;
  float iy1 = ay1 > by1 ? ay1 : by1;
  float ix2 = ax2 < bx2 ? ax2 : bx2;
  float iy2 = ay2 < by2 ? ay2 : by2;
  float iw = ix2 - ix1 + 1.0f;
  float ih = iy2 - iy1 + 1.0f;
  if (iw <= 0.0f || ih <= 0.0f) return 0.0f;
  float inter = iw * ih;
  float aw = ax2 - ax1 + 1.0f, ah = ay2 - ay1 + 1.0f;
  float bw = bx2 - bx1 + 1.0f, bh = by2 - by1 + 1.0f;
  float aa = aw * ah;
  float ab = bw * bh;
  float ua = aa + ab;
  ua = ua - inter;
  return inter / ua;
}

#define MXDET_BBOX_XFORM_CLIP 4.135166556742356f /* log(1000/16) */

/* decode deltas (dx,dy,dw,dh) at box (x1,y1,x2,y2); out clipped to the image [0,W-1]x[0,H-1] */
MXDET_HD void mxdet_decode_clip(float x1, float y1, float x2, float y2, float dx, float dy, float dw,
                               float dh, float im_h, float im_w, float* o) {
  float w = x2 - x1 + 1.0f;
  float h = y2 - y1 + 1.0f;
  float cx = x1 + 0.5f * (w - 1.0f);
  float cy = y1 + 0.5f * (h - 1.0f);
  if (dw > MXDET_BBOX_XFORM_CLIP) dw = MXDET_BBOX_XFORM_CLIP;
  if (dh > MXDET_BBOX_XFORM_CLIP) dh = MXDET_BBOX_XFORM_CLIP;
  float pcx = dx * w;
  pcx = pcx + cx;
  float pcy = dy * h;
  pcy = pcy + cy;
  float pw = mxdet_expf(dw) * w;
  float ph = mxdet_expf(dh) * h;
  float hw = 0.5f * (pw - 1.0f);
  float hh = 0.5f * (ph - 1.0f);
  float ox1 = pcx - hw, oy1 = pcy - hh, ox2 = pcx + hw, oy2 = pcy + hh;
  float mx = im_w - 1.0f, my = im_h - 1.0f;
  ox1 = ox1 < 0.0f ? 0.0f : (ox1 > mx ? mx : ox1);
  oy1 = oy1 < 0.0f ? 0.0f : (oy1 > my ? my : oy1);
  ox2 = ox2 < 0.0f ? 0.0f : (ox2 > mx ? mx : ox2);
  oy2 = oy2 < 0.0f ? 0.0f : (oy2 > my ? my : oy2);
  o[0] = ox1; o[1] = oy1; o[2] = ox2; o[3] = oy2;
}

/* encode gt box g relative to example box e */
MXDET_HD void mxdet_encode(float ex1, float ey1, float ex2, float ey2, float gx1, float gy1,
                          float gx2, float gy2, float* o) {
  float ew = ex2 - ex1 + 1.0f, eh = ey2 - ey1 + 1.0f;
  float ecx = ex1 + 0.5f * (ew - 1.0f), ecy = ey1 + 0.5f * (eh - 1.0f);
  float gw = gx2 - gx1 + 1.0f, gh = gy2 - gy1 + 1.0f;
  float gcx = gx1 + 0.5f * (gw - 1.0f), gcy = gy1 + 0.5f * (gh - 1.0f);
  o[0] = (gcx - ecx) / ew;
  o[1] = (gcy - ecy) / eh;
  o[2] = mxdet_logf(gw / ew);
  o[3] = mxdet_logf(gh / eh);
}

/* FPN level of a roi: k = clamp(floor(4 + log2(sqrt(w*h)/224)), 2, 5), evaluated with exact
 * area thresholds (112^2, 224^2, 448^2) instead of log2 so there is nothing to round. */
MXDET_HD int mxdet_fpn_level(float x1, float y1, float x2, float y2) {
  float w = x2 - x1 + 1.0f, h = y2 - y1 + 1.0f;
  float a = w * h;
  if (!(a >= 12544.0f)) return 2;
  if (a < 50176.0f) return 3;
  if (a < 200704.0f) return 4;
  return 5;
}

/* smooth-L1 with sigma: 0.5*(sigma*x)^2 if |x| < 1/sigma^2 else |x| - 0.5/sigma^2 */
MXDET_HD float mxdet_smooth_l1(float x, float sigma2) {
  float ax = x < 0.0f ? -x : x;
  float inv = 1.0f / sigma2;
  if (ax < inv) {
    float t = 0.5f * sigma2;
    t = t * x;
    return t * x;
  }
  return ax - 0.5f * inv;
}
MXDET_HD float mxdet_smooth_l1_grad(float x, float sigma2) {
  float ax = x < 0.0f ? -x : x;
  float inv = 1.0f / sigma2;
  if (ax < inv) return sigma2 * x;
  return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
}

#endif /* MXDET_MATH_H_ */
