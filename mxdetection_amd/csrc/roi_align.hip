// roi_align.hip -- multi-level RoIAlign forward / backward over a channels-last bf16 feature pyramid.
//
// Slot: roi_extractors (/root/reference/README.md:32); MXNet role contrib.ROIAlign (README.md:37,
// Detectron aligned=False semantics), one launch for all pyramid levels instead of one op per level
// plus a concat/gather. Channels-last makes every bilinear tap a contiguous C*2-byte row segment:
// a lane owns 8 channels (one 16-B load per tap), so a 256-channel tap is one 512-B coalesced read.
// HBM/L2-gather bound: algorithmic bytes = R*PH*PW*C*2 written + the touched feature rows read.
#include "common.h"

namespace mxdet {

struct FeatPyr {
  int num_levels, lvl_min, N;
  int H[8], W[8];
  float scale[8];
  void* feat[8];
};

struct RoiGeom {
  float start_w, start_h, bin_w, bin_h;
  int gh, gw, H, W, batch, lvl;
};

__device__ __forceinline__ RoiGeom roi_geom(const FeatPyr& f, const float* __restrict__ rois,
                                            const int32_t* __restrict__ levels, long long r, int PH,
                                            int PW, int sampling_ratio) {
  RoiGeom g;
  const float* q = rois + r * 5;
  int l = levels[r] - f.lvl_min;
  l = l < 0 ? 0 : (l >= f.num_levels ? f.num_levels - 1 : l);
  g.lvl = l;
  g.batch = (int)q[0];
  g.batch = g.batch < 0 ? 0 : (g.batch >= f.N ? f.N - 1 : g.batch);  // never index outside the batch
  float s = f.scale[l];
  g.start_w = q[1] * s;
  g.start_h = q[2] * s;
  float end_w = q[3] * s, end_h = q[4] * s;
  float rw = end_w - g.start_w, rh = end_h - g.start_h;
  rw = rw > 1.0f ? rw : 1.0f;
  rh = rh > 1.0f ? rh : 1.0f;
  g.bin_h = rh / (float)PH;
  g.bin_w = rw / (float)PW;
  if (sampling_ratio > 0) {
    g.gh = sampling_ratio;
    g.gw = sampling_ratio;
  } else {
    float ch = rh / (float)PH, cw = rw / (float)PW;
    int ih = (int)ch, iw = (int)cw;
    g.gh = ((float)ih < ch) ? ih + 1 : ih;
    g.gw = ((float)iw < cw) ? iw + 1 : iw;
  }
  g.H = f.H[l];
  g.W = f.W[l];
  return g;
}

struct Taps {
  int yl, yh, xl, xh;
  float w1, w2, w3, w4;
  bool valid;
};

__device__ __forceinline__ Taps bilinear_taps(float y, float x, int H, int W) {
  Taps t;
  t.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.0f) y = 0.0f;
  if (x <= 0.0f) x = 0.0f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  float ly = y - (float)yl, lx = x - (float)xl;
  float hy = 1.0f - ly, hx = 1.0f - lx;
  t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
  t.yl = yl; t.yh = yh; t.xl = xl; t.xh = xh;
  return t;
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// forward: one workgroup per roi; work item = (bin, 8-channel group), channel group fastest.
__global__ void __launch_bounds__(256)
roi_align_fwd_kernel(FeatPyr f, int C, const float* __restrict__ rois,
                     const int32_t* __restrict__ levels, int PH, int PW, int sampling_ratio,
                     uint16_t* __restrict__ out) {
  const long long r = blockIdx.x;
  const RoiGeom g = roi_geom(f, rois, levels, r, PH, PW, sampling_ratio);
  const int CG = C >> 3;
  const int items = PH * PW * CG;
  const uint16_t* feat = (const uint16_t*)f.feat[g.lvl] + (long long)g.batch * g.H * g.W * C;
  const float count = (float)(g.gh * g.gw);
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    int cg = it % CG;
    int bin = it / CG;
    int pw = bin % PW, ph = bin / PW;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
    for (int iy = 0; iy < g.gh; ++iy) {
      float y = g.start_h + (float)ph * g.bin_h;
      y = y + (((float)iy + 0.5f) * g.bin_h) / (float)g.gh;
      for (int ix = 0; ix < g.gw; ++ix) {
        float x = g.start_w + (float)pw * g.bin_w;
        x = x + (((float)ix + 0.5f) * g.bin_w) / (float)g.gw;
        Taps t = bilinear_taps(y, x, g.H, g.W);
        if (!t.valid) continue;
        const uint4 v1 = *(const uint4*)(feat + ((long long)t.yl * g.W + t.xl) * C + cg * 8);
        const uint4 v2 = *(const uint4*)(feat + ((long long)t.yl * g.W + t.xh) * C + cg * 8);
        const uint4 v3 = *(const uint4*)(feat + ((long long)t.yh * g.W + t.xl) * C + cg * 8);
        const uint4 v4 = *(const uint4*)(feat + ((long long)t.yh * g.W + t.xh) * C + cg * 8);
        float a[8], b[8], c[8], d[8];
        unpack8(v1, a); unpack8(v2, b); unpack8(v3, c); unpack8(v4, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float s = t.w1 * a[k];
          s = s + t.w2 * b[k];
          s = s + t.w3 * c[k];
          s = s + t.w4 * d[k];
          acc[k] = acc[k] + s;
        }
      }
    }
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t lo = mxdet_f32_to_bf16(acc[2 * k] / count);
      uint32_t hi = mxdet_f32_to_bf16(acc[2 * k + 1] / count);
      o[k] = lo | (hi << 16);
    }
    *(uint4*)(out + (r * PH * PW + bin) * C + cg * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// backward: work item = (bin, channel), channel fastest, so one wave's atomic instruction covers
// 64 consecutive channels = 256 contiguous bytes of the fp32 accumulator (the full-rate shape of
// global_atomic_add_f32, MI355X_MICROARCH.md "Global float atomics").
__global__ void __launch_bounds__(256)
roi_align_bwd_kernel(FeatPyr f, int C, const float* __restrict__ rois,
                     const int32_t* __restrict__ levels, int PH, int PW, int sampling_ratio,
                     const uint16_t* __restrict__ gout) {
  const long long r = blockIdx.x;
  const RoiGeom g = roi_geom(f, rois, levels, r, PH, PW, sampling_ratio);
  const int items = PH * PW * C;
  float* dfeat = (float*)f.feat[g.lvl] + (long long)g.batch * g.H * g.W * C;
  const float count = (float)(g.gh * g.gw);
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    int c = it % C;
    int bin = it / C;
    int pw = bin % PW, ph = bin / PW;
    float go = bf16_bits_to_f32(gout[(r * PH * PW + bin) * C + c]) / count;
    if (go == 0.0f) continue;
    for (int iy = 0; iy < g.gh; ++iy) {
      float y = g.start_h + (float)ph * g.bin_h;
      y = y + (((float)iy + 0.5f) * g.bin_h) / (float)g.gh;
      for (int ix = 0; ix < g.gw; ++ix) {
        float x = g.start_w + (float)pw * g.bin_w;
        x = x + (((float)ix + 0.5f) * g.bin_w) / (float)g.gw;
        Taps t = bilinear_taps(y, x, g.H, g.W);
        if (!t.valid) continue;
        atomicAdd(dfeat + ((long long)t.yl * g.W + t.xl) * C + c, t.w1 * go);
        atomicAdd(dfeat + ((long long)t.yl * g.W + t.xh) * C + c, t.w2 * go);
        atomicAdd(dfeat + ((long long)t.yh * g.W + t.xl) * C + c, t.w3 * go);
        atomicAdd(dfeat + ((long long)t.yh * g.W + t.xh) * C + c, t.w4 * go);
      }
    }
  }
}

static int fill(FeatPyr& d, const mxdet_feat_pyramid_t* f, int N, const char* who) {
  MXDET_REQUIRE(f != nullptr, MXDET_EINVAL, "%s: null pyramid", who);
  MXDET_REQUIRE(f->num_levels > 0 && f->num_levels <= 8, MXDET_ESHAPE, "%s: bad level count", who);
  memset(&d, 0, sizeof(d));
  d.num_levels = f->num_levels;
  d.lvl_min = f->lvl_min;
  d.N = N;
  for (int l = 0; l < f->num_levels; ++l) {
    MXDET_REQUIRE(f->H[l] > 0 && f->W[l] > 0 && f->feat[l], MXDET_EINVAL, "%s: level %d incomplete",
                  who, l);
    d.H[l] = f->H[l]; d.W[l] = f->W[l]; d.scale[l] = f->spatial_scale[l]; d.feat[l] = f->feat[l];
  }
  return MXDET_OK;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" int mxdet_roi_align_fwd(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C,
                                   const float* rois, const int32_t* levels, int64_t R, int32_t PH,
                                   int32_t PW, int32_t sampling_ratio, uint16_t* out,
                                   mxdet_stream_t stream) {
  clear_error();
  FeatPyr d;
  int rc = fill(d, f, N, "roi_align_fwd");
  if (rc) return rc;
  MXDET_REQUIRE(N > 0 && C > 0 && (C % 8) == 0 && PH > 0 && PW > 0 && R >= 0, MXDET_ESHAPE,
                "roi_align_fwd: bad shape (C must be a multiple of 8)");
  if (R == 0) return MXDET_OK;
  MXDET_REQUIRE(rois && levels && out, MXDET_EINVAL, "roi_align_fwd: null pointer");
  hipLaunchKernelGGL(roi_align_fwd_kernel, dim3((unsigned)R), dim3(256), 0, as_stream(stream), d, C,
                     rois, levels, PH, PW, sampling_ratio, out);
  return check_launch("roi_align_fwd");
}

extern "C" int mxdet_roi_align_bwd(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C,
                                   const float* rois, const int32_t* levels, int64_t R, int32_t PH,
                                   int32_t PW, int32_t sampling_ratio, const uint16_t* grad_out,
                                   mxdet_stream_t stream) {
  clear_error();
  FeatPyr d;
  int rc = fill(d, f, N, "roi_align_bwd");
  if (rc) return rc;
  MXDET_REQUIRE(N > 0 && C > 0 && PH > 0 && PW > 0 && R >= 0, MXDET_ESHAPE, "roi_align_bwd: bad shape");
  if (R == 0) return MXDET_OK;
  MXDET_REQUIRE(rois && levels && grad_out, MXDET_EINVAL, "roi_align_bwd: null pointer");
  hipLaunchKernelGGL(roi_align_bwd_kernel, dim3((unsigned)R), dim3(256), 0, as_stream(stream), d, C,
                     rois, levels, PH, PW, sampling_ratio, grad_out);
  return check_launch("roi_align_bwd");
}
