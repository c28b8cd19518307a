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

// ---- backward, gather form --------------------------------------------------------------------------------------------
// The scatter form above is bound by the fp32-atomic rate (1024 rois x 49 bins x 4 samples x 4 corners x 256 channels =
// 205 M atomics: 310 us, plus a zero-fill and a finalize pass over 182 MB of fp32 accumulators) and its sums depend on
// the order the atomics land in. The gather form turns it around: bilinear weights are separable, so for every roi a
// table of its PH*gh row samples (yl, yh, hy, ly, valid) and PW*gw column samples is built once; every pyramid row gets
// the list of rois that touch it (ascending roi index); one workgroup then owns 16 consecutive pixels of one row (a lane
// owns a channel, 16 fp32 sums in registers), walks the row's rois, their matching row samples and column samples,
// loads grad_out[r][bin][c] (coalesced, L2-resident: 25 MB) and adds w * go -- no atomics, fixed order (roi, row sample,
// yl before yh, column sample, xl before xh), results written straight to the bf16 gradient map (optionally added to
// what is there). ~50 us instead of ~410, and bit-reproducible.
constexpr int kRoiMaxSamples = 32;    // PH * sampling_ratio and PW * sampling_ratio must not exceed this
constexpr int kRoiTileW = 16;

struct RoiTab {                        // per roi
  int n, lvl, ylo, yhi, xlo, xhi;      // pixel bounding box of its valid samples (ylo > yhi: none)
  short yl[kRoiMaxSamples], yh[kRoiMaxSamples], xl[kRoiMaxSamples], xh[kRoiMaxSamples];
  float hy[kRoiMaxSamples], ly[kRoiMaxSamples], hx[kRoiMaxSamples], lx[kRoiMaxSamples];
  unsigned char vy[kRoiMaxSamples], vx[kRoiMaxSamples];
};

struct RoiBox { int n, lvl, ylo, yhi, xlo, xhi, pad0, pad1; };   // the part of RoiTab the list builders scan: compact (32 B)

// One WAVE per roi: lanes 0..31 compute row sample `lane`, lanes 32..63 column sample `lane - 32` (one thread per roi
// walked 28 samples in a serial loop: 16 waves on the whole chip, 73 us); the bounding box is a butterfly min / max.
__global__ void __launch_bounds__(64)
roi_bwd_tab_kernel(FeatPyr f, const float* __restrict__ rois, const int32_t* __restrict__ levels,
                   long long R, int PH, int PW, int sr, RoiTab* __restrict__ tab, RoiBox* __restrict__ box) {
  const long long r = (long long)blockIdx.x;
  const int lane = threadIdx.x;
  if (r >= R) return;
  const RoiGeom g = roi_geom(f, rois, levels, r, PH, PW, sr);
  RoiTab& t = tab[r];
  const bool isx = lane >= 32;
  const int s = lane & 31;
  const int NS = isx ? PW * g.gw : PH * g.gh;
  int lo = 1 << 30, hi = -1;
  if (s < NS) {
    if (!isx) {
      const int ph = s / g.gh, iy = s - ph * g.gh;
      float y = g.start_h + (float)ph * g.bin_h;
      y = y + (((float)iy + 0.5f) * g.bin_h) / (float)g.gh;
      const Taps tp = bilinear_taps(y, 0.0f, g.H, g.W);       // the row part does not depend on x
      const bool vy = !(y < -1.0f || y > (float)g.H);
      // hy, ly exactly as bilinear_taps derives them
      float yy = y <= 0.0f ? 0.0f : y;
      if ((int)yy >= g.H - 1) yy = (float)(g.H - 1);
      const float lyv = yy - (float)tp.yl, hyv = 1.0f - lyv;
      t.yl[s] = (short)tp.yl; t.yh[s] = (short)tp.yh; t.hy[s] = hyv; t.ly[s] = lyv; t.vy[s] = vy ? 1 : 0;
      if (vy) { lo = tp.yl; hi = tp.yh; }
    } else {
      const int pw = s / g.gw, ix = s - pw * g.gw;
      float x = g.start_w + (float)pw * g.bin_w;
      x = x + (((float)ix + 0.5f) * g.bin_w) / (float)g.gw;
      const Taps tp = bilinear_taps(0.0f, x, g.H, g.W);
      const bool vx = !(x < -1.0f || x > (float)g.W);
      float xx = x <= 0.0f ? 0.0f : x;
      if ((int)xx >= g.W - 1) xx = (float)(g.W - 1);
      const float lxv = xx - (float)tp.xl, hxv = 1.0f - lxv;
      t.xl[s] = (short)tp.xl; t.xh[s] = (short)tp.xh; t.hx[s] = hxv; t.lx[s] = lxv; t.vx[s] = vx ? 1 : 0;
      if (vx) { lo = tp.xl; hi = tp.xh; }
    }
  }
  // min / max inside each 32-lane half (rows | columns)
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const int olo = __shfl_xor(lo, d), ohi = __shfl_xor(hi, d);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  const int xlo = __shfl(lo, 32), xhi = __shfl(hi, 32);
  if (lane == 0) {
    t.n = g.batch; t.lvl = g.lvl;
    t.ylo = lo; t.yhi = hi; t.xlo = xlo; t.xhi = xhi;
    RoiBox bx;
    bx.n = g.batch; bx.lvl = g.lvl; bx.ylo = lo; bx.yhi = hi; bx.xlo = xlo; bx.xhi = xhi; bx.pad0 = 0; bx.pad1 = 0;
    box[r] = bx;
  }
}

struct RoiRows {                       // row r of level l, image n has id row0[l] + n * H[l] + r
  int row0[8], rows_total;
  int tiles_w[8], block0[9];           // workgroups: block0[l] + (n * H[l] + y) * tiles_w[l] + tile
};

// one wave per pyramid row: the rois whose valid samples touch it, in ascending index order (ballot compaction)
__global__ void __launch_bounds__(64)
roi_bwd_rows_kernel(FeatPyr f, RoiRows rr, const RoiBox* __restrict__ box, int R,
                    unsigned short* __restrict__ row_list, int* __restrict__ row_count) {
  const int row = (int)blockIdx.x, lane = threadIdx.x;
  if (row >= rr.rows_total) return;
  int l = 0;
  while (l + 1 < f.num_levels && row >= rr.row0[l + 1]) ++l;
  const int rel = row - rr.row0[l];
  const int n = rel / f.H[l], y = rel - n * f.H[l];
  int cnt = 0;
  unsigned short* dst = row_list + (size_t)row * R;
  for (int r0 = 0; r0 < R; r0 += 64) {
    const int r = r0 + lane;
    bool hit = false;
    if (r < R) {
      const RoiBox t = box[r];            // 32 B per lane, consecutive lanes consecutive rois
      hit = t.lvl == l && t.n == n && y >= t.ylo && y <= t.yhi && t.xlo <= t.xhi;
    }
    const unsigned long long m = __ballot(hit);
    if (hit) dst[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)r;
    cnt += __popcll(m);
  }
  if (lane == 0) row_count[row] = cnt;
}

// One WAVE per (pyramid row, 16-pixel segment); a lane owns 4 consecutive channels of each 256-channel chunk. All
// control is lane-parallel or register-resident: the roi's sample tables sit one sample per lane, the samples that
// hit this row / this segment come out of two ballots, and the loops walk the set bits with v_readlane -- no memory
// latency inside them.
__global__ void __launch_bounds__(64)
roi_align_bwd_gather_kernel(FeatPyr f, RoiRows rr, int C, const RoiTab* __restrict__ tab, const RoiBox* __restrict__ box, int R,
                            const unsigned short* __restrict__ row_list, const int* __restrict__ row_count,
                            int PH, int PW, int sr, const uint16_t* __restrict__ gout, int accumulate) {
  int l = 0;
  const int bid = (int)blockIdx.x, lane = threadIdx.x;
  while (l + 1 < f.num_levels && bid >= rr.block0[l + 1]) ++l;
  const int rel = bid - rr.block0[l];
  const int tw = rr.tiles_w[l];
  const int rowrel = rel / tw, tile = rel - rowrel * tw;
  const int H = f.H[l], W = f.W[l];
  const int n = rowrel / H, Y = rowrel - n * H;
  const int x0 = tile * kRoiTileW;
  const int x1 = x0 + kRoiTileW - 1 < W - 1 ? x0 + kRoiTileW - 1 : W - 1;
  const int row = rr.row0[l] + rowrel;
  const int cnt = row_count[row];
  const unsigned short* list = row_list + (size_t)row * R;
  const int NSY = PH * sr, NSX = PW * sr;
  const float count = (float)(sr * sr);
  const bool pow2 = ((sr * sr) & (sr * sr - 1)) == 0;
  const float inv_count = 1.0f / count;
  uint16_t* out = (uint16_t*)f.feat[l] + ((size_t)(n * H + Y) * W + x0) * C;
  for (int cb = 0; cb < C; cb += 256) {          // uniform trip count: every lane takes part in the ballots below
    const bool live = cb + lane * 4 < C;
    const int c0 = live ? cb + lane * 4 : 0;
    // four flat register arrays (one per channel of the lane), each indexed with a wave-uniform register index
    float a0[kRoiTileW], a1[kRoiTileW], a2[kRoiTileW], a3[kRoiTileW];
#pragma unroll
    for (int j = 0; j < kRoiTileW; ++j) { a0[j] = 0.0f; a1[j] = 0.0f; a2[j] = 0.0f; a3[j] = 0.0f; }
    bool touched = false;                          // wave-uniform: did any roi reach this segment?
    for (int base = 0; base < cnt; base += 64) {
     // 64 rois of the row at a time: lane k tests roi base+k against this 16-pixel segment; only the hits (one in ten
     // on the finest level) get their tables loaded
     const int kk = base + lane;
     const int rk = kk < cnt ? list[kk] : 0;
     const bool hitk = kk < cnt && box[rk].xhi >= x0 && box[rk].xlo <= x1;
     unsigned long long mr = __ballot(hitk);
     while (mr) {
      const int kb = __ffsll((long long)mr) - 1;
      mr &= mr - 1;
      const int r = __builtin_amdgcn_readlane(rk, kb);
      const RoiTab& t = tab[r];
      // lane s holds row sample s and column sample s of this roi
      const int ls = lane < kRoiMaxSamples ? lane : 0;
      const int yl_ = t.yl[ls], yh_ = t.yh[ls], xl_ = t.xl[ls], xh_ = t.xh[ls];
      const float hy_ = t.hy[ls], ly_ = t.ly[ls], hx_ = t.hx[ls], lx_ = t.lx[ls];
      const bool vy_ = lane < NSY && t.vy[ls] != 0, vx_ = lane < NSX && t.vx[ls] != 0;
      unsigned long long my = __ballot(vy_ && (yl_ == Y || yh_ == Y));
      const unsigned long long mx = __ballot(vx_ && xh_ >= x0 && xl_ <= x1);
      if (mx == 0ull || my == 0ull) continue;
      touched = true;
      const uint16_t* g0 = gout + (size_t)r * PH * PW * C + c0;
      while (my) {
        const int sy = __ffsll((long long)my) - 1;
        my &= my - 1;
        const int yl = __builtin_amdgcn_readlane(yl_, sy), yh = __builtin_amdgcn_readlane(yh_, sy);
        const float hy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hy_), sy));
        const float ly = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ly_), sy));
        const int ph = sy / sr;
        unsigned long long m2 = mx;
        // two column samples per trip: their gradient loads are independent and overlap (the loop is otherwise one
        // L2 round trip per sample)
        while (m2) {
          const int sxa = __ffsll((long long)m2) - 1;
          m2 &= m2 - 1;
          const bool two = m2 != 0ull;
          const int sxb = two ? __ffsll((long long)m2) - 1 : sxa;
          if (two) m2 &= m2 - 1;
          const uint2 gva = *(const uint2*)(g0 + (size_t)(ph * PW + sxa / sr) * C);
          const uint2 gvb = *(const uint2*)(g0 + (size_t)(ph * PW + sxb / sr) * C);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            if (half == 1 && !two) break;
            const int sx = half ? sxb : sxa;
            const uint2 gv = half ? gvb : gva;
            const int xl = __builtin_amdgcn_readlane(xl_, sx), xh = __builtin_amdgcn_readlane(xh_, sx);
            const float hx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hx_), sx));
            const float lx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lx_), sx));
            float go[4] = {__uint_as_float(gv.x << 16), __uint_as_float(gv.x & 0xffff0000u),
                           __uint_as_float(gv.y << 16), __uint_as_float(gv.y & 0xffff0000u)};
            // grad / count: a power-of-two count (sampling_ratio 1, 2, 4) divides exactly as a multiplication
            if (pow2) { go[0] *= inv_count; go[1] *= inv_count; go[2] *= inv_count; go[3] *= inv_count; }
            else { go[0] /= count; go[1] /= count; go[2] /= count; go[3] /= count; }
            // corner order of the scatter form: (yl,xl) (yl,xh) (yh,xl) (yh,xh); a zero gradient adds nothing there
            // and adds +0.0 here, which leaves every sum unchanged
            const int jl = xl - x0, jh = xh - x0;
            const bool inl = (unsigned)jl < (unsigned)kRoiTileW, inh = (unsigned)jh < (unsigned)kRoiTileW;
            if (yl == Y) {
              const float wl = hy * hx, wh = hy * lx;
              if (inl) { a0[jl] = a0[jl] + wl * go[0]; a1[jl] = a1[jl] + wl * go[1]; a2[jl] = a2[jl] + wl * go[2]; a3[jl] = a3[jl] + wl * go[3]; }
              if (inh) { a0[jh] = a0[jh] + wh * go[0]; a1[jh] = a1[jh] + wh * go[1]; a2[jh] = a2[jh] + wh * go[2]; a3[jh] = a3[jh] + wh * go[3]; }
            }
            if (yh == Y) {
              const float wl = ly * hx, wh = ly * lx;
              if (inl) { a0[jl] = a0[jl] + wl * go[0]; a1[jl] = a1[jl] + wl * go[1]; a2[jl] = a2[jl] + wl * go[2]; a3[jl] = a3[jl] + wl * go[3]; }
              if (inh) { a0[jh] = a0[jh] + wh * go[0]; a1[jh] = a1[jh] + wh * go[1]; a2[jh] = a2[jh] + wh * go[2]; a3[jh] = a3[jh] + wh * go[3]; }
            }
          }
        }
      }
     }
    }
    if (accumulate && !touched) continue;          // nothing to add: leave the map as it is (no read-modify-write)
#pragma unroll
    for (int j = 0; j < kRoiTileW; ++j) {
      if (x0 + j >= W || !live) break;
      uint16_t* o = out + (size_t)j * C + c0;
      float v[4] = {a0[j], a1[j], a2[j], a3[j]};
      if (accumulate) {
        const uint2 e = *(const uint2*)o;
        v[0] = v[0] + __uint_as_float(e.x << 16); v[1] = v[1] + __uint_as_float(e.x & 0xffff0000u);
        v[2] = v[2] + __uint_as_float(e.y << 16); v[3] = v[3] + __uint_as_float(e.y & 0xffff0000u);
      }
      uint2 w;
      w.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
      w.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
      *(uint2*)o = w;
    }
  }
}

// ---- backward, segment form (default) ----------------------------------------------------------------------------------
// Two launches: a per-roi box pass (R threads) and the gather. A gather workgroup owns a tile of `rows_per_wg` pyramid
// rows x up to kSegWaves (4) consecutive 8-pixel segments (one WAVE per segment column; a lane owns 4 channels, 8 x 4 fp32 sums in
// registers per row). It first builds, in LDS and in ascending roi order, the list of rois whose valid samples touch its
// tile (one coalesced 32-B record per roi), then every wave walks the list once per row of the tile. Per (roi, row,
// segment) hit the bilinear weights are collapsed per pooling bin -- they are separable:
//     d[Y][X][c] += sum_ph ay(Y,ph) * sum_pw ax(X,pw) * dY[r][ph][pw][c],   ay(Y,ph) = sum of the row weights of bin ph's
// samples on row Y (likewise ax) -- so the wave loads the <= 3 x PW bins it needs in ONE round of independent loads
// (the sample-by-sample walk of the table form is one L2 round trip per sample pair), folds the rows
// (T[pw] = sum_by ay * dY) and distributes T over its 8 pixels with the column weights (lane 32+s holds column sample
// s; v_readlane hands them out as scalars). Fixed summation order (roi ascending, row bins, column bins): bit-
// reproducible. The sums differ from the oracle's sample-by-sample order in the last bits (tolerance in the test).
constexpr int kSegW = 8;          // pixels per wave
constexpr int kSegWaves = 4;      // waves (segment columns) per workgroup
constexpr int kSegChunk = 1024;   // rois per list round

struct RoiSegGrid {
  int block0[9];                  // first workgroup of level l
  int chunks[8];                  // workgroups along a row
  int segs_per_wg[8];             // segments each of them owns (<= kSegWaves)
  int rows_per_wg[8];             // rows of a tile
  int bands[8];                   // tiles along a column = ceil(H / rows_per_wg)
};

struct RoiSeg {                   // per roi, 32 B
  int nl;                         // image | level << 16 (level 0xffff: no valid sample)
  unsigned yy, xx;                // ylo | yhi << 16, xlo | xhi << 16: pixel bounding box of its valid samples
  int r;
  float start_h, bin_h, start_w, bin_w;
};

struct AxisSample { int lo, hi; float wlo, whi; bool valid; };
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// value of lane + 1 inside a 16-lane row (DPP row_shl:1; the last lane of a row reads 0): the two samples of a bin
// never straddle a row, so this replaces a ds_bpermute round trip per weight
__device__ __forceinline__ float lane_plus1(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x101, 0xf, 0xf, true));
}

// coordinate of sample s of an axis (bin = s / sr), exactly as the forward / the oracle computes it
template <int SR>
__device__ __forceinline__ float roi_axis_coord(float start, float bin, int s, int sr) {
  const int b = SR ? s / SR : s / sr, i = s - b * (SR ? SR : sr);
  float v = start + (float)b * bin;
  return v + (((float)i + 0.5f) * bin) / (float)(SR ? SR : sr);
}

// the per-axis half of bilinear_taps
__device__ __forceinline__ AxisSample roi_axis_sample(float v, int L) {
  AxisSample a;
  a.valid = !(v < -1.0f || v > (float)L);
  if (v <= 0.0f) v = 0.0f;
  int lo = (int)v;
  if (lo >= L - 1) { a.hi = lo = L - 1; v = (float)lo; } else { a.hi = lo + 1; }
  a.lo = lo;
  a.whi = v - (float)lo;
  a.wlo = 1.0f - a.whi;
  return a;
}

// pixel range [lo, hi] touched by the valid samples of one axis (lo > hi: none)
__device__ __forceinline__ void roi_axis_range(float start, float bin, int NS, int sr, int L, int* lo, int* hi) {
  int s0 = 0, s1 = NS - 1;
  while (s0 <= s1 && !roi_axis_sample(roi_axis_coord<0>(start, bin, s0, sr), L).valid) ++s0;
  while (s1 >= s0 && !roi_axis_sample(roi_axis_coord<0>(start, bin, s1, sr), L).valid) --s1;
  if (s0 > s1) { *lo = 1; *hi = 0; return; }
  *lo = roi_axis_sample(roi_axis_coord<0>(start, bin, s0, sr), L).lo;
  *hi = roi_axis_sample(roi_axis_coord<0>(start, bin, s1, sr), L).hi;
}

__global__ void __launch_bounds__(256)
roi_bwd_box_kernel(FeatPyr f, const float* __restrict__ rois, const int32_t* __restrict__ levels, int R, int PH, int PW,
                   int sr, RoiSeg* __restrict__ seg) {
  const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (r >= R) return;
  const RoiGeom g = roi_geom(f, rois, levels, r, PH, PW, sr);
  int ylo, yhi, xlo, xhi;
  roi_axis_range(g.start_h, g.bin_h, PH * sr, sr, g.H, &ylo, &yhi);
  roi_axis_range(g.start_w, g.bin_w, PW * sr, sr, g.W, &xlo, &xhi);
  RoiSeg o;
  const bool any = ylo <= yhi && xlo <= xhi;
  o.nl = g.batch | ((any ? g.lvl : 0xffff) << 16);
  o.yy = (unsigned)ylo | ((unsigned)yhi << 16);
  o.xx = (unsigned)xlo | ((unsigned)xhi << 16);
  o.r = r;
  o.start_h = g.start_h; o.bin_h = g.bin_h; o.start_w = g.start_w; o.bin_w = g.bin_w;
  seg[r] = o;
}

template <int PWT, int SR>        // PW <= PWT; SR = the sampling ratio when it is 2 (0: run-time value)
__global__ void __launch_bounds__(kSegWaves * 64)
roi_bwd_seg_kernel(FeatPyr f, RoiSegGrid gr, int C, const RoiSeg* __restrict__ seg, int R, int PH, int PW, int sr_,
                   const uint16_t* __restrict__ gout, int accumulate) {
  __shared__ unsigned e_xx[kSegChunk], e_yy[kSegChunk];   // the fields the walk scans, one word per lane
  __shared__ int e_r[kSegChunk];
  __shared__ float4 e_g[kSegChunk];                        // start_h, bin_h, start_w, bin_w
  __shared__ int wcnt[kSegWaves];
  __shared__ int s_cnt;
  const int sr = SR ? SR : sr_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int l = 0;
  const int bid = (int)blockIdx.x;
  while (l + 1 < f.num_levels && bid >= gr.block0[l + 1]) ++l;
  const int rel = bid - gr.block0[l];
  const int per_img = gr.bands[l] * gr.chunks[l];
  const int n = rel / per_img, rem = rel - n * per_img;
  const int band = rem / gr.chunks[l], chunk = rem - band * gr.chunks[l];
  const int H = f.H[l], W = f.W[l];
  const int Y0 = band * gr.rows_per_wg[l];
  const int Y1 = (Y0 + gr.rows_per_wg[l] - 1 < H - 1) ? Y0 + gr.rows_per_wg[l] - 1 : H - 1;
  const int spw = gr.segs_per_wg[l];
  const int X0 = chunk * spw * kSegW;                                    // the workgroup's pixel span
  const int X1 = (X0 + spw * kSegW - 1 < W - 1) ? X0 + spw * kSegW - 1 : W - 1;
  const int x0 = X0 + wid * kSegW;                                       // this wave's segment column
  const int x1 = (x0 + kSegW - 1 < W - 1) ? x0 + kSegW - 1 : W - 1;
  const bool wave_live = wid < spw && x0 < W;
  const int cb = (int)blockIdx.y * 256;
  const bool live = cb + lane * 4 < C;
  const int c0 = live ? cb + lane * 4 : 0;
  const int NSY = PH * sr, NSX = PW * sr;
  const float count = (float)(sr * sr);
  const bool pow2 = ((sr * sr) & (sr * sr - 1)) == 0;
  const float inv_count = 1.0f / count;
  const int key = n | (l << 16);
  const bool isx = lane >= 32;
  const int s = lane & 31;

  // A one-row tile keeps its sums in registers across list rounds and writes once; a multi-row tile writes every row
  // at the end of each round (the host only makes multi-row tiles when all rois fit one round).
  const bool multi_row = Y1 > Y0;
  f32x2_t acc[kSegW][2];
#pragma unroll
  for (int j = 0; j < kSegW; ++j) { acc[j][0] = (f32x2_t){0.f, 0.f}; acc[j][1] = (f32x2_t){0.f, 0.f}; }
  bool touched = false;
  for (int rc = 0; rc < R; rc += kSegChunk) {
    // ---- the list of this round: rois rc .. rc+kSegChunk-1 that touch the tile, ascending ----
    __syncthreads();                      // every wave is done with the previous round's list
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const int rend = rc + kSegChunk < R ? rc + kSegChunk : R;
    for (int rb = rc; rb < rend; rb += kSegWaves * 64) {
      const int r = rb + tid;
      bool hit = false;
      RoiSeg t;
      if (r < rend) {
        t = seg[r];
        hit = t.nl == key && (int)(t.yy >> 16) >= Y0 && (int)(t.yy & 0xffffu) <= Y1 &&
              (int)(t.xx >> 16) >= X0 && (int)(t.xx & 0xffffu) <= X1;
      }
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wcnt[wid] = __popcll(m);
      __syncthreads();
      int base = s_cnt;
#pragma unroll
      for (int w = 0; w < kSegWaves; ++w) base += (w < wid) ? wcnt[w] : 0;
      if (hit) {
        const int k = base + __popcll(m & ((1ull << lane) - 1ull));
        e_xx[k] = t.xx; e_yy[k] = t.yy; e_r[k] = t.r;
        e_g[k] = make_float4(t.start_h, t.bin_h, t.start_w, t.bin_w);
      }
      __syncthreads();
      if (tid == 0) {
        int tt = s_cnt;
#pragma unroll
        for (int w = 0; w < kSegWaves; ++w) tt += wcnt[w];
        s_cnt = tt;
      }
      __syncthreads();
    }
    const int cnt = s_cnt;
    if (!wave_live) continue;             // (the barriers above are reached by every wave: `continue` re-enters the loop)
    const bool first_round = rc == 0, last_round = rend == R;

    for (int Y = Y0; Y <= Y1; ++Y) {
      if (multi_row) {
#pragma unroll
        for (int j = 0; j < kSegW; ++j) { acc[j][0] = (f32x2_t){0.f, 0.f}; acc[j][1] = (f32x2_t){0.f, 0.f}; }
        touched = false;
      }
      // ---- walk the list ----
      for (int base = 0; base < cnt; base += 64) {
        const int kk = base + lane;
        unsigned xk = 0u, yk = 0u;
        if (kk < cnt) { xk = e_xx[kk]; yk = e_yy[kk]; }
        const bool hitk = kk < cnt && (int)(xk >> 16) >= x0 && (int)(xk & 0xffffu) <= x1 &&
                          (int)(yk >> 16) >= Y && (int)(yk & 0xffffu) <= Y;
        unsigned long long mr = __ballot(hitk);
        while (mr) {
          const int kb = base + __ffsll((long long)mr) - 1;
          mr &= mr - 1;
          const int r = e_r[kb];
          const float4 gq = e_g[kb];
          const float gsh = gq.x, gbh = gq.y, gsw = gq.z, gbw = gq.w;
          // lane s < 32: row sample s; lane 32 + s: column sample s
          const float v = roi_axis_coord<SR>(isx ? gsw : gsh, isx ? gbw : gbh, s, sr);
          const AxisSample a = roi_axis_sample(v, isx ? W : H);
          const bool ok = a.valid && s < (isx ? NSX : NSY);
          const unsigned long long mb = __ballot(ok && !isx && (a.lo == Y || a.hi == Y));
          const unsigned long long mc = __ballot(ok && isx && a.hi >= x0 && a.lo <= x1);
          if (mb == 0ull || mc == 0ull) continue;
          touched = true;
          // per-bin weights: the first lane of each bin's sr samples gets the bin's sum
          const float wy = (ok && !isx) ? ((a.lo == Y ? a.wlo : 0.0f) + (a.hi == Y ? a.whi : 0.0f)) : 0.0f;
          float ayb = wy;
          if (SR == 2) ayb = ayb + lane_plus1(wy);
          else for (int i = 1; i < sr; ++i) ayb = ayb + __shfl_down(wy, i);
          if (pow2) ayb = ayb * inv_count;               // a power-of-two count divides exactly as a multiplication
          float axb[kSegW];
#pragma unroll
          for (int j = 0; j < kSegW; ++j) {
            const int X = x0 + j;
            const float wx = (ok && isx) ? ((a.lo == X ? a.wlo : 0.0f) + (a.hi == X ? a.whi : 0.0f)) : 0.0f;
            float t = wx;
            if (SR == 2) t = t + lane_plus1(wx);
            else for (int i = 1; i < sr; ++i) t = t + __shfl_down(wx, i);
            axb[j] = t;
          }
          const int pb0 = (__ffsll((long long)mb) - 1) / sr, pb1 = (63 - __clzll((long long)mb)) / sr;
          const int pc0 = (__ffsll((long long)mc) - 1 - 32) / sr, pc1 = (63 - __clzll((long long)mc) - 32) / sr;
          const uint16_t* g0 = gout + (size_t)r * PH * PW * C + c0;
#ifdef MXDET_ROI_ABL_NOMATH
          acc[0][0][0] += ayb + axb[0] + axb[7] + (float)(pb0 + pb1 + pc0 + pc1) + (float)g0[0];
          continue;
#endif
          for (int b0 = pb0; b0 <= pb1; b0 += 3) {
            uint2 g[3][PWT];
            float wyb[3];
#pragma unroll
            for (int by = 0; by < 3; ++by) {
              const bool rowok = b0 + by <= pb1;
              const int ph = rowok ? b0 + by : pb1;
              wyb[by] = rowok ? __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ayb), ph * sr)) : 0.0f;
#pragma unroll
              for (int pw = 0; pw < PWT; ++pw) {
                g[by][pw] = make_uint2(0u, 0u);
#ifndef MXDET_ROI_ABL_NOLOAD
                if (rowok && pw >= pc0 && pw <= pc1) g[by][pw] = *(const uint2*)(g0 + (size_t)(ph * PW + pw) * C);
#else
                if (rowok && pw >= pc0 && pw <= pc1) g[by][pw] = make_uint2((unsigned)r, (unsigned)ph);
#endif
              }
            }
#pragma unroll
            for (int pw = 0; pw < PWT; ++pw) {
              if (pw < pc0 || pw > pc1) continue;        // wave-uniform
              // (explicit fused multiply-adds, two channels per instruction: this file is built without contraction for
              // the bit-exact forward, and the walk is VALU-issue bound)
              f32x2_t T0 = {0.f, 0.f}, T1 = {0.f, 0.f};
#pragma unroll
              for (int by = 0; by < 3; ++by) {
                f32x2_t a0 = {__uint_as_float(g[by][pw].x << 16), __uint_as_float(g[by][pw].x & 0xffff0000u)};
                f32x2_t a1 = {__uint_as_float(g[by][pw].y << 16), __uint_as_float(g[by][pw].y & 0xffff0000u)};
                if (!pow2) { a0[0] /= count; a0[1] /= count; a1[0] /= count; a1[1] /= count; }
                const f32x2_t wv = {wyb[by], wyb[by]};
                T0 = __builtin_elementwise_fma(wv, a0, T0);
                T1 = __builtin_elementwise_fma(wv, a1, T1);
              }
#pragma unroll
              for (int j = 0; j < kSegW; ++j) {
                const float w = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(axb[j]), 32 + pw * sr));
                const f32x2_t wv = {w, w};
                acc[j][0] = __builtin_elementwise_fma(wv, T0, acc[j][0]);
                acc[j][1] = __builtin_elementwise_fma(wv, T1, acc[j][1]);
              }
            }
          }
        }
      }
      // ---- write row Y: a one-row tile after its last round, a multi-row tile every round (rounds after the first add) ----
      if (!live || (!multi_row && !last_round)) continue;
      const bool add = accumulate || (multi_row && !first_round);
      if (add && !touched) continue;                  // nothing to add: leave the map as it is (no read-modify-write)
      uint16_t* out = (uint16_t*)f.feat[l] + ((size_t)(n * H + Y) * W + x0) * C + c0;
#pragma unroll
      for (int j = 0; j < kSegW; ++j) {
        if (x0 + j >= W) break;
        uint16_t* o = out + (size_t)j * C;
        float v[4] = {acc[j][0][0], acc[j][0][1], acc[j][1][0], acc[j][1][1]};
        if (add) {
          const uint2 ee = *(const uint2*)o;
          v[0] = v[0] + __uint_as_float(ee.x << 16); v[1] = v[1] + __uint_as_float(ee.x & 0xffff0000u);
          v[2] = v[2] + __uint_as_float(ee.y << 16); v[3] = v[3] + __uint_as_float(ee.y & 0xffff0000u);
        }
        uint2 w;
        w.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
        w.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
        *(uint2*)o = w;
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

static size_t roi_gather_carve(const FeatPyr& d, long long R, RoiRows* rr, size_t* off_tab, size_t* off_list, size_t* off_cnt,
                               size_t* off_box) {
  int rows = 0, blocks = 0;
  for (int l = 0; l < d.num_levels; ++l) {
    rr->row0[l] = rows;
    rows += d.N * d.H[l];
    rr->tiles_w[l] = ceil_div(d.W[l], kRoiTileW);
    rr->block0[l] = blocks;
    blocks += d.N * d.H[l] * rr->tiles_w[l];
  }
  rr->block0[d.num_levels] = blocks;
  rr->rows_total = rows;
  size_t off = 0;
  *off_tab = off; off = align_up(off + (size_t)R * sizeof(RoiTab), 256);
  *off_list = off; off = align_up(off + (size_t)rows * (size_t)R * sizeof(unsigned short), 256);
  *off_cnt = off; off = align_up(off + (size_t)rows * sizeof(int), 256);
  *off_box = off; off = align_up(off + (size_t)R * sizeof(RoiBox), 256);
  return off;
}

extern "C" size_t mxdet_roi_align_bwd_gather_workspace_bytes(const mxdet_feat_pyramid_t* f, int32_t N, int64_t R) {
  if (!f || N <= 0 || R <= 0 || f->num_levels <= 0 || f->num_levels > 8) return 0;
  FeatPyr d;
  memset(&d, 0, sizeof(d));
  d.num_levels = f->num_levels; d.N = N;
  for (int l = 0; l < f->num_levels; ++l) { d.H[l] = f->H[l]; d.W[l] = f->W[l]; }
  RoiRows rr; size_t a, b, c, e;
  return roi_gather_carve(d, R, &rr, &a, &b, &c, &e);
}

// mode: 0 = records + gather, 1 = records only (prepare), 2 = gather only (records already in the workspace)
static int roi_bwd_gather_impl(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C, const float* rois,
                               const int32_t* levels, int64_t R, int32_t PH, int32_t PW,
                               int32_t sampling_ratio, const uint16_t* grad_out, int32_t accumulate,
                               void* workspace, size_t workspace_bytes, mxdet_stream_t stream, int mode) {
  clear_error();
  FeatPyr d;
  int rc = fill(d, f, N, "roi_align_bwd_gather");
  if (rc) return rc;
  MXDET_REQUIRE(N > 0 && C > 0 && (C % 4) == 0 && PH > 0 && PW > 0 && R >= 0, MXDET_ESHAPE, "roi_align_bwd_gather: bad shape (C must be a multiple of 4)");
  MXDET_REQUIRE(sampling_ratio > 0 && PH * sampling_ratio <= kRoiMaxSamples && PW * sampling_ratio <= kRoiMaxSamples,
                MXDET_ESHAPE, "roi_align_bwd_gather: needs 0 < sampling_ratio and at most %d samples per axis", kRoiMaxSamples);
  MXDET_REQUIRE(R <= 65535, MXDET_ESHAPE, "roi_align_bwd_gather: at most 65535 rois");
  for (int l = 0; l < d.num_levels; ++l)
    MXDET_REQUIRE(d.H[l] < 32768 && d.W[l] < 32768, MXDET_ESHAPE, "roi_align_bwd_gather: level %d too large", l);
  MXDET_REQUIRE(rois && levels && (grad_out || mode == 1), MXDET_EINVAL, "roi_align_bwd_gather: null pointer");
  hipStream_t s = as_stream(stream);
  RoiRows rr; size_t o_tab, o_list, o_cnt, o_box;
  const size_t need = roi_gather_carve(d, R > 0 ? R : 1, &rr, &o_tab, &o_list, &o_cnt, &o_box);
  MXDET_REQUIRE(workspace && workspace_bytes >= need, MXDET_EWORKSPACE, "roi_align_bwd_gather: workspace %zu < %zu",
                workspace_bytes, need);
  if (PW <= 14 && R > 0 && tuning(MXDET_TUNE_ROI_TABLE) == 0) {
    // segment form: per-roi records, then the gather
    static_assert(sizeof(RoiSeg) == sizeof(RoiBox), "the records live in the table form's box array");
    RoiSeg* seg = (RoiSeg*)((char*)workspace + o_box);
    RoiSegGrid gr;
    memset(&gr, 0, sizeof(gr));
    int blocks = 0;
    const int rows_big = (int)tuning(MXDET_TUNE_ROI_ROWS);
    for (int l = 0; l < d.num_levels; ++l) {
      const int segs = ceil_div(d.W[l], kSegW);
      gr.chunks[l] = ceil_div(segs, kSegWaves);
      gr.segs_per_wg[l] = ceil_div(segs, gr.chunks[l]);
      // finer maps get multi-row tiles (the roi list is built once per tile); coarse maps -- few pixels, many rois per
      // pixel -- keep one row per tile for parallelism; more rois than one list round: one row (sums stay in registers)
      gr.rows_per_wg[l] = (R > kSegChunk || d.H[l] < 64 || rows_big < 1) ? 1 : rows_big;
      gr.bands[l] = ceil_div(d.H[l], gr.rows_per_wg[l]);
      gr.block0[l] = blocks;
      blocks += d.N * gr.bands[l] * gr.chunks[l];
    }
    gr.block0[d.num_levels] = blocks;
    if (mode != 2)
      hipLaunchKernelGGL(roi_bwd_box_kernel, dim3((unsigned)ceil_div((int)R, 256)), dim3(256), 0, s, d, rois, levels, (int)R,
                         PH, PW, sampling_ratio, seg);
    if (mode == 1) return check_launch("roi_align_bwd_gather_prepare");
    const dim3 grid((unsigned)blocks, (unsigned)ceil_div(C, 256));
#define MXDET_ROI_SEG(PWT, SR)                                                                                          \
  hipLaunchKernelGGL((roi_bwd_seg_kernel<PWT, SR>), grid, dim3(kSegWaves * 64), 0, s, d, gr, C, (const RoiSeg*)seg, (int)R, \
                     PH, PW, sampling_ratio, grad_out, accumulate)
    if (PW <= 7) { if (sampling_ratio == 2) MXDET_ROI_SEG(7, 2); else MXDET_ROI_SEG(7, 0); }
    else { if (sampling_ratio == 2) MXDET_ROI_SEG(14, 2); else MXDET_ROI_SEG(14, 0); }
#undef MXDET_ROI_SEG
    return check_launch("roi_align_bwd_gather");
  }
  if (mode == 1) return MXDET_OK;          // the table form builds its tables in the gather call
  RoiTab* tab = (RoiTab*)((char*)workspace + o_tab);
  unsigned short* list = (unsigned short*)((char*)workspace + o_list);
  int* cnt = (int*)((char*)workspace + o_cnt);
  RoiBox* box = (RoiBox*)((char*)workspace + o_box);
  if (R > 0)
    hipLaunchKernelGGL(roi_bwd_tab_kernel, dim3((unsigned)R), dim3(64), 0, s, d, rois, levels, (long long)R, PH, PW,
                       sampling_ratio, tab, box);
  hipLaunchKernelGGL(roi_bwd_rows_kernel, dim3(rr.rows_total), dim3(64), 0, s, d, rr, (const RoiBox*)box,
                     (int)R, list, cnt);
  hipLaunchKernelGGL(roi_align_bwd_gather_kernel, dim3((unsigned)rr.block0[d.num_levels]), dim3(64), 0, s, d, rr, C,
                     (const RoiTab*)tab, (const RoiBox*)box, (int)R, (const unsigned short*)list, (const int*)cnt, PH, PW,
                     sampling_ratio, grad_out, accumulate);
  return check_launch("roi_align_bwd_gather");
}

extern "C" int mxdet_roi_align_bwd_gather(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C, const float* rois,
                                          const int32_t* levels, int64_t R, int32_t PH, int32_t PW,
                                          int32_t sampling_ratio, const uint16_t* grad_out, int32_t accumulate,
                                          void* workspace, size_t workspace_bytes, mxdet_stream_t stream) {
  return roi_bwd_gather_impl(f, N, C, rois, levels, R, PH, PW, sampling_ratio, grad_out, accumulate, workspace,
                             workspace_bytes, stream, 0);
}

extern "C" int mxdet_roi_align_bwd_gather_prepare(const mxdet_feat_pyramid_t* f, int32_t N, const float* rois,
                                                  const int32_t* levels, int64_t R, int32_t PH, int32_t PW,
                                                  int32_t sampling_ratio, void* workspace, size_t workspace_bytes,
                                                  mxdet_stream_t stream) {
  return roi_bwd_gather_impl(f, N, 4, rois, levels, R, PH, PW, sampling_ratio, nullptr, 0, workspace, workspace_bytes,
                             stream, 1);
}

extern "C" int mxdet_roi_align_bwd_gather_prepared(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C, const float* rois,
                                                   const int32_t* levels, int64_t R, int32_t PH, int32_t PW,
                                                   int32_t sampling_ratio, const uint16_t* grad_out, int32_t accumulate,
                                                   void* workspace, size_t workspace_bytes, mxdet_stream_t stream) {
  return roi_bwd_gather_impl(f, N, C, rois, levels, R, PH, PW, sampling_ratio, grad_out, accumulate, workspace,
                             workspace_bytes, stream, 2);
}
