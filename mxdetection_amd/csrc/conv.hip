// conv.hip -- bf16 implicit-GEMM convolution on MFMA for gfx950: forward and data-gradient.
//
// Slots: backbones / necks / rpn_heads / bbox_heads / mask_heads (/root/reference/README.md:27-31);
// MXNet roles Convolution(+BatchNorm(use_global_stats)+Activation+elemwise_add), FullyConnected and
// their backward-data (README.md:37). One kernel template serves both directions:
//
//   GEMM rows  m = destination pixel (n, hd, wd)        (fwd: output pixel; dgrad: input pixel)
//   GEMM cols  j = destination channel                  (fwd: Cout;         dgrad: Cin)
//   reduction  k = (kh, kw, c) with c contiguous        (fwd: c = Cin;      dgrad: c = Cout)
//
// Activations are channels-last, so an A-row's 64-channel K-slice is one 128-B line and the MFMA
// fragment (8 consecutive k per lane) is one ds_read_b128. Tiles: BM x BN x 64 per step, 4 waves,
// v_mfma_f32_16x16x32_bf16, fp32 accumulators. LDS tiles are [row][64] bf16 with the 16-B chunk index
// XOR-swizzled by (row>>1)&7 (conflict-free ds_read_b128 / ds_write_b128).
// Global->LDS is register staged and double buffered: the loads of step t+1 are issued before the
// MFMAs of step t and written to the other buffer after them (one barrier per step).
// Epilogue: accumulators -> LDS (fp32) -> rows of 8 channels per lane: + bias, + residual
// (optionally nearest-upsampled: FPN top-down), ReLU / ReLU-mask, bf16 pack, 16-B coalesced stores.
#include "common.h"

namespace mxdet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct ConvP {
  const uint16_t* x;     // gathered source [N,Hs,Ws,C]
  const uint16_t* w;     // [Ncols][KH*KW*C]
  const float* bias;     // [Ncols] or null
  const uint16_t* res;   // residual [M,Ncols] (or coarse map when res_up) or null
  const uint16_t* mask;  // ReLU mask [M,Ncols] or null (dgrad)
  uint16_t* y;           // [M,Ncols]
  int N, Hs, Ws, C;
  int Hd, Wd, Ncols;
  int KH, KW, stride, pad;
  int relu, res_up;
  int M;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
}

template <int BM, int BN, int WM, int WN, bool DGRAD>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(ConvP p) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int AI = BM / 32;                       // A chunks per thread
  constexpr int BI = (BN >= 32) ? BN / 32 : 1;      // B chunks per thread
  constexpr int STAGE = (BM + BN) * 64;             // elements per buffer
  static_assert(WM * WN == 4, "4 waves");
  static_assert(MT % 2 == 0, "epilogue stages two m-tiles at a time");
  constexpr int EP_STRIDE = WTN + 4;
  constexpr int EP_BYTES = 4 * 32 * EP_STRIDE * 4;
  constexpr int MAIN_BYTES = 2 * STAGE * 2;
  constexpr int SMEM_BYTES = MAIN_BYTES > EP_BYTES ? MAIN_BYTES : EP_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  uint16_t* smem = (uint16_t*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;

  // XCD-aware tile order: blocks that share an XCD (bid % 8) walk consecutive tiles, and consecutive
  // tiles share their A rows (n fastest), so the A tile is re-read from that XCD's L2.
  int bid = blockIdx.x;
  const int nwg = gridDim.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread gather geometry ------------------------------------------------------------
  const int chunk = tid & 7;
  const int r0 = tid >> 3;
  int a_h[AI], a_w[AI], a_base[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      int img = m / (p.Hd * p.Wd);
      int rem = m - img * (p.Hd * p.Wd);
      int hd = rem / p.Wd, wd = rem - hd * p.Wd;
      if (DGRAD) { a_h[i] = hd + p.pad; a_w[i] = wd + p.pad; }
      else { a_h[i] = hd * p.stride - p.pad; a_w[i] = wd * p.stride - p.pad; }
      a_base[i] = img * (p.Hs * p.Ws);
    } else {
      a_h[i] = -1000000; a_w[i] = -1000000; a_base[i] = 0;
    }
  }
  const int Ktot = p.KH * p.KW * p.C;
  const int cpt = p.C >> 6;            // 64-channel slices per tap
  const int KT = p.KH * p.KW * cpt;
  const uint16_t* wrow[BI];
  bool wvalid[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int n = n0 + r0 + 32 * i;
    wvalid[i] = (n < p.Ncols) && (r0 + 32 * i < BN);
    wrow[i] = p.w + (size_t)(wvalid[i] ? n : 0) * Ktot + chunk * 8;
  }

  uint4 ga[AI], gb[BI];
  auto load_tiles = [&](int kt) {
    int tap = kt / cpt;
    int c0 = (kt - tap * cpt) << 6;
    int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int hs, ws;
      bool ok;
      if (DGRAD) {
        int th = a_h[i] - kh, tw = a_w[i] - kw;
        if (p.stride == 1) { hs = th; ws = tw; ok = true; }
        else { hs = th / p.stride; ws = tw / p.stride; ok = (th == hs * p.stride) && (tw == ws * p.stride); }
        ok = ok && th >= 0 && tw >= 0 && hs < p.Hs && ws < p.Ws;
      } else {
        hs = a_h[i] + kh; ws = a_w[i] + kw;
        ok = hs >= 0 && ws >= 0 && hs < p.Hs && ws < p.Ws;
      }
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = *(const uint4*)(p.x + ((size_t)(a_base[i] + hs * p.Ws + ws) * p.C + c0 + chunk * 8));
      ga[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (wvalid[i]) v = *(const uint4*)(wrow[i] + (size_t)kt * 64);
      gb[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    uint16_t* sa = smem + buf * STAGE;
    uint16_t* sb = sa + BM * 64;
#pragma unroll
    for (int i = 0; i < AI; ++i) *(uint4*)(sa + lds_off(r0 + 32 * i, chunk)) = ga[i];
#pragma unroll
    for (int i = 0; i < BI; ++i)
      if (r0 + 32 * i < BN) *(uint4*)(sb + lds_off(r0 + 32 * i, chunk)) = gb[i];
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int frow = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tiles(kt + 1);
    const uint16_t* sa = smem + cur * STAGE;
    const uint16_t* sb = sa + BM * 64;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *(const bf16x8_t*)(sa + lds_off(wm * WTM + i * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bfr[j] = *(const bf16x8_t*)(sb + lds_off(wn * WTN + j * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  float* ep = (float*)smem_raw + wid * 32 * EP_STRIDE;
  constexpr int LPR = WTN / 8;          // lanes per row on read-back
  constexpr int RPP = 64 / LPR;         // rows per pass
  constexpr int PASSES = 32 / RPP;
  const int rl = lane / LPR, cg = lane - rl * LPR;
#pragma unroll
  for (int h = 0; h < MT / 2; ++h) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ep[(t * 16 + fq * 4 + r) * EP_STRIDE + j * 16 + frow] = acc[2 * h + t][j][r];
    // wave-private staging: the LDS ops of one wave execute in order, no barrier needed
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      int row = ps * RPP + rl;
      int m = m0 + wm * WTM + h * 32 + row;
      int col = n0 + wn * WTN + cg * 8;
      float4 v0 = *(const float4*)(ep + row * EP_STRIDE + cg * 8);
      float4 v1 = *(const float4*)(ep + row * EP_STRIDE + cg * 8 + 4);
      if (m < p.M && col < p.Ncols) {
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (p.bias) {
          float4 b0 = *(const float4*)(p.bias + col), b1 = *(const float4*)(p.bias + col + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.res) {
          size_t ri;
          if (p.res_up) {
            int img = m / (p.Hd * p.Wd);
            int rem = m - img * (p.Hd * p.Wd);
            int hd = rem / p.Wd, wd = rem - hd * p.Wd;
            int Hc = (p.Hd + 1) >> 1, Wc = (p.Wd + 1) >> 1;
            ri = ((size_t)(img * Hc + (hd >> 1)) * Wc + (wd >> 1)) * p.Ncols + col;
          } else {
            ri = (size_t)m * p.Ncols + col;
          }
          uint4 rv = *(const uint4*)(p.res + ri);
          v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
          v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
          v[4] += __uint_as_float(rv.z << 16); v[5] += __uint_as_float(rv.z & 0xffff0000u);
          v[6] += __uint_as_float(rv.w << 16); v[7] += __uint_as_float(rv.w & 0xffff0000u);
        }
        if (p.mask) {
          uint4 mv = *(const uint4*)(p.mask + (size_t)m * p.Ncols + col);
          // bf16 > 0  <=>  sign clear and magnitude non-zero
          unsigned mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            unsigned lo = mm[k] & 0xffffu, hi = mm[k] >> 16;
            if (!(lo != 0u && lo < 0x8000u)) v[2 * k] = 0.0f;
            if (!(hi != 0u && hi < 0x8000u)) v[2 * k + 1] = 0.0f;
          }
        }
        if (p.relu && !p.mask) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
        }
        uint4 o;
        o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
        o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
        o.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
        o.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
        *(uint4*)(p.y + (size_t)m * p.Ncols + col) = o;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, bool DGRAD>
static int launch_cfg(ConvP& p, hipStream_t s) {
  p.tiles_m = ceil_div(p.M, BM);
  p.tiles_n = ceil_div(p.Ncols, BN);
  long long nwg = (long long)p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, DGRAD>), dim3((unsigned)nwg), dim3(256), 0, s, p);
  return check_launch("conv2d");
}

template <bool DGRAD>
static int launch(ConvP& p, hipStream_t s) {
  if (p.Ncols <= 16) return launch_cfg<256, 16, 4, 1, DGRAD>(p, s);
  if (p.Ncols <= 64) return launch_cfg<256, 64, 4, 1, DGRAD>(p, s);
  return launch_cfg<128, 128, 2, 2, DGRAD>(p, s);
}

static int validate(const mxdet_conv_desc_t* d, const char* who) {
  MXDET_REQUIRE(d != nullptr, MXDET_EINVAL, "%s: null descriptor", who);
  MXDET_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 &&
                    d->stride > 0 && d->pad >= 0,
                MXDET_ESHAPE, "%s: non-positive dimension", who);
  MXDET_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 &&
                    d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                MXDET_ESHAPE, "%s: Ho/Wo do not match the convolution arithmetic", who);
  MXDET_REQUIRE((long long)d->N * d->H * d->W * d->Cin < (1ll << 31) &&
                    (long long)d->N * d->Ho * d->Wo * d->Cout < (1ll << 31),
                MXDET_ESHAPE, "%s: tensor exceeds 2^31 elements", who);
  return MXDET_OK;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" int mxdet_conv2d_fwd(const mxdet_conv_desc_t* d, const uint16_t* x, const uint16_t* w,
                                const float* bias, const uint16_t* residual, uint16_t* y,
                                mxdet_stream_t stream) {
  clear_error();
  int rc = validate(d, "conv2d_fwd");
  if (rc) return rc;
  MXDET_REQUIRE(d->Cin % 64 == 0, MXDET_ESHAPE, "conv2d_fwd: Cin %d must be a multiple of 64", d->Cin);
  MXDET_REQUIRE(d->Cout % 8 == 0, MXDET_ESHAPE, "conv2d_fwd: Cout %d must be a multiple of 8", d->Cout);
  MXDET_REQUIRE(x && w && y, MXDET_EINVAL, "conv2d_fwd: null pointer");
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w; p.bias = bias; p.res = residual; p.mask = nullptr; p.y = y;
  p.N = d->N; p.Hs = d->H; p.Ws = d->W; p.C = d->Cin;
  p.Hd = d->Ho; p.Wd = d->Wo; p.Ncols = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.relu = d->relu; p.res_up = d->res_upsample;
  p.M = d->N * d->Ho * d->Wo;
  return launch<false>(p, as_stream(stream));
}

extern "C" int mxdet_conv2d_dgrad(const mxdet_conv_desc_t* d, const uint16_t* dy, const uint16_t* wt,
                                  const uint16_t* residual, const uint16_t* relu_mask, uint16_t* dx,
                                  mxdet_stream_t stream) {
  clear_error();
  int rc = validate(d, "conv2d_dgrad");
  if (rc) return rc;
  MXDET_REQUIRE(d->Cout % 64 == 0, MXDET_ESHAPE, "conv2d_dgrad: Cout %d must be a multiple of 64", d->Cout);
  MXDET_REQUIRE(d->Cin % 8 == 0, MXDET_ESHAPE, "conv2d_dgrad: Cin %d must be a multiple of 8", d->Cin);
  MXDET_REQUIRE(dy && wt && dx, MXDET_EINVAL, "conv2d_dgrad: null pointer");
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.x = dy; p.w = wt; p.bias = nullptr; p.y = dx;
  p.res = residual ? residual : (d->accumulate ? dx : nullptr);
  p.mask = d->relu ? relu_mask : nullptr;
  MXDET_REQUIRE(!d->relu || relu_mask, MXDET_EINVAL, "conv2d_dgrad: relu set without relu_mask");
  p.N = d->N; p.Hs = d->Ho; p.Ws = d->Wo; p.C = d->Cout;
  p.Hd = d->H; p.Wd = d->W; p.Ncols = d->Cin;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.relu = 0; p.res_up = 0;
  p.M = d->N * d->H * d->W;
  return launch<true>(p, as_stream(stream));
}
