// dense_misc.hip -- stem convolution, pooling, pyramid resampling, elementwise helpers and the
// optimizer step for gfx950.
//
// Slots: backbones (stem, max-pool; /root/reference/README.md:27), necks (P6 subsample, top-down
// adjoint; README.md:31) and the kvstore/optimizer role of MXNet (README.md:37). All HBM-bound except
// the stem, which is a small MFMA GEMM (K = 147 padded to 160).
#include "common.h"

namespace mxdet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

// ---------------------------------------------------------------------------------------------
// Stem: 7x7 / stride 2 / pad 3, 3 -> 64, reading the NCHW image directly (plane rows are contiguous,
// so the patch load is coalesced along W), writing channels-last bf16. One workgroup = 128 output
// pixels of one output row. The 7 x 261 x 3 input patch is staged in LDS as bf16; A fragments are
// gathered from it with the (kh,kw,c) -> patch offset table, B (filters, K padded 147 -> 160) is read
// with ds_read_b128. 4 waves x (32 px x 64 co) x 5 k-steps of v_mfma_f32_16x16x32_bf16.
constexpr int STEM_PW = 264;                  // patch row pitch (261 used)
constexpr int STEM_PLANE = 7 * STEM_PW;       // per input channel
constexpr int STEM_K = 160;

__global__ void __launch_bounds__(256)
stem_conv_kernel(const void* __restrict__ img, int dtype, int N, int H, int W, int Ho, int Wo,
                 const uint16_t* __restrict__ w, const float* __restrict__ bias,
                 uint16_t* __restrict__ y) {
  // one LDS block: [patch | filters] during the K loop, re-used as the fp32 epilogue stage afterwards
  constexpr int PATCH_ELEMS = (3 * STEM_PLANE + 8 + 7) & ~7;                     // +8: zero slot
  constexpr int MAIN_BYTES = (PATCH_ELEMS + 64 * STEM_K) * 2;
  constexpr int EP_BYTES = 4 * 32 * 68 * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[MAIN_BYTES > EP_BYTES ? MAIN_BYTES : EP_BYTES];
  __shared__ int koff[STEM_K];
  uint16_t* patch = (uint16_t*)smem_raw;
  uint16_t* wl = patch + PATCH_ELEMS;
  float* ep = (float*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wtiles = ceil_div(Wo, 128);
  int b = blockIdx.x;
  const int wt = b % wtiles; b /= wtiles;
  const int ho = b % Ho;
  const int n = b / Ho;
  const int wo0 = wt * 128;
  // k -> patch offset table; padded k points at the zero slot
  for (int k = tid; k < STEM_K; k += 256) {
    int off = 3 * STEM_PLANE;
    if (k < 147) {
      int kh = k / 21, rem = k - kh * 21, kw = rem / 3, c = rem - kw * 3;
      off = c * STEM_PLANE + kh * STEM_PW + kw;
    }
    koff[k] = off;
  }
  if (tid < 8) patch[3 * STEM_PLANE + tid] = 0;
  // filters [64][147] -> LDS [64][160], zero padded
  for (int i = tid; i < 64 * STEM_K; i += 256) {
    int co = i / STEM_K, k = i - co * STEM_K;
    wl[i] = (k < 147) ? w[co * 147 + k] : (uint16_t)0;
  }
  // input patch: rows 2*ho-3 .. +6, cols 2*wo0-3 .. +260
  const int hi0 = 2 * ho - 3, wi0 = 2 * wo0 - 3;
  for (int i = tid; i < 3 * 7 * STEM_PW; i += 256) {
    int c = i / STEM_PLANE, rem = i - c * STEM_PLANE, r = rem / STEM_PW, col = rem - r * STEM_PW;
    int hi = hi0 + r, wi = wi0 + col;
    float v = 0.0f;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W && col < 261)
      v = load_as_f32(img, ((long long)(n * 3 + c) * H + hi) * W + wi, dtype);
    patch[i] = f32_to_bf16_bits(v);
  }
  __syncthreads();

  f32x4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 5; ++kk) {
    bf16x8_t af[2], bfr[4];
    int ko[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ko[j] = koff[kk * 32 + fq * 8 + j];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int px = wid * 32 + i * 16 + frow;   // local output pixel
      s16x8_t v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int o = ko[j];
        o = (o == 3 * STEM_PLANE) ? o : o + 2 * px;
        v[j] = (short)patch[o];
      }
      af[i] = __builtin_bit_cast(bf16x8_t, v);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bfr[j] = *(const bf16x8_t*)(wl + (j * 16 + frow) * STEM_K + kk * 32 + fq * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  }
  // epilogue through wave-private LDS: rows = 32 pixels, 64 channels
  __syncthreads();   // every wave is done reading patch / filters before the stage overwrites them
  float* e = ep + wid * 32 * 68;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) e[(i * 16 + fq * 4 + r) * 68 + j * 16 + frow] = acc[i][j][r];
  const int rl = lane >> 3, cg = lane & 7;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    int row = ps * 8 + rl;
    int wo = wo0 + wid * 32 + row;
    float4 v0 = *(const float4*)(e + row * 68 + cg * 8);
    float4 v1 = *(const float4*)(e + row * 68 + cg * 8 + 4);
    if (wo < Wo) {
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t = v[k] + (bias ? bias[cg * 8 + k] : 0.0f);
        v[k] = t > 0.0f ? t : 0.0f;
      }
      uint4 o;
      o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
      o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
      o.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
      o.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
      *(uint4*)(y + (((long long)n * Ho + ho) * Wo + wo) * 64 + cg * 8) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack8f(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8f(const float* v) {
  uint4 o;
  o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
  o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
  o.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
  o.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
  return o;
}

__global__ void maxpool3x3s2_kernel(const uint16_t* __restrict__ x, int N, int H, int W, int C, int Ho,
                                    int Wo, uint16_t* __restrict__ y) {
  const int CG = C >> 3;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)N * Ho * Wo * CG;
  if (idx >= total) return;
  int cg = (int)(idx % CG);
  long long pix = idx / CG;
  int wo = (int)(pix % Wo);
  long long t = pix / Wo;
  int ho = (int)(t % Ho), n = (int)(t / Ho);
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = -3.0e38f;
  for (int kh = 0; kh < 3; ++kh) {
    int hi = 2 * ho + kh - 1;
    if (hi < 0 || hi >= H) continue;
    for (int kw = 0; kw < 3; ++kw) {
      int wi = 2 * wo + kw - 1;
      if (wi < 0 || wi >= W) continue;
      uint4 v = *(const uint4*)(x + (((long long)n * H + hi) * W + wi) * C + cg * 8);
      float f[8];
      unpack8f(v, f);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = f[k] > m[k] ? f[k] : m[k];
    }
  }
  *(uint4*)(y + pix * C + cg * 8) = pack8f(m);
}

__global__ void subsample2_kernel(const uint16_t* __restrict__ x, int N, int H, int W, int C, int Ho,
                                  int Wo, uint16_t* __restrict__ y) {
  const int CG = C >> 3;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)N * Ho * Wo * CG;
  if (idx >= total) return;
  int cg = (int)(idx % CG);
  long long pix = idx / CG;
  int wo = (int)(pix % Wo);
  long long t = pix / Wo;
  int ho = (int)(t % Ho), n = (int)(t / Ho);
  *(uint4*)(y + pix * C + cg * 8) = *(const uint4*)(x + (((long long)n * H + 2 * ho) * W + 2 * wo) * C + cg * 8);
}

// dcoarse[n,hc,wc,:] (+)= sum over the (up to) 2x2 fine cells that read it in the nearest upsample
__global__ void upsample2_bwd_kernel(const uint16_t* __restrict__ dfine, int N, int Hf, int Wf, int C,
                                     int Hc, int Wc, int accumulate, uint16_t* __restrict__ dcoarse) {
  const int CG = C >> 3;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)N * Hc * Wc * CG;
  if (idx >= total) return;
  int cg = (int)(idx % CG);
  long long pix = idx / CG;
  int wc = (int)(pix % Wc);
  long long t = pix / Wc;
  int hc = (int)(t % Hc), n = (int)(t / Hc);
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.0f;
  if (accumulate) unpack8f(*(const uint4*)(dcoarse + pix * C + cg * 8), s);
  for (int dh = 0; dh < 2; ++dh) {
    int hf = 2 * hc + dh;
    if (hf >= Hf) continue;
    for (int dw = 0; dw < 2; ++dw) {
      int wf = 2 * wc + dw;
      if (wf >= Wf) continue;
      float f[8];
      unpack8f(*(const uint4*)(dfine + (((long long)n * Hf + hf) * Wf + wf) * C + cg * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += f[k];
    }
  }
  *(uint4*)(dcoarse + pix * C + cg * 8) = pack8f(s);
}

// adjoint of subsample2: dx = 0 except dx[n,2i,2j,:] = dy[n,i,j,:]  (+)=
__global__ void subsample2_bwd_kernel(const uint16_t* __restrict__ dy, int N, int H, int W, int C, int Ho,
                                      int Wo, int accumulate, uint16_t* __restrict__ dx) {
  const int CG = C >> 3;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)N * H * W * CG;
  if (idx >= total) return;
  int cg = (int)(idx % CG);
  long long pix = idx / CG;
  int wi = (int)(pix % W);
  long long t = pix / W;
  int hi = (int)(t % H), n = (int)(t / H);
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.0f;
  if (accumulate) unpack8f(*(const uint4*)(dx + pix * C + cg * 8), s);
  if (!(hi & 1) && !(wi & 1) && (hi >> 1) < Ho && (wi >> 1) < Wo) {
    float f[8];
    unpack8f(*(const uint4*)(dy + (((long long)n * Ho + (hi >> 1)) * Wo + (wi >> 1)) * C + cg * 8), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += f[k];
  }
  *(uint4*)(dx + pix * C + cg * 8) = pack8f(s);
}

__global__ void add_bf16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, long long n8,
                                uint4* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float fa[8], fb[8];
  unpack8f(a[i], fa);
  unpack8f(b[i], fb);
#pragma unroll
  for (int k = 0; k < 8; ++k) fa[k] += fb[k];
  out[i] = pack8f(fa);
}

__global__ void relu_bwd_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y, long long n8,
                                uint4* __restrict__ dx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float g[8], a[8];
  unpack8f(dy[i], g);
  unpack8f(y[i], a);
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = a[k] > 0.0f ? g[k] : 0.0f;
  dx[i] = pack8f(g);
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, long long n, int accumulate,
                                   uint16_t* __restrict__ y) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  if (i + 8 <= n) {
    float4 a = *(const float4*)(x + i), b = *(const float4*)(x + i + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (accumulate) {
      float o[8];
      unpack8f(*(const uint4*)(y + i), o);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += o[k];
    }
    *(uint4*)(y + i) = pack8f(v);
  } else {
    for (long long j = i; j < n; ++j) {
      float v = x[j];
      if (accumulate) v += bf16_bits_to_f32(y[j]);
      y[j] = f32_to_bf16_bits(v);
    }
  }
}

// NCHW (f32|bf16) -> NHWC bf16 through a 32x32 LDS tile per (n, h)
__global__ void nchw_to_nhwc_kernel(const void* __restrict__ x, int dtype, int N, int C, int H, int W,
                                    uint16_t* __restrict__ y) {
  __shared__ uint16_t tile[32][33];
  const int nh = blockIdx.z, n = nh / H, h = nh - n * H;
  const int w0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int c = c0 + r, w = w0 + threadIdx.x;
    uint16_t v = 0;
    if (c < C && w < W) v = f32_to_bf16_bits(load_as_f32(x, (((long long)n * C + c) * H + h) * W + w, dtype));
    tile[r][threadIdx.x] = v;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int w = w0 + r, c = c0 + threadIdx.x;
    if (c < C && w < W) y[(((long long)n * H + h) * W + w) * C + c] = tile[threadIdx.x][r];
  }
}
__global__ void nhwc_to_nchw_kernel(const uint16_t* __restrict__ x, int N, int C, int H, int W,
                                    float* __restrict__ y) {
  __shared__ float tile[32][33];
  const int nh = blockIdx.z, n = nh / H, h = nh - n * H;
  const int w0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int w = w0 + r, c = c0 + threadIdx.x;
    float v = 0.f;
    if (c < C && w < W) v = bf16_bits_to_f32(x[(((long long)n * H + h) * W + w) * C + c]);
    tile[r][threadIdx.x] = v;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int c = c0 + r, w = w0 + threadIdx.x;
    if (c < C && w < W) y[(((long long)n * C + c) * H + h) * W + w] = tile[threadIdx.x][r];
  }
}

// SGD with momentum over flat arenas; refreshes the bf16 working copy in the same pass
__global__ void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                           uint16_t* __restrict__ wb, long long n, float lr, float mom, float wd,
                           float rescale, const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;      // scheduled learning rate: read at run time so that a captured graph follows it
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 wv = *(float4*)(w + i), gv = *(const float4*)(g + i), mv = *(float4*)(m + i);
    float ww[4] = {wv.x, wv.y, wv.z, wv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gr = gg[k] * rescale + wd * ww[k];
      mm[k] = mom * mm[k] + gr;
      ww[k] = ww[k] - lr * mm[k];
    }
    *(float4*)(w + i) = make_float4(ww[0], ww[1], ww[2], ww[3]);
    *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    if (wb) {
      uint2 o;
      o.x = (unsigned)f32_to_bf16_bits(ww[0]) | ((unsigned)f32_to_bf16_bits(ww[1]) << 16);
      o.y = (unsigned)f32_to_bf16_bits(ww[2]) | ((unsigned)f32_to_bf16_bits(ww[3]) << 16);
      *(uint2*)(wb + i) = o;
    }
  } else {
    for (long long j = i; j < n; ++j) {
      float gr = g[j] * rescale + wd * w[j];
      float mv = mom * m[j] + gr;
      m[j] = mv;
      w[j] = w[j] - lr * mv;
      if (wb) wb[j] = f32_to_bf16_bits(w[j]);
    }
  }
}

static inline unsigned blocks_for(long long n, int per) { return (unsigned)ceil_div<long long>(n, per); }

}  // namespace mxdet

using namespace mxdet;

extern "C" int mxdet_stem_conv7x7(const void* image, int32_t dtype, int32_t N, int32_t H, int32_t W,
                                  const uint16_t* w, const float* bias, uint16_t* y,
                                  mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && H > 0 && W > 0, MXDET_ESHAPE, "stem_conv7x7: bad shape");
  MXDET_REQUIRE(dtype == MXDET_DTYPE_F32 || dtype == MXDET_DTYPE_BF16, MXDET_EINVAL, "stem_conv7x7: dtype");
  MXDET_REQUIRE(image && w && y, MXDET_EINVAL, "stem_conv7x7: null pointer");
  int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  long long nwg = (long long)N * Ho * ceil_div(Wo, 128);
  hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)nwg), dim3(256), 0, as_stream(stream), image, dtype,
                     N, H, W, Ho, Wo, w, bias, y);
  return check_launch("stem_conv7x7");
}

extern "C" int mxdet_maxpool3x3s2(const uint16_t* x, int32_t N, int32_t H, int32_t W, int32_t C,
                                  uint16_t* y, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MXDET_ESHAPE, "maxpool3x3s2: bad shape");
  MXDET_REQUIRE(x && y, MXDET_EINVAL, "maxpool3x3s2: null pointer");
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  long long total = (long long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, as_stream(stream), x,
                     N, H, W, C, Ho, Wo, y);
  return check_launch("maxpool3x3s2");
}

extern "C" int mxdet_subsample2(const uint16_t* x, int32_t N, int32_t H, int32_t W, int32_t C,
                                uint16_t* y, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MXDET_ESHAPE, "subsample2: bad shape");
  MXDET_REQUIRE(x && y, MXDET_EINVAL, "subsample2: null pointer");
  int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  long long total = (long long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(subsample2_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, as_stream(stream), x, N,
                     H, W, C, Ho, Wo, y);
  return check_launch("subsample2");
}

extern "C" int mxdet_subsample2_bwd(const uint16_t* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                                    int32_t accumulate, uint16_t* dx, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MXDET_ESHAPE, "subsample2_bwd: bad shape");
  MXDET_REQUIRE(dy && dx, MXDET_EINVAL, "subsample2_bwd: null pointer");
  int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  long long total = (long long)N * H * W * (C / 8);
  hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, as_stream(stream),
                     dy, N, H, W, C, Ho, Wo, accumulate, dx);
  return check_launch("subsample2_bwd");
}

extern "C" int mxdet_upsample2_bwd(const uint16_t* dfine, int32_t N, int32_t Hf, int32_t Wf, int32_t C,
                                   int32_t accumulate, uint16_t* dcoarse, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && Hf > 0 && Wf > 0 && C > 0 && C % 8 == 0, MXDET_ESHAPE, "upsample2_bwd: bad shape");
  MXDET_REQUIRE(dfine && dcoarse, MXDET_EINVAL, "upsample2_bwd: null pointer");
  int Hc = (Hf + 1) / 2, Wc = (Wf + 1) / 2;
  long long total = (long long)N * Hc * Wc * (C / 8);
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, as_stream(stream),
                     dfine, N, Hf, Wf, C, Hc, Wc, accumulate, dcoarse);
  return check_launch("upsample2_bwd");
}

extern "C" int mxdet_add_bf16(const uint16_t* a, const uint16_t* b, int64_t n, uint16_t* out,
                              mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n >= 0 && n % 8 == 0, MXDET_ESHAPE, "add_bf16: n must be a multiple of 8");
  if (n == 0) return MXDET_OK;
  MXDET_REQUIRE(a && b && out, MXDET_EINVAL, "add_bf16: null pointer");
  hipLaunchKernelGGL(add_bf16_kernel, dim3(blocks_for(n / 8, 256)), dim3(256), 0, as_stream(stream),
                     (const uint4*)a, (const uint4*)b, (long long)(n / 8), (uint4*)out);
  return check_launch("add_bf16");
}

extern "C" int mxdet_relu_bwd_bf16(const uint16_t* dy, const uint16_t* y, int64_t n, uint16_t* dx,
                                   mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n >= 0 && n % 8 == 0, MXDET_ESHAPE, "relu_bwd_bf16: n must be a multiple of 8");
  if (n == 0) return MXDET_OK;
  MXDET_REQUIRE(dy && y && dx, MXDET_EINVAL, "relu_bwd_bf16: null pointer");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for(n / 8, 256)), dim3(256), 0, as_stream(stream),
                     (const uint4*)dy, (const uint4*)y, (long long)(n / 8), (uint4*)dx);
  return check_launch("relu_bwd_bf16");
}

extern "C" int mxdet_f32_to_bf16(const float* x, int64_t n, uint16_t* y, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n >= 0, MXDET_ESHAPE, "f32_to_bf16: negative size");
  if (n == 0) return MXDET_OK;
  MXDET_REQUIRE(x && y, MXDET_EINVAL, "f32_to_bf16: null pointer");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks_for(ceil_div<long long>(n, 8), 256)), dim3(256), 0,
                     as_stream(stream), x, (long long)n, 0, y);
  return check_launch("f32_to_bf16");
}

extern "C" int mxdet_f32_accum_to_bf16(const float* x, int64_t n, int32_t accumulate, uint16_t* y,
                                       mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n >= 0, MXDET_ESHAPE, "f32_accum_to_bf16: negative size");
  if (n == 0) return MXDET_OK;
  MXDET_REQUIRE(x && y, MXDET_EINVAL, "f32_accum_to_bf16: null pointer");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks_for(ceil_div<long long>(n, 8), 256)), dim3(256), 0,
                     as_stream(stream), x, (long long)n, accumulate, y);
  return check_launch("f32_accum_to_bf16");
}

extern "C" int mxdet_nchw_to_nhwc_bf16(const void* x, int32_t dtype, int32_t N, int32_t C, int32_t H,
                                       int32_t W, uint16_t* y, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, MXDET_ESHAPE, "nchw_to_nhwc_bf16: bad shape");
  MXDET_REQUIRE((long long)N * H <= 65535, MXDET_ESHAPE, "nchw_to_nhwc_bf16: N*H exceeds the grid limit");
  MXDET_REQUIRE(x && y, MXDET_EINVAL, "nchw_to_nhwc_bf16: null pointer");
  dim3 grid(ceil_div(W, 32), ceil_div(C, 32), N * H);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(32, 8), 0, as_stream(stream), x, dtype, N, C, H, W, y);
  return check_launch("nchw_to_nhwc_bf16");
}

extern "C" int mxdet_nhwc_to_nchw_f32(const uint16_t* x, int32_t N, int32_t C, int32_t H, int32_t W,
                                      float* y, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, MXDET_ESHAPE, "nhwc_to_nchw_f32: bad shape");
  MXDET_REQUIRE((long long)N * H <= 65535, MXDET_ESHAPE, "nhwc_to_nchw_f32: N*H exceeds the grid limit");
  MXDET_REQUIRE(x && y, MXDET_EINVAL, "nhwc_to_nchw_f32: null pointer");
  dim3 grid(ceil_div(W, 32), ceil_div(C, 32), N * H);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(32, 8), 0, as_stream(stream), x, N, C, H, W, y);
  return check_launch("nhwc_to_nchw_f32");
}

extern "C" int mxdet_sgd_momentum_update(float* w, const float* grad, float* mom, uint16_t* w_bf16,
                                         int64_t n, float lr, float momentum, float wd, float rescale,
                                         mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n >= 0, MXDET_ESHAPE, "sgd_momentum_update: negative size");
  if (n == 0) return MXDET_OK;
  MXDET_REQUIRE(w && grad && mom, MXDET_EINVAL, "sgd_momentum_update: null pointer");
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks_for(ceil_div<long long>(n, 4), 256)), dim3(256), 0,
                     as_stream(stream), w, grad, mom, w_bf16, (long long)n, lr, momentum, wd, rescale, (const float*)nullptr);
  return check_launch("sgd_momentum_update");
}

extern "C" int mxdet_sgd_momentum_update_sched(float* w, const float* grad, float* mom, uint16_t* w_bf16, int64_t n,
                                               const float* lr_dev, float momentum, float wd, float rescale,
                                               mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n >= 0, MXDET_ESHAPE, "sgd_momentum_update_sched: negative size");
  if (n == 0) return MXDET_OK;
  MXDET_REQUIRE(w && grad && mom && lr_dev, MXDET_EINVAL, "sgd_momentum_update_sched: null pointer");
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks_for(ceil_div<long long>(n, 4), 256)), dim3(256), 0,
                     as_stream(stream), w, grad, mom, w_bf16, (long long)n, 0.0f, momentum, wd, rescale, lr_dev);
  return check_launch("sgd_momentum_update_sched");
}
