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
// Stem + max-pool fused: image [N,3,H,W] -> relu(conv7x7/2 + bias) -> maxpool 3x3/2 pad 1 -> bf16 [N,Hp,Wp,64], one pass.
// The unfused pair writes the 68.8 MB stem map and reads it back to produce a quarter of it; here a workgroup owns
// 2 pooled rows x 63 pooled columns: the 5 x 128 stem outputs underneath are computed row by row on the matrix pipes,
// rounded to bf16 exactly where the unfused kernel rounds (max is monotonic: the pooled bits are those of
// maxpool(stem) on the same accumulators), maxed vertically in registers and horizontally through LDS.
//   LDS patch [15 rows][296 cols][4 ch] bf16 (channel 3 = 0): a k-run of 8 = 2 columns x 4 channels is 16 contiguous
//   bytes, so an A fragment (16 stem outputs x 32 k) of filter row kh is ONE ds_read_b128 per lane at
//   ((2r + kh) * 296 + 2 * col + 2 * fq) * 8 -- always 16-B aligned (the column stride of the conv is 2), consecutive
//   chunks across lanes (conflict-free). K = 7 filter rows x (8 columns x 4 channels, column 7 and channel 3 zero).
//   Waves are 2 x 2: wave (wm, wn) owns stem columns 64wm .. 64wm+63 (4 m-tiles) x output channels 32wn .. 32wn+31
//   (2 n-tiles); its filters (2 n-tiles x 7 k-steps, 56 registers) stay in registers for the whole workgroup
//   (all 64 channels per wave took 256+ registers: one workgroup per CU).
constexpr int SP_PC = 63;                       // pooled columns per workgroup (stem columns 0..127 of the tile)
constexpr int SP_PW = 296;                      // patch columns (2*127 + 7 + padding of the last fragment)
constexpr int SP_ROWS = 15;                     // input rows under 5 stem rows
constexpr int SP_PATCH_BYTES = SP_ROWS * SP_PW * 8;
constexpr int SP_HB_BYTES = 2 * 128 * 64 * 2;   // two vertically pooled stem rows [128 cols][64 ch] bf16
constexpr int SP_LDS = SP_PATCH_BYTES > SP_HB_BYTES ? SP_PATCH_BYTES : SP_HB_BYTES;

template <typename T>
__device__ __forceinline__ float stem_px(const T* p);
template <>
__device__ __forceinline__ float stem_px<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float stem_px<uint16_t>(const uint16_t* p) { return bf16_bits_to_f32(*p); }

template <typename T>
__global__ void __launch_bounds__(256, 2)      // two waves per SIMD: at most 256 registers (the need is ~180)
stem_pool_kernel(const T* __restrict__ img, int N, int H, int W, int Ho, int Wo, int Hp, int Wp,
                 const uint16_t* __restrict__ w, const float* __restrict__ bias, uint16_t* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SP_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int wtiles = ceil_div(Wp, SP_PC), htiles = ceil_div(Hp, 2);
  int b = blockIdx.x;
  const int wt = b % wtiles; b /= wtiles;
  const int ht = b % htiles;
  const int n = b / htiles;
  const int ph0 = 2 * ht, pw0 = wt * SP_PC;
  const int cr0 = 2 * ph0 - 1, cc0 = 2 * pw0 - 1;        // first stem row / column of the tile
  const int ir0 = 2 * cr0 - 3, ic0 = 2 * cc0 - 3;        // first input row / column of the patch

  // ---- filters -> registers: lane (frow = co within the n-tile, fq) holds k = 8fq..8fq+7 of filter row kh,
  // i.e. columns 2fq, 2fq+1 x channels 0..3 (column 7 / channel 3 are zero padding)
  // (the raw [64][147] filter goes through LDS first: 112 two-byte gathers per lane straight from global memory
  // kept the address unit busy for ~14 us per workgroup)
  {
    const uint4* w4 = (const uint4*)w;                    // 64*147*2 B = 18816 B = 1176 x 16 B
    for (int i = tid; i < 1176; i += 256) ((uint4*)smem)[i] = w4[i];
  }
  __syncthreads();
  bf16x8_t bw[2][7];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint16_t* wr = (const uint16_t*)smem + (wn * 32 + j * 16 + frow) * 147;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
      s16x8_t v;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kw = 2 * fq + (e >> 2), c = e & 3;
        v[e] = (kw < 7 && c < 3) ? (short)wr[(kh * 7 + kw) * 3 + c] : (short)0;
      }
      bw[j][kh] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  float bz[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bz[j] = bias ? bias[wn * 32 + j * 16 + frow] : 0.0f;
  __syncthreads();                                        // the filter image is consumed: the patch may overwrite it

  // ---- input patch -> LDS, 4 bf16 per (row, col): plane rows are contiguous, so loads coalesce along the column.
  // Every load is unconditional (clamped coordinates, zero selected afterwards) and a whole pass of 15 rows x 3 planes
  // is issued before the first value is used: inside `if (in range)` hipcc waits for each load separately, 30 dependent
  // round trips per thread.
  {
    const size_t plane = (size_t)H * W;
    const T* base = img + (size_t)(n * 3) * plane;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int col = tid + pass * 256;
      if (pass == 1 && col >= SP_PW) break;
      const int wi = ic0 + col;
      const bool cok = wi >= 0 && wi < W;
      const int wic = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
      float v[SP_ROWS][3];
#pragma unroll
      for (int r = 0; r < SP_ROWS; ++r) {
        const int hi = ir0 + r;
        const int hic = hi < 0 ? 0 : (hi >= H ? H - 1 : hi);
        const T* p0 = base + (size_t)hic * W + wic;
        v[r][0] = stem_px<T>(p0);
        v[r][1] = stem_px<T>(p0 + plane);
        v[r][2] = stem_px<T>(p0 + 2 * plane);
      }
#pragma unroll
      for (int r = 0; r < SP_ROWS; ++r) {
        const int hi = ir0 + r;
        const bool ok = cok && hi >= 0 && hi < H;
        uint2 o;
        o.x = (unsigned)f32_to_bf16_bits(v[r][0]) | ((unsigned)f32_to_bf16_bits(v[r][1]) << 16);
        o.y = (unsigned)f32_to_bf16_bits(v[r][2]);
        if (!ok) o = make_uint2(0u, 0u);
        *(uint2*)(smem + ((size_t)r * SP_PW + col) * 8) = o;
      }
    }
  }
  __syncthreads();

  // ---- five stem rows; vertical max of rows {0,1,2} and {2,3,4} in registers. Values are >= 0 after the ReLU, so (i) an
  // out-of-range stem row / column contributes an exact 0, which never wins over the in-range pixel every window has,
  // and (ii) bf16 bit patterns order like the values: the running maxima are kept as packed bf16 pairs and updated with
  // packed 16-bit integer maxima (half the registers, a third of the instructions of the fp32 form).
  typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  u16x2_t pmax[2][4][2][2];        // [pooled row][m-tile][n-tile][r pair]
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r) pmax[q][i][j][r] = (u16x2_t){0, 0};
  // tiles that touch no border of the stem map need no per-value validity select (wave-uniform test)
  const bool interior = cr0 >= 0 && cr0 + 5 <= Ho && cc0 >= 0 && cc0 + 128 <= Wo;
#pragma unroll 1
  for (int cr = 0; cr < 5; ++cr) {
    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
      bf16x8_t af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = wm * 64 + i * 16 + frow;                     // stem column inside the tile
        af[i] = *(const bf16x8_t*)(smem + ((size_t)(2 * cr + kh) * SP_PW + 2 * col + 2 * fq) * 8);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bw[j][kh], acc[i][j], 0, 0, 0);
    }
    const bool rowok = (cr0 + cr) >= 0 && (cr0 + cr) < Ho;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // D layout: row = fq*4 + r -> stem column, col = frow -> channel
      const int ccb = cc0 + wm * 64 + i * 16 + fq * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaxf(acc[i][j][r] + bz[j], 0.0f);
          if (!interior) v[r] = (rowok && ccb + r >= 0 && ccb + r < Wo) ? v[r] : 0.0f;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // the stem map's own rounding (RNE, v_cvt_pk_bf16_f32), two values per register
          const bf16x2_t pk = (bf16x2_t){(__bf16)v[2 * h], (__bf16)v[2 * h + 1]};
          const u16x2_t u = __builtin_bit_cast(u16x2_t, pk);
          if (cr <= 2) pmax[0][i][j][h] = __builtin_elementwise_max(pmax[0][i][j][h], u);
          if (cr >= 2) pmax[1][i][j][h] = __builtin_elementwise_max(pmax[1][i][j][h], u);
        }
      }
    }
  }
  __syncthreads();                                   // every wave is done with the patch: the LDS becomes hbuf
  uint16_t* hb = (uint16_t*)smem;                    // [2][128 cols][64 ch]
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hb[(q * 128 + wm * 64 + i * 16 + fq * 4 + r) * 64 + wn * 32 + j * 16 + frow] = pmax[q][i][j][r >> 1][r & 1];
  __syncthreads();
  // ---- horizontal max: pooled column pw <- stem columns 2pw, 2pw+1, 2pw+2 of the tile; 8 channels per lane
  for (int it = tid; it < 2 * SP_PC * 8; it += 256) {
    const int cg = it & 7, pw = (it >> 3) % SP_PC, q = (it >> 3) / SP_PC;
    const int ph = ph0 + q, pwg = pw0 + pw;
    if (ph >= Hp || pwg >= Wp) continue;
    const uint4 a = *(const uint4*)(hb + (q * 128 + 2 * pw) * 64 + cg * 8);
    const uint4 c = *(const uint4*)(hb + (q * 128 + 2 * pw + 1) * 64 + cg * 8);
    const uint4 d = *(const uint4*)(hb + (q * 128 + 2 * pw + 2) * 64 + cg * 8);
    // non-negative bf16 values order like their bit patterns: the max works on the packed halves directly
    auto mx = [](unsigned x, unsigned y2) {
      const unsigned lo = (x & 0xffffu) > (y2 & 0xffffu) ? (x & 0xffffu) : (y2 & 0xffffu);
      const unsigned hi = (x >> 16) > (y2 >> 16) ? (x >> 16) : (y2 >> 16);
      return lo | (hi << 16);
    };
    uint4 o;
    o.x = mx(mx(a.x, c.x), d.x); o.y = mx(mx(a.y, c.y), d.y); o.z = mx(mx(a.z, c.z), d.z); o.w = mx(mx(a.w, c.w), d.w);
    *(uint4*)(y + (((size_t)n * Hp + ph) * Wp + pwg) * 64 + cg * 8) = o;
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

extern "C" int mxdet_stem_conv7x7_pool(const void* image, int32_t dtype, int32_t N, int32_t H, int32_t W,
                                       const uint16_t* w, const float* bias, uint16_t* y, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && H > 0 && W > 0, MXDET_ESHAPE, "stem_conv7x7_pool: bad shape");
  MXDET_REQUIRE(dtype == MXDET_DTYPE_F32 || dtype == MXDET_DTYPE_BF16, MXDET_EINVAL, "stem_conv7x7_pool: dtype");
  MXDET_REQUIRE(image && w && y, MXDET_EINVAL, "stem_conv7x7_pool: null pointer");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
  const long long nwg = (long long)N * ceil_div(Hp, 2) * ceil_div(Wp, SP_PC);
  MXDET_REQUIRE(nwg < (1ll << 31), MXDET_ESHAPE, "stem_conv7x7_pool: image too large");
  if (dtype == MXDET_DTYPE_F32)
    hipLaunchKernelGGL(stem_pool_kernel<float>, dim3((unsigned)nwg), dim3(256), 0, as_stream(stream), (const float*)image,
                       N, H, W, Ho, Wo, Hp, Wp, w, bias, y);
  else
    hipLaunchKernelGGL(stem_pool_kernel<uint16_t>, dim3((unsigned)nwg), dim3(256), 0, as_stream(stream),
                       (const uint16_t*)image, N, H, W, Ho, Wo, Hp, Wp, w, bias, y);
  return check_launch("stem_conv7x7_pool");
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
