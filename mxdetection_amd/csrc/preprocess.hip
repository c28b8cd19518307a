// preprocess.hip -- the data-pipeline step in front of the training path (process_data, README.md:23):
// decoded 8-bit frames -> flipped / resized / normalised / padded bf16 NCHW batch, and polygon -> instance-mask
// rasterisation at network resolution. Integer (OpenCV 8-bit bilinear) arithmetic, bit-exact with oracle/mxdet_oracle.c.
#include "common.h"

namespace mxdet {

struct PreTab {
  mxdet_image_desc_t im[MXDET_PREPROCESS_MAX_BATCH];
};

struct PreNorm {
  float mean[3], stdv[3];
};

// 11-bit coefficient pair + source index of one destination coordinate (see include/mxdet.h)
__device__ __forceinline__ void resize_coef(int d, double inv, int slen, int& s0, int& s1, int& c0, int& c1) {
  float f = (float)(((double)d + 0.5) * inv - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.0f; s = 0; }
  if (s >= slen - 1) { f = 0.0f; s = slen - 1; }
  s0 = s;
  s1 = s + 1 < slen ? s + 1 : slen - 1;
  c0 = __float2int_rn((1.0f - f) * 2048.0f);
  c1 = __float2int_rn(f * 2048.0f);
}

// One lane = 8 consecutive output columns of one output row, all three planes: three 16-byte stores. A workgroup's
// lanes walk consecutive column groups, so the planar stores are fully coalesced; the source taps (<= 2 rows x ~18
// columns x 3 bytes per lane) come through L2/TCP.
__global__ void __launch_bounds__(256)
image_preprocess_kernel(PreTab tab, PreNorm nm, int N, int Hp, int Wp, int swap_rb, uint16_t* __restrict__ out) {
  const int wch = Wp >> 3;
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= (long long)N * Hp * wch) return;
  const int xc = (int)(item % wch);
  const long long r = item / wch;
  const int y = (int)(r % Hp), n = (int)(r / Hp);
  const mxdet_image_desc_t& d = tab.im[n];
  uint16_t* o = out + ((long long)n * 3 * Hp + y) * Wp + xc * 8;
  const long long plane = (long long)Hp * Wp;
  unsigned pk[3][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  if (y < d.dst_h && xc * 8 < d.dst_w) {
    int sy0, sy1, b0, b1;
    resize_coef(y, d.inv_scale, d.src_h, sy0, sy1, b0, b1);
    const uint8_t* r0 = d.src + (long long)sy0 * d.src_w * 3;
    const uint8_t* r1 = d.src + (long long)sy1 * d.src_w * 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int x = xc * 8 + j;
      if (x >= d.dst_w) continue;
      int sx0, sx1, a0, a1;
      resize_coef(x, d.inv_scale, d.src_w, sx0, sx1, a0, a1);
      if (d.flip) { sx0 = d.src_w - 1 - sx0; sx1 = d.src_w - 1 - sx1; }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int sc = swap_rb ? 2 - c : c;
        const int t0 = (int)r0[sx0 * 3 + sc] * a0 + (int)r0[sx1 * 3 + sc] * a1;
        const int t1 = (int)r1[sx0 * 3 + sc] * a0 + (int)r1[sx1 * 3 + sc] * a1;
        const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
        const float f = __fdiv_rn(__fsub_rn((float)v, nm.mean[c]), nm.stdv[c]);
        pk[c][j >> 1] |= (unsigned)f32_to_bf16_bits(f) << (16 * (j & 1));
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) *(uint4*)(o + c * plane) = make_uint4(pk[c][0], pk[c][1], pk[c][2], pk[c][3]);
}

// Row-staged form (the one launched when both source rows fit in LDS): one workgroup = one output row. The two source
// rows the row blends are copied to LDS with aligned 4-byte loads (the byte-granular global gathers of the direct form
// -- 96 one-byte loads per lane -- were what bounded it, not HBM), the taps are then LDS byte reads.
__global__ void __launch_bounds__(256)
image_preprocess_rows_kernel(PreTab tab, PreNorm nm, int N, int Hp, int Wp, int swap_rb, uint16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rows_lds[];
  const int y = blockIdx.x % Hp, n = blockIdx.x / Hp;
  const mxdet_image_desc_t& d = tab.im[n];
  const long long plane = (long long)Hp * Wp;
  uint16_t* orow = out + ((long long)n * 3 * Hp + y) * Wp;
  const int wch = Wp >> 3;
  if (y >= d.dst_h) {
    for (int i = threadIdx.x; i < 3 * wch; i += 256) {
      const int c = i / wch, xc = i - c * wch;
      *(uint4*)(orow + c * plane + xc * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  int sy0, sy1, b0, b1;
  resize_coef(y, d.inv_scale, d.src_h, sy0, sy1, b0, b1);
  const int rb = d.src_w * 3;                       // bytes per source row
  const int pitch = (rb + 3 + 3) & ~3;              // staged bytes per row: lead (<= 3) + row, rounded to dwords
  int lead[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint8_t* row = d.src + (long long)(r ? sy1 : sy0) * rb;
    lead[r] = (int)((uintptr_t)row & 3);
    const uint32_t* arow = (const uint32_t*)(row - lead[r]);
    const int nd = (lead[r] + rb + 3) >> 2;
    uint32_t* dst = (uint32_t*)(rows_lds + r * pitch);
    for (int i = threadIdx.x; i < nd; i += 256) dst[i] = arow[i];
  }
  __syncthreads();
  const uint8_t* r0 = rows_lds + lead[0];
  const uint8_t* r1 = rows_lds + pitch + lead[1];
  for (int xc = threadIdx.x; xc < wch; xc += 256) {
    unsigned pk[3][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if (xc * 8 < d.dst_w) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int x = xc * 8 + j;
        if (x >= d.dst_w) continue;
        int sx0, sx1, a0, a1;
        resize_coef(x, d.inv_scale, d.src_w, sx0, sx1, a0, a1);
        if (d.flip) { sx0 = d.src_w - 1 - sx0; sx1 = d.src_w - 1 - sx1; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int sc = swap_rb ? 2 - c : c;
          const int t0 = (int)r0[sx0 * 3 + sc] * a0 + (int)r0[sx1 * 3 + sc] * a1;
          const int t1 = (int)r1[sx0 * 3 + sc] * a0 + (int)r1[sx1 * 3 + sc] * a1;
          const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
          const float f = __fdiv_rn(__fsub_rn((float)v, nm.mean[c]), nm.stdv[c]);
          pk[c][j >> 1] |= (unsigned)f32_to_bf16_bits(f) << (16 * (j & 1));
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) *(uint4*)(orow + c * plane + xc * 8) = make_uint4(pk[c][0], pk[c][1], pk[c][2], pk[c][3]);
  }
}

// One lane = 4 consecutive pixels of one row of one instance mask (one 32-bit store). The row test of an edge is
// uniform over the workgroup's row, so rows that no edge straddles cost two compares per edge and no division.
__global__ void __launch_bounds__(256)
polygon_masks_kernel(const float2* __restrict__ verts, const int* __restrict__ poly_start,
                     const int* __restrict__ inst_first, int H, int W, uint8_t* __restrict__ masks) {
  const int inst = blockIdx.z, y = blockIdx.y;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= W) return;
  const float py = (float)y + 0.5f;
  unsigned inside = 0u;   // bit j: pixel x0+j is inside some polygon of the instance
  const int pb = inst_first[inst], pe = inst_first[inst + 1];
  for (int p = pb; p < pe; ++p) {
    const int vb = poly_start[p], ve = poly_start[p + 1];
    if (ve - vb < 3) continue;
    unsigned par = 0u;
    float2 a = verts[ve - 1];
    for (int v = vb; v < ve; ++v) {
      const float2 b = verts[v];
      if ((a.y <= py) != (b.y <= py)) {
        const float xi = __fadd_rn(a.x, __fdiv_rn(__fmul_rn(__fsub_rn(py, a.y), __fsub_rn(b.x, a.x)), __fsub_rn(b.y, a.y)));
#pragma unroll
        for (int j = 0; j < 4; ++j) par ^= ((float)(x0 + j) + 0.5f < xi) ? (1u << j) : 0u;
      }
      a = b;
    }
    inside |= par;
  }
  unsigned w = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) w |= ((inside >> j) & 1u) << (8 * j);
  *(unsigned*)(masks + ((long long)inst * H + y) * W + x0) = w;
}

}  // namespace mxdet

using namespace mxdet;

static int g_preprocess_direct = 0;   // test hook (mxdet_debug_preprocess_direct): force the direct-gather form
extern "C" int mxdet_debug_preprocess_direct(int32_t on) {
  g_preprocess_direct = on;
  return MXDET_OK;
}

extern "C" int mxdet_image_preprocess(const mxdet_image_desc_t* images, int32_t N, int32_t Hp, int32_t Wp,
                                      const float* mean3, const float* std3, int32_t swap_rb, uint16_t* out,
                                      mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(images && mean3 && std3 && out, MXDET_EINVAL, "image_preprocess: null argument");
  MXDET_REQUIRE(N >= 0 && N <= MXDET_PREPROCESS_MAX_BATCH, MXDET_ESHAPE, "image_preprocess: N=%d outside 0..%d", N,
                MXDET_PREPROCESS_MAX_BATCH);
  MXDET_REQUIRE(Hp > 0 && Wp > 0 && Wp % 8 == 0, MXDET_ESHAPE, "image_preprocess: Hp=%d Wp=%d (Wp must be a multiple of 8)",
                Hp, Wp);
  MXDET_REQUIRE(((uintptr_t)out & 15) == 0, MXDET_EINVAL, "image_preprocess: out must be 16-byte aligned");
  if (N == 0) return MXDET_OK;
  PreTab tab;
  memset(&tab, 0, sizeof(tab));
  for (int n = 0; n < N; ++n) {
    const mxdet_image_desc_t& d = images[n];
    MXDET_REQUIRE(d.src && d.src_h > 0 && d.src_w > 0, MXDET_EINVAL, "image_preprocess: image %d has no source", n);
    MXDET_REQUIRE(d.dst_h > 0 && d.dst_w > 0 && d.dst_h <= Hp && d.dst_w <= Wp, MXDET_ESHAPE,
                  "image_preprocess: image %d resized %dx%d does not fit the %dx%d batch", n, d.dst_h, d.dst_w, Hp, Wp);
    MXDET_REQUIRE(d.inv_scale > 0.0, MXDET_EINVAL, "image_preprocess: image %d inv_scale must be positive", n);
    tab.im[n] = d;
  }
  PreNorm nm;
  for (int c = 0; c < 3; ++c) {
    MXDET_REQUIRE(std3[c] != 0.0f, MXDET_EINVAL, "image_preprocess: std[%d] is zero", c);
    nm.mean[c] = mean3[c];
    nm.stdv[c] = std3[c];
  }
  int max_w = 0;
  for (int n = 0; n < N; ++n) max_w = max_w > images[n].src_w ? max_w : images[n].src_w;
  const size_t lds = 2 * (size_t)((max_w * 3 + 6) & ~3);
  if (lds <= 60 * 1024 && !g_preprocess_direct) {
    hipLaunchKernelGGL(image_preprocess_rows_kernel, dim3((unsigned)(N * Hp)), dim3(256), lds, as_stream(stream), tab, nm, N,
                       Hp, Wp, swap_rb, out);
  } else {      // very wide frames: direct global gathers
    const long long items = (long long)N * Hp * (Wp / 8);
    hipLaunchKernelGGL(image_preprocess_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, as_stream(stream), tab,
                       nm, N, Hp, Wp, swap_rb, out);
  }
  return check_launch("image_preprocess");
}

extern "C" int mxdet_polygon_masks(const float* verts, const int32_t* poly_start, const int32_t* inst_first, int32_t N,
                                   int32_t G, int32_t H, int32_t W, uint8_t* masks, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(poly_start && inst_first && masks, MXDET_EINVAL, "polygon_masks: null argument");
  MXDET_REQUIRE(N >= 0 && G >= 0 && H > 0 && W > 0 && W % 4 == 0, MXDET_ESHAPE,
                "polygon_masks: N=%d G=%d H=%d W=%d (W must be a multiple of 4)", N, G, H, W);
  MXDET_REQUIRE((long long)N * G <= 65535 && H <= 65535, MXDET_ESHAPE, "polygon_masks: N*G=%lld or H=%d exceeds the grid",
                (long long)N * G, H);
  MXDET_REQUIRE(((uintptr_t)masks & 3) == 0, MXDET_EINVAL, "polygon_masks: masks must be 4-byte aligned");
  if (N * G == 0) return MXDET_OK;
  hipLaunchKernelGGL(polygon_masks_kernel, dim3((unsigned)ceil_div(W, 1024), (unsigned)H, (unsigned)(N * G)), dim3(256), 0,
                     as_stream(stream), (const float2*)verts, poly_start, inst_first, H, W, masks);
  return check_launch("polygon_masks");
}
