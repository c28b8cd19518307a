// wgrad.hip -- bf16 weight-gradient convolution on MFMA for gfx950 (deterministic split-K).
//
// Slots: backbones / necks / rpn_heads / bbox_heads / mask_heads (/root/reference/README.md:27-31);
// MXNet role: Convolution / FullyConnected backward-weights (README.md:37).
//
//   dW[co][kh][kw][ci] = sum over output pixels m of dY[m][co] * X[src(m,kh,kw)][ci]
//
// GEMM view per tap: rows = co, cols = ci, reduction = pixels. Both operands are channels-last, i.e.
// the reduction index is the *strided* one, so the MFMA fragments (8 consecutive k per lane) are
// columns of the LDS image: they are fetched with ds_read_b64_tr_b16 (the CDNA4 transposing LDS
// read), two per fragment. LDS images are [pixel][128 ch] with the 32-B granule index XOR-swizzled
// by (row&3)|((row>>3)&1)<<2, which makes every transposed read conflict-free.
// The pixel range is split over `ksplit` workgroups per tile; each writes an fp32 slab and a second kernel adds
// the slabs in index order (bit-reproducible, no float atomics) and optionally accumulates into dw (filters
// shared across pyramid levels: RPN head); with one split the tile goes straight to dw. The bias gradient
// (column sums of dy) rides along as co_tiles*ksplit extra workgroups appended to the same grid (a ones-vector
// MFMA inside the main loop was measured first: +16 accumulators took the kernel from 120 to 176 registers and
// halved its occupancy) and is folded over the splits by the same second kernel.
// (A "last workgroup of the tile folds the slabs" epilogue was measured and dropped: the device-scope fence it
// needs is buffer_wbl2 + buffer_inv of the whole XCD L2 per workgroup, which destroys the tap-sharing L2 reuse
// -- 2.3 -> 8.9 ms per step.)
#include <stdlib.h>

#include "common.h"
#include "wgrad_tile.h"
#include "wgrad3_tile.h"

namespace mxdet {

template <int BKP, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
wgrad_kernel(WgradP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * 2 * BKP * 256];
  if (wgrad_plain(p)) wgrad_tile<BKP, NS, true>(p, (int)blockIdx.x, smem);
  else wgrad_tile<BKP, NS, false>(p, (int)blockIdx.x, smem);
}

// ---- grouped form: the weight gradients of MANY layers in one launch ----------------------------------------------
// A ResNet stage at batch 2 is 10-20 layers whose own grids (one or two workgroups per CU, 17-step K loops, a slab
// fold each) leave the chip latency-bound; issued as one grid of a few thousand workgroups they run at the
// occupancy the big P2-level layers get, need 3-4x less split-K (slab traffic) and two launches instead of 40.
// The table lives in device memory (built once per group by mxdet_conv2d_wgrad_grouped_plan, pointers are static
// under hipGraph replay); slab addresses are offsets from the workspace passed at launch.

template <int BKP, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NS == 2 ? 4 : (NS == 3 ? 3 : 2), 4)))
wgrad_grouped_kernel(const WgradG* __restrict__ table, int n, unsigned char* __restrict__ workspace) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * 2 * BKP * 256];
  const int bid = (int)blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (table[mid].block0 <= bid) lo = mid; else hi = mid - 1;
  }
  const int b = bid - table[lo].block0;
  if (b >= table[lo].nblocks) return;              // alignment padding between layers
  WgradP p = table[lo].p;
  p.slab = (float*)(workspace + (size_t)p.slab);
  p.bslab = (float*)(workspace + (size_t)p.bslab);
  if (wgrad_plain(p)) wgrad_tile<BKP, NS, true>(p, b, smem);
  else wgrad_tile<BKP, NS, false>(p, b, smem);
}

// the three-tap tiles (wgrad3_tile.h): 256 threads, NS * 25 KiB of LDS, two workgroups per CU
template <int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad3_kernel(WgradP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * kT3Stage];
  wgrad3_tile<NS>(p, (int)blockIdx.x, smem);
}

template <int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad3_grouped_kernel(const WgradG* __restrict__ table, int n, unsigned char* __restrict__ workspace) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * kT3Stage];
  const int bid = (int)blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (table[mid].bblock0 <= bid) lo = mid; else hi = mid - 1;
  }
  const int b = bid - table[lo].bblock0;
  if (b >= table[lo].bnblocks) return;             // alignment padding between layers
  WgradP p = table[lo].p;
  p.slab = (float*)(workspace + (size_t)p.slab);
  wgrad3_tile<NS>(p, b, smem);
}

// Both tile kinds of a group in ONE grid. The 3x3 tiles are MFMA-bound, the 1x1 / strided tiles mostly HBM-bound (a 1x1
// layer reads x and dy once for Cin*Cout/(Cin+Cout) flop per byte): as two launches in a row each phase leaves the other
// resource idle (measured: the three-tap kernel at 1.24 PFLOP/s in the step, the one-tap kernel behind it at 0.4), mixed
// in one grid a CU holds one workgroup of each kind. Groups of 8 workgroups (one per XCD) alternate between the two lists
// in proportion to their lengths (n3g : n1g groups), so that both lists run out together.
template <int NS3, int NS1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_mixed_grouped_kernel(const WgradG* __restrict__ table, int n, unsigned char* __restrict__ workspace, int n3g, int n1g) {
  constexpr int kLds = NS3 * kT3Stage > NS1 * 2 * kWgradBKP * 256 ? NS3 * kT3Stage : NS1 * 2 * kWgradBKP * 256;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kLds];
  const int G = (int)blockIdx.x >> 3, x8 = (int)blockIdx.x & 7, T = n3g + n1g;
  const int a = (int)(((long long)G * n3g) / T), a2 = (int)(((long long)(G + 1) * n3g) / T);
  if (a2 > a) {                                       // the a-th group of three-tap tiles
    const int bid = a * 8 + x8;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (table[mid].bblock0 <= bid) lo = mid; else hi = mid - 1;
    }
    const int b = bid - table[lo].bblock0;
    if (b >= table[lo].bnblocks) return;             // alignment padding between layers
    WgradP p = table[lo].p;
    p.slab = (float*)(workspace + (size_t)p.slab);
    wgrad3_tile<NS3>(p, b, smem);
  } else {                                            // the (G - a)-th group of one-tap tiles / bias workgroups
    const int bid = (G - a) * 8 + x8;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (table[mid].block0 <= bid) lo = mid; else hi = mid - 1;
    }
    const int b = bid - table[lo].block0;
    if (b >= table[lo].nblocks) return;
    WgradP p = table[lo].p;
    p.slab = (float*)(workspace + (size_t)p.slab);
    p.bslab = (float*)(workspace + (size_t)p.bslab);
    if (wgrad_plain(p)) wgrad_tile<kWgradBKP, NS1, true>(p, b, smem);
    else wgrad_tile<kWgradBKP, NS1, false>(p, b, smem);
  }
}

// dw[i] (+)= sum_ks slab[ks][i] in index order; the trailing workgroups fold the bias partials the same way.
// Four independent 16-B loads in flight per thread (the slabs are read exactly once: latency, not bandwidth, bounds
// a small grid).
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab, int ksplit, long long n,
                    int Cout, int wblocks, int accumulate, float* __restrict__ dw, float* __restrict__ db) {
  if ((int)blockIdx.x >= wblocks) {
    int c = ((int)blockIdx.x - wblocks) * 256 + threadIdx.x;
    if (c >= Cout) return;
    float s = bslab[c];
    for (int k = 1; k < ksplit; ++k) s += bslab[(size_t)k * Cout + c];
    db[c] = accumulate ? db[c] + s : s;
    return;
  }
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float* sp = slab + i;
  float4 s = *(const float4*)sp;
  int k = 1;
  for (; k + 3 < ksplit; k += 4) {
    float4 v0 = *(const float4*)(sp + (long long)k * n);
    float4 v1 = *(const float4*)(sp + (long long)(k + 1) * n);
    float4 v2 = *(const float4*)(sp + (long long)(k + 2) * n);
    float4 v3 = *(const float4*)(sp + (long long)(k + 3) * n);
    s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
    s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
    s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
    s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
  }
  for (; k < ksplit; ++k) {
    float4 v = *(const float4*)(sp + (long long)k * n);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (accumulate) {
    float4 o = *(const float4*)(dw + i);
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
  }
  *(float4*)(dw + i) = s;
}

// fold kernel of the grouped form: one launch for every layer of the group
__global__ void __launch_bounds__(256)
wgrad_reduce_grouped_kernel(const WgradG* __restrict__ table, int n, const unsigned char* __restrict__ workspace) {
  int lo = 0, hi = n - 1;
  const int bid = (int)blockIdx.x;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (table[mid].rblock0 <= bid) lo = mid; else hi = mid - 1;
  }
  const WgradG& g = table[lo];
  const int b = bid - g.rblock0;
  const int ksplit = g.fold_ksplit, Cout = g.p.Cout, accumulate = g.p.accumulate;
  if (b >= g.wblocks + g.bblocks) return;      // nothing to fold (single split: dw / db written directly; or not the owner)
  const float* slab = (const float*)(workspace + (size_t)g.p.slab);
  const float* bslab = (const float*)(workspace + (size_t)g.p.bslab);
  float* dw = g.p.dw;
  float* db = g.p.db;
  const long long nn = g.nparams;
  if (b >= g.wblocks) {
    int c = (b - g.wblocks) * 256 + threadIdx.x;
    if (c >= Cout) return;
    float s = bslab[c];
    for (int k = 1; k < ksplit; ++k) s += bslab[(size_t)k * Cout + c];
    db[c] = accumulate ? db[c] + s : s;
    return;
  }
  long long i = ((long long)b * blockDim.x + threadIdx.x) * 4;
  if (i >= nn) return;
  const float* sp = slab + i;
  float4 s = *(const float4*)sp;
  int k = 1;
  for (; k + 3 < ksplit; k += 4) {
    float4 v0 = *(const float4*)(sp + (long long)k * nn);
    float4 v1 = *(const float4*)(sp + (long long)(k + 1) * nn);
    float4 v2 = *(const float4*)(sp + (long long)(k + 2) * nn);
    float4 v3 = *(const float4*)(sp + (long long)(k + 3) * nn);
    s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
    s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
    s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
    s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
  }
  for (; k < ksplit; ++k) {
    float4 v = *(const float4*)(sp + (long long)k * nn);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (accumulate) {
    float4 o = *(const float4*)(dw + i);
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
  }
  *(float4*)(dw + i) = s;
}

// w [Cout][taps][Cin] -> wt [Cin][taps][Cout]
__global__ void filter_transpose_kernel(const uint16_t* __restrict__ w, int Cout, int taps, int Cin,
                                        uint16_t* __restrict__ wt) {
  __shared__ uint16_t tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int co = co0 + r, ci = ci0 + threadIdx.x;
    uint16_t v = 0;
    if (co < Cout && ci < Cin) v = w[((size_t)co * taps + tap) * Cin + ci];
    tile[r][threadIdx.x] = v;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int ci = ci0 + r, co = co0 + threadIdx.x;
    if (co < Cout && ci < Cin) wt[((size_t)ci * taps + tap) * Cout + co] = tile[threadIdx.x][r];
  }
}

// batched form: one launch for every filter of the model (descriptor table in device memory)
struct TransposeDesc { const uint16_t* w; uint16_t* wt; int Cout, taps, Cin, tile0; };
__global__ void __launch_bounds__(256)
filter_transpose_batched_kernel(const TransposeDesc* __restrict__ descs, int ndesc) {
  // 64 (co) x 64 (ci) tile per workgroup: 16-byte global loads along ci, 16-byte global stores along co. LDS pitch 66
  // halfwords: the column gathers of the store phase (lanes = 8 ci x 8 co-groups) fall on 32 distinct banks, two lanes
  // per bank reading the same dword.
  __shared__ uint32_t tile32[64 * 33];
  uint16_t* tile = (uint16_t*)tile32;
  int lo = 0, hi = ndesc - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const TransposeDesc d = descs[lo];
  int t = b - d.tile0;
  const int tiles_ci = (d.Cin + 63) >> 6, tiles_co = (d.Cout + 63) >> 6;
  const int tci = t % tiles_ci; t /= tiles_ci;
  const int tco = t % tiles_co; t /= tiles_co;
  const int tap = t;
  const int ci0 = tci * 64, co0 = tco * 64;
  const bool vec = ((d.Cin | d.Cout) & 7) == 0 && ((((uintptr_t)d.w) | ((uintptr_t)d.wt)) & 15) == 0;
  const int tid = threadIdx.x;
#pragma unroll
  for (int c = tid; c < 512; c += 256) {
    const int r = c >> 3, k = (c & 7) * 8;
    const int co = co0 + r, ci = ci0 + k;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (co < d.Cout && ci < d.Cin) {
      const uint16_t* src = d.w + ((size_t)co * d.taps + tap) * d.Cin + ci;
      if (vec) {
        const uint4 q = *(const uint4*)src;
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
        for (int j = 0; j < 8; ++j)
          if (ci + j < d.Cin) v[j >> 1] |= (uint32_t)src[j] << (16 * (j & 1));
      }
    }
    uint32_t* dst = tile32 + r * 33 + (k >> 1);
    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
  }
  __syncthreads();
#pragma unroll
  for (int c = tid; c < 512; c += 256) {
    const int r = c >> 3, k = (c & 7) * 8;        // r: ci within the tile, k: first of 8 co
    const int ci = ci0 + r, co = co0 + k;
    if (ci >= d.Cin || co >= d.Cout) continue;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v[j] = (uint32_t)tile[(k + 2 * j) * 66 + r] | ((uint32_t)tile[(k + 2 * j + 1) * 66 + r] << 16);
    uint16_t* dst = d.wt + ((size_t)ci * d.taps + tap) * d.Cout + co;
    if (vec) {
      *(uint4*)dst = make_uint4(v[0], v[1], v[2], v[3]);
    } else {
      for (int j = 0; j < 8; ++j)
        if (co + j < d.Cout) dst[j] = (uint16_t)(v[j >> 1] >> (16 * (j & 1)));
    }
  }
}

struct WgradPlan {
  int co_tiles, ci_tiles, taps, ksplit, steps_per_split;
  int t3;                    // three-tap tiles (wgrad3_tile.h): ksplit / steps_per_split count 32-pixel steps all the same
  int t3_ci_tiles, t3_steps;
  size_t slab_bytes, bslab_off, bslab_bytes, total;
};

static thread_local int g_force_ksplit = 0;   // tuning hook (mxdet_debug_force_wgrad_ksplit), 0 = heuristic

static WgradPlan plan_wgrad(const mxdet_conv_desc_t* d) {
  WgradPlan w;
  w.co_tiles = ceil_div(d->Cout, 128);
  w.ci_tiles = ceil_div(d->Cin, 128);
  w.taps = d->KH * d->KW;
  long long M = (long long)d->N * d->Ho * d->Wo;
  int steps = (int)ceil_div<long long>(M, kWgradBKP);
  int tiles = w.co_tiles * w.ci_tiles * w.taps;
  // aim for ~4 workgroups per CU overall, at least 8 steps per split
  // 1024 workgroups are resident at once (4 per CU: 32 KiB of LDS, <= 128 registers); one more would run alone in
  // a second round, so round the split DOWN
  int want = 1024 / tiles > 0 ? 1024 / tiles : 1;
  const int min_steps = 512 / kWgradBKP;   // at least 512 pixels per split
  int maxsplit = steps / min_steps > 0 ? steps / min_steps : 1;
  int ks = want < maxsplit ? want : maxsplit;
  ks = ks < 1 ? 1 : (ks > 64 ? 64 : ks);
  if (g_force_ksplit > 0) ks = g_force_ksplit < steps ? g_force_ksplit : steps;
  w.steps_per_split = ceil_div(steps, ks);
  w.ksplit = ceil_div(steps, w.steps_per_split);
  w.t3 = tuning(MXDET_TUNE_T3_ENABLE) != 0 && wgrad3_eligible(d->KH, d->KW, d->stride, d->pad, d->H, d->W);
  w.t3_ci_tiles = ceil_div(d->Cin, 64);
  w.t3_steps = 0;
  if (w.t3) {
    // 512 workgroups are resident at once (two per CU); one full round if the layer has the pixels for it
    const int steps3 = (int)ceil_div<long long>(wgrad3_vpixels(d->N, d->H, d->W), kT3Px);
    const int tiles3 = w.co_tiles * w.t3_ci_tiles * 3;
    int want3 = 512 / tiles3 > 0 ? 512 / tiles3 : 1;
    const int max3 = steps3 / 8 > 0 ? steps3 / 8 : 1;          // at least 512 pixels per split
    int ks3 = want3 < max3 ? want3 : max3;
    ks3 = ks3 > 64 ? 64 : ks3;
    if (g_force_ksplit > 0) ks3 = g_force_ksplit < steps3 ? g_force_ksplit : steps3;
    w.t3_steps = ceil_div(steps3, ks3);
    w.ksplit = ceil_div(steps3, w.t3_steps);
    w.steps_per_split = ceil_div(steps, w.ksplit);              // the bias workgroups: any partition into ksplit ranges
  }
  size_t params = (size_t)d->Cout * w.taps * d->Cin;
  w.slab_bytes = align_up((size_t)w.ksplit * params * sizeof(float), 256);
  w.bslab_off = w.slab_bytes;
  w.bslab_bytes = align_up((size_t)w.ksplit * d->Cout * sizeof(float), 256);
  w.total = w.slab_bytes + w.bslab_bytes;
  return w;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" int mxdet_debug_force_wgrad_ksplit(int32_t ks) {
  g_force_ksplit = ks;
  return MXDET_OK;
}

extern "C" size_t mxdet_conv2d_wgrad_workspace_bytes(const mxdet_conv_desc_t* d) {
  if (!d || d->N <= 0 || d->Cout <= 0 || d->Cin <= 0 || d->KH <= 0 || d->KW <= 0 || d->Ho <= 0 || d->Wo <= 0)
    return 0;
  return plan_wgrad(d).total;
}

extern "C" int mxdet_conv2d_wgrad(const mxdet_conv_desc_t* d, const uint16_t* x, const uint16_t* dy,
                                  float* dw, float* db, void* workspace,
                                  size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(d != nullptr, MXDET_EINVAL, "conv2d_wgrad: null descriptor");
  MXDET_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 &&
                    d->stride > 0 && d->pad >= 0,
                MXDET_ESHAPE, "conv2d_wgrad: non-positive dimension");
  MXDET_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 &&
                    d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                MXDET_ESHAPE, "conv2d_wgrad: Ho/Wo do not match the convolution arithmetic");
  MXDET_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0, MXDET_ESHAPE,
                "conv2d_wgrad: Cin and Cout must be multiples of 8");
  MXDET_REQUIRE((long long)d->N * d->H * d->W * d->Cin < (1ll << 31) &&
                    (long long)d->N * d->Ho * d->Wo * d->Cout < (1ll << 31),
                MXDET_ESHAPE, "conv2d_wgrad: tensor exceeds 2^31 elements");
  MXDET_REQUIRE(x && dy && dw, MXDET_EINVAL, "conv2d_wgrad: null pointer");
  WgradPlan w = plan_wgrad(d);
  MXDET_REQUIRE(workspace && workspace_bytes >= w.total, MXDET_EWORKSPACE,
                "conv2d_wgrad: workspace %zu < %zu", workspace_bytes, w.total);
  MXDET_REQUIRE(d->Cin % 4 == 0, MXDET_ESHAPE, "conv2d_wgrad: Cin must be a multiple of 4");
  hipStream_t s = as_stream(stream);
  const long long tiles = (long long)w.co_tiles * w.ci_tiles * w.taps;
  WgradP p;
  p.x = x; p.dy = dy; p.slab = (float*)workspace;
  p.bslab = (float*)((char*)workspace + w.bslab_off);
  p.dw = dw; p.db = db; p.accumulate = d->accumulate; p.force_slab = 0;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
  p.M = d->N * d->Ho * d->Wo;
  p.co_tiles = w.co_tiles; p.ci_tiles = w.ci_tiles; p.ksplit = w.ksplit;
  p.steps_per_split = w.steps_per_split;
  long long nwg = tiles * w.ksplit;
  p.t3_nwg = 0; p.t3_ci_tiles = w.t3_ci_tiles; p.t3_ksplit = w.ksplit; p.t3_steps = w.t3_steps;
  if (w.t3) {
    // three-tap tiles first; the one-tap kernel keeps only the bias workgroups (same pixel ranges)
    p.t3_nwg = w.co_tiles * w.t3_ci_tiles * 3 * w.ksplit;
    p.nwg_main = 0;
    if (tuning(MXDET_TUNE_T3_NS) == 3)
      hipLaunchKernelGGL((wgrad3_kernel<3>), dim3((unsigned)p.t3_nwg), dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((wgrad3_kernel<2>), dim3((unsigned)p.t3_nwg), dim3(256), 0, s, p);
    nwg = 0;
  }
  p.nwg_main = (int)nwg;
  if (db) nwg += (long long)w.co_tiles * w.ksplit;
  // few workgroups per CU: a deeper ring hides the load latency that co-resident workgroups would otherwise hide
  if (nwg > 0 && nwg <= 2 * 256)
    hipLaunchKernelGGL((wgrad_kernel<kWgradBKP, 4>), dim3((unsigned)nwg), dim3(256), 0, s, p);
  else if (nwg > 0)
    hipLaunchKernelGGL((wgrad_kernel<kWgradBKP, 2>), dim3((unsigned)nwg), dim3(256), 0, s, p);
  if (w.ksplit > 1) {
    long long params = (long long)d->Cout * w.taps * d->Cin;
    int wblocks = (int)ceil_div<long long>(params / 4, 256);
    int bblocks = db ? ceil_div(d->Cout, 256) : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(wblocks + bblocks)), dim3(256), 0, s,
                       (const float*)p.slab, (const float*)p.bslab, w.ksplit, params, d->Cout, wblocks,
                       d->accumulate, dw, db);
  }
  return check_launch("conv2d_wgrad");
}

// ---- grouped weight gradients -----------------------------------------------------------------------------------------
static int validate_wgrad_desc(const mxdet_conv_desc_t* d, const char* who) {
  MXDET_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 &&
                    d->stride > 0 && d->pad >= 0,
                MXDET_ESHAPE, "%s: non-positive dimension", who);
  MXDET_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 &&
                    d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                MXDET_ESHAPE, "%s: Ho/Wo do not match the convolution arithmetic", who);
  MXDET_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0, MXDET_ESHAPE, "%s: Cin and Cout must be multiples of 8", who);
  MXDET_REQUIRE((long long)d->N * d->H * d->W * d->Cin < (1ll << 31) &&
                    (long long)d->N * d->Ho * d->Wo * d->Cout < (1ll << 31),
                MXDET_ESHAPE, "%s: tensor exceeds 2^31 elements", who);
  return MXDET_OK;
}

extern "C" size_t mxdet_conv2d_wgrad_grouped_table_bytes(int32_t n) {
  return n > 0 ? (size_t)n * sizeof(WgradG) : 0;
}

extern "C" int mxdet_conv2d_wgrad_grouped_plan(const mxdet_wgrad_item_t* items, int32_t n, void* table_host,
                                               size_t table_bytes, size_t* workspace_bytes, int32_t* grid_wgrad,
                                               int32_t* grid_big, int32_t* grid_reduce) {
  clear_error();
  MXDET_REQUIRE(items && n > 0 && table_host && workspace_bytes && grid_wgrad && grid_big && grid_reduce, MXDET_EINVAL,
                "wgrad_grouped_plan: null pointer or empty group");
  MXDET_REQUIRE(table_bytes >= (size_t)n * sizeof(WgradG), MXDET_EWORKSPACE, "wgrad_grouped_plan: table too small");
  WgradG* t = (WgradG*)table_host;
  size_t off = 0;
  long long blocks = 0, rblocks = 0;
  // pixels per workgroup: long enough K loops to amortise the 64-KiB slab a workgroup writes, short enough that the
  // group has a few thousand workgroups; never below 16 steps
  long long tiles_total = 0;
  // 3x3 / stride 1 / pad 1 items go to the three-tap kernel (wgrad3_tile.h). All of them get the same step count S per
  // workgroup (64-pixel steps), the smallest S >= the minimum for which the group has at most the target number of
  // workgroups: equal-length workgroups pack the rounds of the 512 resident ones evenly, whatever the map size.
  const bool use_t3 = tuning(MXDET_TUNE_T3_ENABLE) != 0;
  auto is_big = [&](const mxdet_conv_desc_t& d) {
    return use_t3 && wgrad3_eligible(d.KH, d.KW, d.stride, d.pad, d.H, d.W);
  };
  int big_S = 0;
  {
    // workgroups aimed for: the tuned total for a large group (the two groups of the single-GPU step hold 9-10 such layers),
    // proportionally fewer for a small one (the per-stage groups of the exchange schedule hold 2-5: split as deep as the
    // large groups, they paid more in slabs and folds than the extra workgroups gave back)
    int n_t3 = 0;
    for (int i = 0; i < n; ++i) n_t3 += is_big(items[i].desc) ? 1 : 0;
    long long target = tuning(MXDET_TUNE_T3_TARGET);
    const long long per_item = tuning(MXDET_TUNE_T3_PER_ITEM);
    if (per_item > 0 && per_item * n_t3 < target) target = per_item * n_t3;
    const int smin = (int)tuning(MXDET_TUNE_T3_MINSTEPS);
    auto count = [&](int S) {
      long long c = 0;
      for (int i = 0; i < n; ++i) {
        const mxdet_conv_desc_t& d = items[i].desc;
        if (!is_big(d)) continue;
        const long long steps = ceil_div<long long>(wgrad3_vpixels(d.N, d.H, d.W), kT3Px);
        long long ks = ceil_div<long long>(steps, S);
        ks = ks > 64 ? 64 : ks;
        c += (long long)ceil_div(d.Cout, 128) * ceil_div(d.Cin, 64) * 3 * ks;
      }
      return c;
    };
    int lo = smin < 1 ? 1 : smin, hi = 1 << 16;
    if (count(lo) <= target) hi = lo;
    while (lo < hi) {                         // count() is non-increasing in S
      const int mid = (lo + hi) >> 1;
      if (count(mid) <= target) hi = mid; else lo = mid + 1;
    }
    big_S = lo;
  }
  long long bblocks_total = 0;
  for (int i = 0; i < n; ++i)
    if (!is_big(items[i].desc))
      tiles_total += (long long)ceil_div(items[i].desc.Cout, 128) * ceil_div(items[i].desc.Cin, 128) *
                     items[i].desc.KH * items[i].desc.KW;
  for (int i = 0; i < n; ++i) {
    const mxdet_conv_desc_t* d = &items[i].desc;
    int rc = validate_wgrad_desc(d, "wgrad_grouped_plan");
    if (rc) return rc;
    MXDET_REQUIRE(items[i].x && items[i].dy && items[i].dw, MXDET_EINVAL, "wgrad_grouped_plan: item %d: null pointer", i);
    WgradG& g = t[i];
    memset(&g, 0, sizeof(g));
    WgradP& p = g.p;
    p.x = (const uint16_t*)items[i].x; p.dy = (const uint16_t*)items[i].dy;
    p.dw = items[i].dw; p.db = items[i].db; p.accumulate = d->accumulate;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
    p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
    p.M = d->N * d->Ho * d->Wo;
    p.co_tiles = ceil_div(d->Cout, 128); p.ci_tiles = ceil_div(d->Cin, 128);
    const int taps = d->KH * d->KW;
    if (is_big(*d)) {
      const int bsteps = (int)ceil_div<long long>(wgrad3_vpixels(d->N, d->H, d->W), kT3Px);   // virtual pixels (pad per row)
      int ks = ceil_div(bsteps, big_S);
      ks = ks > 64 ? 64 : ks;
      p.t3_steps = ceil_div(bsteps, ks);
      p.t3_ksplit = ceil_div(bsteps, p.t3_steps);
      p.t3_ci_tiles = ceil_div(d->Cin, 64);
      p.t3_nwg = p.co_tiles * p.t3_ci_tiles * 3 * p.t3_ksplit;
      // the bias partial sums stay with the one-tap kernel, over the same pixel ranges
      p.ksplit = p.t3_ksplit;
      p.steps_per_split = ceil_div(ceil_div(p.M, kWgradBKP), p.ksplit);   // any partition of the pixels into ksplit ranges
      p.nwg_main = 0;
      g.nparams = (long long)d->Cout * taps * d->Cin;
      g.nblocks = p.db ? p.co_tiles * p.ksplit : 0;
      g.block0 = (int)blocks;
      blocks += align_up((size_t)g.nblocks, 8);
      g.bnblocks = p.t3_nwg;
      g.bblock0 = (int)bblocks_total;
      bblocks_total += align_up((size_t)g.bnblocks, 8);
      MXDET_REQUIRE(blocks < (1ll << 30) && bblocks_total < (1ll << 30), MXDET_ESHAPE, "wgrad_grouped_plan: group too large");
      continue;
    }
    g.bblock0 = (int)bblocks_total;      // no three-tap tiles: an empty range keeps the table sorted for the search
    g.bnblocks = 0;
    const long long tiles = (long long)p.co_tiles * p.ci_tiles * taps;
    const int steps = ceil_div(p.M, kWgradBKP);
    // aim for ~3000 workgroups over the group (3 rounds of the 1024 resident ones)
    // measured sweep (profiles/r01_h_grouped_wgrad_sweep.txt): ~3072 workgroups per group, 64..128 steps each
    // (mxdet_debug_set_tuning overrides for whole-step sweeps, e.g. shorter workgroups for the fused backward launch)
    const long long target = tuning(MXDET_TUNE_WG_TARGET);
    const int min_steps = (int)tuning(MXDET_TUNE_WG_MINSTEPS) * 32 / kWgradBKP;      // the tunings count 32-pixel steps
    const int max_steps = (int)tuning(MXDET_TUNE_WG_MAXSTEPS) * 32 / kWgradBKP;
    long long want = tiles_total > 0 ? (target + tiles_total - 1) / tiles_total : 1;
    int ks = (int)want;
    const int max_ks = steps / min_steps > 0 ? steps / min_steps : 1, min_ks = ceil_div(steps, max_steps);
    ks = ks > max_ks ? max_ks : ks;
    ks = ks < min_ks ? min_ks : ks;
    ks = ks < 1 ? 1 : (ks > 64 ? 64 : ks);
    p.steps_per_split = ceil_div(steps, ks);
    p.ksplit = ceil_div(steps, p.steps_per_split);
    const size_t params = (size_t)d->Cout * taps * d->Cin;
    g.nparams = (long long)params;
    p.nwg_main = (int)(tiles * p.ksplit);
    g.nblocks = p.nwg_main + (p.db ? p.co_tiles * p.ksplit : 0);
    g.block0 = (int)blocks;
    blocks += align_up((size_t)g.nblocks, 8);
    MXDET_REQUIRE(blocks < (1ll << 30), MXDET_ESHAPE, "wgrad_grouped_plan: group too large");
  }
  // slabs: items that share dw (one filter applied at several pyramid levels) write into one run of slabs that the
  // first of them (the owner) folds; everything else owns its run
  for (int i = 0; i < n; ++i) {
    int owner = i;
    for (int j = 0; j < i; ++j)
      if (items[j].dw == items[i].dw) { owner = j; break; }
    if (owner != i) continue;
    int total_ks = 0, members = 0;
    for (int j = i; j < n; ++j)
      if (items[j].dw == items[i].dw) { total_ks += t[j].p.ksplit; ++members; }
    const size_t params = (size_t)t[i].nparams, cout = (size_t)t[i].p.Cout;
    const size_t slab0 = off;
    off += align_up((size_t)total_ks * params * sizeof(float), 256);
    const size_t bslab0 = off;
    off += align_up((size_t)total_ks * cout * sizeof(float), 256);
    int ks0 = 0;
    for (int j = i; j < n; ++j) {
      if (items[j].dw != items[i].dw) continue;
      MXDET_REQUIRE((size_t)t[j].nparams == params && items[j].db == items[i].db, MXDET_ESHAPE,
                    "wgrad_grouped_plan: items %d and %d share dw but differ in shape or db", i, j);
      t[j].p.slab = (float*)(slab0 + (size_t)ks0 * params * sizeof(float));
      t[j].p.bslab = (float*)(bslab0 + (size_t)ks0 * cout * sizeof(float));
      t[j].p.force_slab = members > 1 ? 1 : 0;
      ks0 += t[j].p.ksplit;
    }
    WgradG& g = t[i];
    g.fold_ksplit = total_ks;
    const bool fold = members > 1 || g.p.ksplit > 1;
    g.wblocks = fold ? (int)ceil_div<long long>((long long)params / 4, 256) : 0;
    g.bblocks = (fold && g.p.db) ? ceil_div((int)cout, 256) : 0;
    // the owner's fold reads from the start of the run
    g.rblock0 = 0;
  }
  for (int i = 0; i < n; ++i) {
    t[i].rblock0 = (int)rblocks;
    rblocks += t[i].wblocks + t[i].bblocks;
    MXDET_REQUIRE(rblocks < (1ll << 30), MXDET_ESHAPE, "wgrad_grouped_plan: group too large");
  }
  *workspace_bytes = off;
  *grid_wgrad = (int32_t)blocks;
  *grid_big = (int32_t)bblocks_total;
  *grid_reduce = (int32_t)rblocks;
  return MXDET_OK;
}

extern "C" int mxdet_conv2d_wgrad_grouped(const void* table_dev, int32_t n, int32_t grid_wgrad, int32_t grid_big,
                                          int32_t grid_reduce, void* workspace, size_t workspace_bytes,
                                          size_t workspace_needed, mxdet_stream_t stream) {
  return mxdet_conv2d_wgrad_grouped_parts(table_dev, n, grid_wgrad, grid_big, grid_reduce, 7, workspace, workspace_bytes,
                                          workspace_needed, stream);
}

// parts: bit 0 = the three-tap kernel, bit 1 = the one-tap kernel (+ bias workgroups), bit 2 = the fold. The two tile
// kernels are independent of each other (a caller may issue them on two streams: the 3x3 tiles are MFMA-bound, the 1x1
// tiles HBM-bound); the fold needs both.
extern "C" int mxdet_conv2d_wgrad_grouped_parts(const void* table_dev, int32_t n, int32_t grid_wgrad, int32_t grid_big,
                                                int32_t grid_reduce, int32_t parts, void* workspace, size_t workspace_bytes,
                                                size_t workspace_needed, mxdet_stream_t stream) {
  clear_error();
  if (!(parts & 1)) grid_big = 0;
  if (!(parts & 4)) grid_reduce = 0;
  if (!(parts & 2)) {
    if (grid_big == 0 && grid_reduce == 0) return MXDET_OK;
    grid_wgrad = 0;
  } else if (grid_wgrad == 0 && grid_big == 0 && grid_reduce == 0) {
    return MXDET_OK;
  }
  MXDET_REQUIRE(table_dev && n > 0 && grid_wgrad >= 0 && grid_big >= 0 && grid_wgrad + grid_big + grid_reduce > 0, MXDET_EINVAL,
                "wgrad_grouped: empty group");
  MXDET_REQUIRE(workspace_needed == 0 || (workspace && workspace_bytes >= workspace_needed), MXDET_EWORKSPACE,
                "wgrad_grouped: workspace %zu < %zu", workspace_bytes, workspace_needed);
  hipStream_t s = as_stream(stream);
  if (grid_big > 0 && grid_wgrad > 0 && tuning(MXDET_TUNE_T3_MIX) != 0 && (grid_big & 7) == 0 && (grid_wgrad & 7) == 0) {
    const int n3g = grid_big >> 3, n1g = grid_wgrad >> 3;
    if (tuning(MXDET_TUNE_T3_MIX) == 2)
      hipLaunchKernelGGL((wgrad_mixed_grouped_kernel<2, 2>), dim3((unsigned)(grid_big + grid_wgrad)), dim3(256), 0, s,
                         (const WgradG*)table_dev, n, (unsigned char*)workspace, n3g, n1g);
    else
      hipLaunchKernelGGL((wgrad_mixed_grouped_kernel<2, 3>), dim3((unsigned)(grid_big + grid_wgrad)), dim3(256), 0, s,
                         (const WgradG*)table_dev, n, (unsigned char*)workspace, n3g, n1g);
    grid_big = 0;
    grid_wgrad = 0;
  }
  if (grid_big > 0) {
    if (tuning(MXDET_TUNE_T3_NS) == 3)
      hipLaunchKernelGGL((wgrad3_grouped_kernel<3>), dim3((unsigned)grid_big), dim3(256), 0, s, (const WgradG*)table_dev, n,
                         (unsigned char*)workspace);
    else
      hipLaunchKernelGGL((wgrad3_grouped_kernel<2>), dim3((unsigned)grid_big), dim3(256), 0, s, (const WgradG*)table_dev, n,
                         (unsigned char*)workspace);
  }
  if (grid_wgrad > 0) {
    // ring depth: 2 stages (32 KiB, 4 workgroups per CU), 3 (48 KiB, 3 per CU) or 4 (64 KiB, 2 per CU)
    switch ((int)tuning(MXDET_TUNE_WG_NS)) {
      case 3:
        hipLaunchKernelGGL((wgrad_grouped_kernel<kWgradBKP, 3>), dim3((unsigned)grid_wgrad), dim3(256), 0, s,
                           (const WgradG*)table_dev, n, (unsigned char*)workspace);
        break;
      case 4:
        hipLaunchKernelGGL((wgrad_grouped_kernel<kWgradBKP, 4>), dim3((unsigned)grid_wgrad), dim3(256), 0, s,
                           (const WgradG*)table_dev, n, (unsigned char*)workspace);
        break;
      default:
        hipLaunchKernelGGL((wgrad_grouped_kernel<kWgradBKP, 2>), dim3((unsigned)grid_wgrad), dim3(256), 0, s,
                           (const WgradG*)table_dev, n, (unsigned char*)workspace);
    }
  }
  if (grid_reduce > 0)
    hipLaunchKernelGGL(wgrad_reduce_grouped_kernel, dim3((unsigned)grid_reduce), dim3(256), 0, s,
                       (const WgradG*)table_dev, n, (const unsigned char*)workspace);
  return check_launch("conv2d_wgrad_grouped");
}

extern "C" int mxdet_filter_transpose_batched(const void* descs_dev, int32_t ndesc, int32_t total_tiles,
                                              mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(ndesc > 0 && total_tiles > 0, MXDET_ESHAPE, "filter_transpose_batched: empty table");
  MXDET_REQUIRE(descs_dev != nullptr, MXDET_EINVAL, "filter_transpose_batched: null pointer");
  hipLaunchKernelGGL(filter_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream),
                     (const TransposeDesc*)descs_dev, ndesc);
  return check_launch("filter_transpose_batched");
}

extern "C" int mxdet_filter_transpose(const uint16_t* w, int32_t Cout, int32_t KH, int32_t KW,
                                      int32_t Cin, uint16_t* wt, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(Cout > 0 && KH > 0 && KW > 0 && Cin > 0, MXDET_ESHAPE, "filter_transpose: bad shape");
  MXDET_REQUIRE(w && wt, MXDET_EINVAL, "filter_transpose: null pointer");
  dim3 grid(ceil_div(Cin, 32), ceil_div(Cout, 32), KH * KW);
  hipLaunchKernelGGL(filter_transpose_kernel, grid, dim3(32, 8), 0, as_stream(stream), w, Cout, KH * KW,
                     Cin, wt);
  return check_launch("filter_transpose");
}
