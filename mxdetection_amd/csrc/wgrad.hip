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
#include "common.h"

namespace mxdet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kWgradBKP = 32;   // pixels per step (32: 32 KiB of LDS -> 4+ workgroups per CU)
struct WgradP {
  const uint16_t* x;   // [N,H,W,Cin]
  const uint16_t* dy;  // [N,Ho,Wo,Cout]
  float* slab;         // [ksplit][Cout][KH*KW*Cin]
  float* bslab;        // [ksplit][Cout] bias-gradient partials (db != null)
  float* dw;           // [Cout][KH*KW*Cin]
  float* db;           // [Cout] or null
  int accumulate;
  int force_slab;      // grouped form, filters shared by several items: always write slabs (the owner folds them all)
  int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
  int M;               // N*Ho*Wo
  int co_tiles, ci_tiles, ksplit, steps_per_split;
  int nwg_main;        // MFMA workgroups; the grid continues with co_tiles*ksplit bias workgroups when db != null
};

// 16 bytes of zeros read by out-of-range lanes, so that every global load is unconditional (a branch around a
// load makes hipcc wait for each load separately: 8 serialized round trips per step)
__device__ __attribute__((aligned(256))) unsigned int g_wgrad_zero[64];

// byte offset inside a [64 px][256 B] image of 16-B chunk c16 of pixel row `row`
__device__ __forceinline__ int wg_off(int row, int c16) {
  int f = (row & 3) | (((row >> 3) & 1) << 2);
  return row * 256 + ((((c16 >> 1) ^ f) << 5) | ((c16 & 1) << 4));
}

__device__ __forceinline__ s16x4_t tr_read(const unsigned char* lds_base, int byte_off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(lds_base + byte_off));
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One workgroup = one 128(co) x 128(ci) tile of one tap over a range of 64-pixel steps. Global -> LDS is a
// 2-stage LDS-DMA ring (global_load_lds_dwordx4, 1 KiB per wave instruction = 4 pixel rows x 256 B): the loads
// of step t+1 are in flight while step t is multiplied; one s_waitcnt vmcnt(0) + one raw s_barrier per step.
// The LDS image is lane-linear, so the granule swizzle is applied to the per-lane SOURCE chunk.
template <int BKP, int NS>
__device__ __forceinline__ void wgrad_tile(const WgradP& p, int b) {
  constexpr int GI = BKP / 16;      // DMA instructions per wave per image per stage (4 pixel rows each)
  static_assert(NS >= 2 && (NS - 2) * 2 * (BKP / 16) <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS][2][BKP * 256];  // [buf][dy|x]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  // XCD-aware order with the tap fastest: the KH*KW workgroups that share one (dy tile, shifted x tile)
  // pair sit next to each other in one XCD's queue and hit that XCD's L2 for 8 of 9 reads.
  if (b >= p.nwg_main) {
    // bias-gradient workgroups (appended to the grid, they stream dy while the MFMA workgroups compute):
    // column sums of this split's pixel range for one 128-channel co tile, fixed order
    b -= p.nwg_main;
    const int co_t = b % p.co_tiles, ks = b / p.co_tiles;
    const int c8 = tid & 15, r0 = tid >> 4;              // 16 x 16-B chunks, 16 pixel rows in flight
    const int co = co_t * 128 + c8 * 8;
    int m0 = ks * p.steps_per_split * BKP, m1 = m0 + p.steps_per_split * BKP;
    m1 = m1 > p.M ? p.M : m1;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (co < p.Cout)
      for (int m = m0 + r0; m < m1; m += 16) {
        uint4 v = *(const uint4*)(p.dy + (size_t)m * p.Cout + co);
        s[0] += __uint_as_float(v.x << 16); s[1] += __uint_as_float(v.x & 0xffff0000u);
        s[2] += __uint_as_float(v.y << 16); s[3] += __uint_as_float(v.y & 0xffff0000u);
        s[4] += __uint_as_float(v.z << 16); s[5] += __uint_as_float(v.z & 0xffff0000u);
        s[6] += __uint_as_float(v.w << 16); s[7] += __uint_as_float(v.w & 0xffff0000u);
      }
    float* red = (float*)&smem[0][0][0];                  // [16 rows][128 ch]
#pragma unroll
    for (int k = 0; k < 8; ++k) red[r0 * 128 + c8 * 8 + k] = s[k];
    __syncthreads();
    if (tid < 128 && co_t * 128 + tid < p.Cout) {
      float t = red[tid];
#pragma unroll
      for (int r = 1; r < 16; ++r) t += red[r * 128 + tid];
      const int c = co_t * 128 + tid;
      if (p.ksplit == 1 && !p.force_slab) p.db[c] = p.accumulate ? p.db[c] + t : t;
      else p.bslab[(size_t)ks * p.Cout + c] = t;
    }
    return;
  }
  {
    const int nwg = p.nwg_main;
    int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntaps = p.KH * p.KW;
  const int tap = b % ntaps; b /= ntaps;
  const int ci_t = b % p.ci_tiles; b /= p.ci_tiles;
  const int co_t = b % p.co_tiles; b /= p.co_tiles;
  const int ks = b;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = co_t * 128, ci0 = ci_t * 128;

  const int step0 = ks * p.steps_per_split;
  int nsteps = ceil_div(p.M, BKP) - step0;
  nsteps = nsteps > p.steps_per_split ? p.steps_per_split : nsteps;

  // DMA geometry: wave w, instruction i (0..3) fills pixel rows 4*(4w+i) .. +3 of both images; lane l covers
  // row (l>>4), physical 16-B slot (l&15), i.e. logical chunk ((slot>>1) ^ f(row)) * 2 + (slot & 1)
  const int lrow = lane >> 4, lslot = lane & 15;
  const uint16_t* zero = (const uint16_t*)g_wgrad_zero;
  // Each lane walks its GI pixel rows BKP pixels per step with an exact carry chain (BKP = d_img*HW + d_ho*Wo + d_wo,
  // every component below its modulus), keeps 32-bit element offsets, and selects the zero page without branches.
  const int HW = p.Ho * p.Wo;
  const int d_img = BKP / HW, d_rem = BKP - d_img * HW;
  const int d_ho = d_rem / p.Wo, d_wo = d_rem - d_ho * p.Wo;
  int c_img[GI], c_ho[GI], c_wo[GI], c_m[GI], c_offy[GI], c_chx[GI];
  bool c_yok[GI], c_xok[GI];
#pragma unroll
  for (int i = 0; i < GI; ++i) {
    int row = (wid * GI + i) * 4 + lrow;
    int f = (row & 3) | (((row >> 3) & 1) << 2);
    int chunk = ((((lslot >> 1) ^ f) << 1) | (lslot & 1)) * 8;     // first channel of this lane's 16 bytes
    int m = step0 * BKP + row;
    c_m[i] = m;
    c_img[i] = m / HW;
    int rem = m - c_img[i] * HW;
    c_ho[i] = rem / p.Wo;
    c_wo[i] = rem - c_ho[i] * p.Wo;
    c_offy[i] = m * p.Cout + co0 + chunk;
    c_chx[i] = ci0 + chunk;
    c_yok[i] = (co0 + chunk) < p.Cout;
    c_xok[i] = (ci0 + chunk) < p.Cin;
  }
  const int stepy = BKP * p.Cout;
  auto issue_stage = [&](int buf, bool live) {
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const bool mok = live && c_m[i] < p.M;
      const int hi = c_ho[i] * p.stride - p.pad + kh, wi = c_wo[i] * p.stride - p.pad + kw;
      const bool yok = mok && c_yok[i];
      const bool xok = mok && c_xok[i] && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
      const int offx = ((c_img[i] * p.H + hi) * p.W + wi) * p.Cin + c_chx[i];
      const uint16_t* ay = p.dy + (unsigned)c_offy[i];
      const uint16_t* ax = p.x + (unsigned)offx;
      const uint16_t* py = yok ? ay : zero;
      const uint16_t* px = xok ? ax : zero;
#ifndef MXDET_ABL_NOLOAD
      __builtin_amdgcn_global_load_lds((gptr_t)py, (lptr_t)(smem[buf][0] + (wid * GI + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)px, (lptr_t)(smem[buf][1] + (wid * GI + i) * 1024), 16, 0, 0);
#else
      asm volatile("" ::"v"(py), "v"(px));
#endif
      // advance this row by BKP pixels
      c_m[i] += BKP;
      c_offy[i] += stepy;
      c_wo[i] += d_wo;
      int cw = c_wo[i] >= p.Wo ? 1 : 0;
      c_wo[i] -= cw ? p.Wo : 0;
      c_ho[i] += d_ho + cw;
      int ch = c_ho[i] >= p.Ho ? 1 : 0;
      c_ho[i] -= ch ? p.Ho : 0;
      c_img[i] += d_img + ch;
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // transposed-read geometry: lane 16g + 4q + pp addresses row 8g+q (+4), channels 4pp..4pp+3 of the
  // 16-channel block; it receives channel (lane&15) of those four pixel rows.
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;

  static_assert(BKP == 32, "one 32-pixel MFMA k-step per ring stage");
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&smem[0][0][0];
  unsigned offy[4], offx[4];
  {
    const int rowa = 8 * g + q;                 // first 4 pixel rows of this lane group's k-range (rows +4: offset 1024)
    const int fa = (rowa & 3) | (((rowa >> 3) & 1) << 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      offy[i] = (unsigned)(rowa * 256 + (((wm * 4 + i) ^ fa) << 5) + pp * 8);   // 16-channel granule wm*4+i of the co tile
      offx[i] = (unsigned)(rowa * 256 + (((wn * 4 + i) ^ fa) << 5) + pp * 8);
    }
  }

  // NS-deep ring: stages st+1 .. st+NS-1 are in flight while stage st is multiplied (dummy zero-page stages past
  // the end keep the counted wait uniform)
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) issue_stage(s0, s0 < nsteps);
  int cur = 0, nxt = NS - 1;
  for (int st = 0; st < nsteps; ++st) {
    // this wave's loads of step st have landed (all but the NS-2 youngest stages) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 2 * GI) : "memory");
    __builtin_amdgcn_s_barrier();                      // ... and everyone's; everyone is done with the buffer refilled next
    asm volatile("" ::: "memory");
    issue_stage(nxt, st + NS - 1 < nsteps);
    // Fragment reads in inline asm: behind the ds_read_tr builtin hipcc cannot tell that the read does not touch
    // the ring slot an LDS-DMA is still filling and drains vmcnt(0) before the first read of every step (measured:
    // the ring then overlaps nothing). The asm reads are ordered by hand: LDS returns in issue order, the first
    // fence (lgkmcnt(4)) releases the x fragments and the first two dy fragments, the second the rest, so the last
    // four reads are still in flight under the first eight MFMAs. The fences name the registers they release.
    const unsigned sbase = smem_addr + (unsigned)cur * (2u * BKP * 256u);
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x4_t ylo[4], yhi[4], xlo[4], xhi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned ad = sbase + (unsigned)(BKP * 256) + offx[j];
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xlo[j]) : "v"(ad));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(xhi[j]) : "v"(ad));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned ad = sbase + offy[i];
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ylo[i]) : "v"(ad));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(yhi[i]) : "v"(ad));
    }
    s16x8_t bx[4], ay[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bx[j] = (s16x8_t){xlo[j][0], xlo[j][1], xlo[j][2], xlo[j][3], xhi[j][0], xhi[j][1], xhi[j][2], xhi[j][3]};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ay[i] = (s16x8_t){ylo[i][0], ylo[i][1], ylo[i][2], ylo[i][3], yhi[i][0], yhi[i][1], yhi[i][2], yhi[i][3]};
    asm volatile("s_waitcnt lgkmcnt(4)"
                 : "+v"(bx[0]), "+v"(bx[1]), "+v"(bx[2]), "+v"(bx[3]), "+v"(ay[0]), "+v"(ay[1]));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#ifndef MXDET_ABL_NOMFMA
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ay[i]),
                                                            __builtin_bit_cast(bf16x8_t, bx[j]), acc[i][j], 0, 0, 0);
#else
        asm volatile("" ::"v"(ay[i]), "v"(bx[j]));
#endif
    __builtin_amdgcn_sched_barrier(0);   // keep the first eight MFMAs above the second fence
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ay[2]), "+v"(ay[3]));
#pragma unroll
    for (int i = 2; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#ifndef MXDET_ABL_NOMFMA
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ay[i]),
                                                            __builtin_bit_cast(bf16x8_t, bx[j]), acc[i][j], 0, 0, 0);
#else
        asm volatile("" ::"v"(ay[i]), "v"(bx[j]));
#endif
    cur = (cur + 1 == NS) ? 0 : cur + 1;
    nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // D layout: col = lane&15 -> ci, row = (lane>>4)*4 + r -> co
  const size_t Ktot = (size_t)p.KH * p.KW * p.Cin;
  const bool single = p.ksplit == 1 && !p.force_slab;
  // one split: the tile goes straight to dw/db; otherwise to this split's slab
  float* out = single ? p.dw : p.slab + (size_t)ks * p.Cout * Ktot;
  const bool add_old = single && p.accumulate;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int ci = ci0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int co = co0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (co < p.Cout && ci < p.Cin) {
          size_t o = (size_t)co * Ktot + (size_t)tap * p.Cin + ci;
          out[o] = add_old ? out[o] + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

template <int BKP, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
wgrad_kernel(WgradP p) {
  wgrad_tile<BKP, NS>(p, (int)blockIdx.x);
}

// ---- grouped form: the weight gradients of MANY layers in one launch ----------------------------------------------
// A ResNet stage at batch 2 is 10-20 layers whose own grids (one or two workgroups per CU, 17-step K loops, a slab
// fold each) leave the chip latency-bound; issued as one grid of a few thousand workgroups they run at the
// occupancy the big P2-level layers get, need 3-4x less split-K (slab traffic) and two launches instead of 40.
// The table lives in device memory (built once per group by mxdet_conv2d_wgrad_grouped_plan, pointers are static
// under hipGraph replay); slab addresses are offsets from the workspace passed at launch.
struct WgradG {
  WgradP p;                 // slab / bslab hold byte offsets into the workspace
  int block0, nblocks;      // this layer's workgroups: [block0, block0 + nblocks), block0 a multiple of 8
  int rblock0, wblocks, bblocks;   // fold kernel: first workgroup, workgroups over dw, workgroups over db
  int fold_ksplit;                 // slabs the fold adds up (all items that share this item's dw write into one run)
  long long nparams;
};

template <int BKP, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
wgrad_grouped_kernel(const WgradG* __restrict__ table, int n, unsigned char* __restrict__ workspace, int block_off,
                     int total) {
  // Persistent form: the grid may be smaller than the tile list (gridDim.x a multiple of 8 keeps logical tile -> XCD),
  // each workgroup then walks tiles bid, bid + gridDim.x, ... A grid of one workgroup per CU places instantly and
  // leaves the CU's other wave slots to the short dgrad kernels of the main stream, which a many-round grid would
  // keep waiting until its last workgroup has been placed.
  for (int bid = (int)blockIdx.x + block_off; bid < total; bid += (int)gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (table[mid].block0 <= bid) lo = mid; else hi = mid - 1;
    }
    const int b = bid - table[lo].block0;
    if (b < table[lo].nblocks) {                     // else: alignment padding between layers
      WgradP p = table[lo].p;
      p.slab = (float*)(workspace + (size_t)p.slab);
      p.bslab = (float*)(workspace + (size_t)p.bslab);
      wgrad_tile<BKP, NS>(p, b);
    }
    __syncthreads();                                 // LDS ring is reused by the next tile
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
  size_t slab_bytes, bslab_off, bslab_bytes, total;
};

static int g_group_persist = 0;               // tuning hook (mxdet_debug_wgrad_group_persist), 0 = one workgroup per tile
static int g_group_chunk = 0;                 // tuning hook (mxdet_debug_wgrad_group_chunk), 0 = one launch per group
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
  size_t params = (size_t)d->Cout * w.taps * d->Cin;
  w.slab_bytes = align_up((size_t)w.ksplit * params * sizeof(float), 256);
  w.bslab_off = w.slab_bytes;
  w.bslab_bytes = align_up((size_t)w.ksplit * d->Cout * sizeof(float), 256);
  w.total = w.slab_bytes + w.bslab_bytes;
  return w;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" int mxdet_debug_wgrad_group_persist(int32_t workgroups) {
  g_group_persist = workgroups;
  return MXDET_OK;
}

extern "C" int mxdet_debug_wgrad_group_chunk(int32_t workgroups) {
  g_group_chunk = workgroups;
  return MXDET_OK;
}

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
  p.nwg_main = (int)nwg;
  if (db) nwg += (long long)w.co_tiles * w.ksplit;
  // few workgroups per CU: a deeper ring hides the load latency that co-resident workgroups would otherwise hide
  if (nwg <= 2 * 256)
    hipLaunchKernelGGL((wgrad_kernel<kWgradBKP, 4>), dim3((unsigned)nwg), dim3(256), 0, s, p);
  else
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
                                               int32_t* grid_reduce) {
  clear_error();
  MXDET_REQUIRE(items && n > 0 && table_host && workspace_bytes && grid_wgrad && grid_reduce, MXDET_EINVAL,
                "wgrad_grouped_plan: null pointer or empty group");
  MXDET_REQUIRE(table_bytes >= (size_t)n * sizeof(WgradG), MXDET_EWORKSPACE, "wgrad_grouped_plan: table too small");
  WgradG* t = (WgradG*)table_host;
  size_t off = 0;
  long long blocks = 0, rblocks = 0;
  // pixels per workgroup: long enough K loops to amortise the 64-KiB slab a workgroup writes, short enough that the
  // group has a few thousand workgroups; never below 16 steps
  long long tiles_total = 0;
  for (int i = 0; i < n; ++i)
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
    const long long tiles = (long long)p.co_tiles * p.ci_tiles * taps;
    const int steps = ceil_div(p.M, kWgradBKP);
    // aim for ~3000 workgroups over the group (3 rounds of the 1024 resident ones)
    // measured sweep (profiles/r01_h_grouped_wgrad_sweep.txt): ~3072 workgroups per group, 64..128 steps each
    const long long target = 3072;
    const int min_steps = 64;
    long long want = tiles_total > 0 ? (target + tiles_total - 1) / tiles_total : 1;
    int ks = (int)want;
    const int max_ks = steps / min_steps > 0 ? steps / min_steps : 1, min_ks = ceil_div(steps, 128);
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
  *grid_reduce = (int32_t)rblocks;
  return MXDET_OK;
}

extern "C" int mxdet_conv2d_wgrad_grouped(const void* table_dev, int32_t n, int32_t grid_wgrad, int32_t grid_reduce,
                                          void* workspace, size_t workspace_bytes, size_t workspace_needed,
                                          mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(table_dev && n > 0 && grid_wgrad > 0, MXDET_EINVAL, "wgrad_grouped: empty group");
  MXDET_REQUIRE(workspace_needed == 0 || (workspace && workspace_bytes >= workspace_needed), MXDET_EWORKSPACE,
                "wgrad_grouped: workspace %zu < %zu", workspace_bytes, workspace_needed);
  hipStream_t s = as_stream(stream);
  // Optional chunking (tuning hook): a grid far larger than one resident round keeps the dispatcher on this queue until
  // it has placed every workgroup, which starves the short dgrad kernels of the main stream; chunks of about one
  // resident round let the two queues alternate at launch granularity.
  const int chunk = g_group_chunk > 0 ? g_group_chunk : grid_wgrad;
  const int persist = g_group_persist > 0 ? (g_group_persist + 7) & ~7 : 0;
  for (int off = 0; off < grid_wgrad; off += chunk) {
    const int cnt = grid_wgrad - off < chunk ? grid_wgrad - off : chunk;
    const int grid = persist > 0 && persist < cnt ? persist : cnt;
    hipLaunchKernelGGL((wgrad_grouped_kernel<kWgradBKP, 2>), dim3((unsigned)grid), dim3(256), 0, s, (const WgradG*)table_dev, n,
                       (unsigned char*)workspace, off, off + cnt);
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
