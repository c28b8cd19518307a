// wgrad_tile.h -- the one-tap weight-gradient tile (device function) and the parameter blocks of both tile kinds, used by
// wgrad.hip (plain, grouped and mixed launches). See wgrad.hip for the algorithm notes.
#pragma once
#include "common.h"

namespace mxdet {


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#ifndef MXDET_WGRAD_BKP
#define MXDET_WGRAD_BKP 32
#endif
constexpr int kWgradBKP = MXDET_WGRAD_BKP;   // pixels per ring stage (32: 16 KiB per stage; 64: two MFMA k-steps per barrier)
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
  // three-tap tiles (wgrad3_tile.h; 3x3 / stride 1 / pad 1 layers). An item is tiled by ONE of the two kernels:
  // t3_nwg > 0 means nwg_main == 0 and only the bias workgroups (if any) stay with the one-tap kernel, with ksplit /
  // steps_per_split describing the same pixel ranges as t3_ksplit / t3_steps (64-pixel steps).
  int t3_nwg, t3_ci_tiles, t3_ksplit, t3_steps;
};

// 1x1 / stride 1 / pad 0 (and FC) layers take the PLAIN form of the tile's load bookkeeping (wave-uniform)
__device__ __forceinline__ bool wgrad_plain(const WgradP& p) {
  return p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
}

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

// One 32-pixel MFMA k-step of a stage (HH = which 32 rows of the stage's images): 16 transposed fragment reads, 16 MFMAs.
// Fragment reads in inline asm: behind the ds_read_tr builtin hipcc cannot tell that the read does not touch the ring
// slot an LDS-DMA is still filling and drains vmcnt(0) before the first read of every step (measured: the ring then
// overlaps nothing). The asm reads are ordered by hand: LDS returns in issue order, the first fence (lgkmcnt(4)) releases
// the x fragments and the first two dy fragments, the second the rest, so the last four reads are still in flight under
// the first eight MFMAs. The fences name the registers they release.
template <int BKP, int HH>
__device__ __forceinline__ void wgrad_half(unsigned sbase, const unsigned (&offy)[4], const unsigned (&offx)[4],
                                           f32x4_t (&acc)[4][4]) {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t bx[4], ay[4];
    // (each fragment is packed right where its two reads are issued: with the reads hoisted into arrays first the
    // compiler copied every half into place -- 64 v_mov per step on a loop that is vector-issue bound)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned ad = sbase + (unsigned)(BKP * 256) + offx[j];
      s16x4_t lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ad), "n"(HH * 8192));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ad), "n"(HH * 8192 + 1024));
      bx[j] = (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned ad = sbase + offy[i];
      s16x4_t lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ad), "n"(HH * 8192));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ad), "n"(HH * 8192 + 1024));
      ay[i] = (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
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
}

// One workgroup = one 128(co) x 128(ci) tile of one tap over a range of 64-pixel steps. Global -> LDS is a
// 2-stage LDS-DMA ring (global_load_lds_dwordx4, 1 KiB per wave instruction = 4 pixel rows x 256 B): the loads
// of step t+1 are in flight while step t is multiplied; one s_waitcnt vmcnt(0) + one raw s_barrier per step.
// The LDS image is lane-linear, so the granule swizzle is applied to the per-lane SOURCE chunk.
// PLAIN: the layer is a 1x1 / stride 1 / pad 0 convolution (or an FC layer): source pixel = output pixel, so a lane's two
// offsets advance by constants and the descriptor's range check alone ends the tensor -- 8 vector instructions of
// bookkeeping per step instead of ~50 (the K loop is issue-bound, see below).
template <int BKP, int NS, bool PLAIN>
__device__ __forceinline__ void wgrad_tile(const WgradP& p, int b, unsigned char* lds_pool) {
  constexpr int GI = BKP / 16;      // DMA instructions per wave per image per stage (4 pixel rows each)
  static_assert(NS >= 2 && (NS - 2) * 2 * (BKP / 16) <= 63, "vmcnt is a 6-bit counter");
  typedef unsigned char lds_img_t[2][BKP * 256];                                  // [dy|x] image of one stage
  lds_img_t* const smem = (lds_img_t*)lds_pool;                                    // [buf][dy|x], NS * 2 * BKP * 256 bytes, the kernel's
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
  // the operand with more channels is the one to share inside an XCD: its tile index moves slowest, so the workgroups
  // next to each other re-read THAT tile from L2 (the FC6 weight gradient, 98 x 8 tiles, fetched its 25.7 MB input
  // eight times -- once per XCD -- with ci fastest)
  int ci_t, co_t;
  if (p.Cin > p.Cout) { co_t = b % p.co_tiles; b /= p.co_tiles; ci_t = b % p.ci_tiles; b /= p.ci_tiles; }
  else { ci_t = b % p.ci_tiles; b /= p.ci_tiles; co_t = b % p.co_tiles; b /= p.co_tiles; }
  const int ks = b;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = co_t * 128, ci0 = ci_t * 128;

  const int step0 = ks * p.steps_per_split;
  int nsteps = ceil_div(p.M, BKP) - step0;
  nsteps = nsteps > p.steps_per_split ? p.steps_per_split : nsteps;

  // DMA geometry: wave w, instruction i (0..3) fills pixel rows 4*(4w+i) .. +3 of both images; lane l covers
  // row (l>>4), physical 16-B slot (l&15), i.e. logical chunk ((slot>>1) ^ f(row)) * 2 + (slot & 1)
  const int lrow = lane >> 4, lslot = lane & 15;
  // LDS-DMA in the buffer form: descriptor in SGPRs + 32-bit byte offset per lane (a fifth of the issue cost of 64-bit
  // per-lane addresses, tools/micro/dma_rate.hip); out-of-range lanes write zeros by themselves
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(p.dy, 2u * (unsigned)p.M * (unsigned)p.Cout);
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x, 2u * (unsigned)(p.N * p.H * p.W) * (unsigned)p.Cin);
  // Each lane walks its GI pixel rows BKP pixels per step with an exact carry chain (BKP = d_img*HW + d_ho*Wo + d_wo,
  // every component below its modulus), keeps 32-bit element offsets, and selects the zero page without branches.
  const int HW = p.Ho * p.Wo;
  const int d_img = BKP / HW, d_rem = BKP - d_img * HW;
  const int d_ho = d_rem / p.Wo, d_wo = d_rem - d_ho * p.Wo;
  // (the K loop is VALU-issue bound -- four waves per SIMD share one vector issue port with their MFMAs -- so the
  // per-step bookkeeping is adds, compares and selects only: a row carries the SOURCE coordinates of this tap (hi, wi)
  // and its source offset; all three advance by constants picked by the two carries -- no multiplies, no recomputation)
  int c_hi[GI], c_wi[GI], c_m[GI], c_offy[GI], c_offx[GI];
  bool c_yok[GI], c_xok[GI];
  const int stepx0 = ((d_img * p.H + d_ho * p.stride) * p.W + d_wo * p.stride) * p.Cin;
  const int stepx_w = (p.stride * p.W - p.Wo * p.stride) * p.Cin;
  const int stepx_h = (p.H * p.W - p.Ho * p.stride * p.W) * p.Cin;
  const int d_wo_s = d_wo * p.stride, d_ho_s = d_ho * p.stride;
  const int wrap_w = p.Wo * p.stride, wrap_h = p.Ho * p.stride;
  const int wi_lim = wrap_w - p.pad + kw, hi_lim = wrap_h - p.pad + kh;    // first coordinate of the NEXT row / image
#pragma unroll
  for (int i = 0; i < GI; ++i) {
    int row = (wid * GI + i) * 4 + lrow;
    int f = (row & 3) | (((row >> 3) & 1) << 2);
    int chunk = ((((lslot >> 1) ^ f) << 1) | (lslot & 1)) * 8;     // first channel of this lane's 16 bytes
    int m = step0 * BKP + row;
    c_m[i] = m;
    const int img = m / HW;
    int rem = m - img * HW;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    c_hi[i] = ho * p.stride - p.pad + kh;
    c_wi[i] = wo * p.stride - p.pad + kw;
    c_offy[i] = m * p.Cout + co0 + chunk;
    // element offset of the tap's source pixel (may lie outside the map: then the lane reads nothing)
    c_offx[i] = ((img * p.H + c_hi[i]) * p.W + c_wi[i]) * p.Cin + ci0 + chunk;
    c_yok[i] = (co0 + chunk) < p.Cout;
    c_xok[i] = (ci0 + chunk) < p.Cin;
  }
  const int stepy = BKP * p.Cout;
  // PLAIN: byte offsets, pre-multiplied; a lane whose channels lie outside the tensor carries the out-of-range pattern in
  // `kill` for good (OR-ed in: offsets are 16-byte aligned), dummy stages past the end OR it in through a scalar
  unsigned pl_offy[GI], pl_offx[GI], pl_killy[GI], pl_killx[GI];
  if constexpr (PLAIN) {
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      pl_offy[i] = 2u * (unsigned)c_offy[i];
      pl_offx[i] = 2u * (unsigned)c_offx[i];
      pl_killy[i] = c_yok[i] ? 0u : kDmaOob;
      pl_killx[i] = c_xok[i] ? 0u : kDmaOob;
    }
  }
  const unsigned pl_stepy = 2u * (unsigned)BKP * (unsigned)p.Cout, pl_stepx = 2u * (unsigned)BKP * (unsigned)p.Cin;
  auto issue_stage = [&](int buf, bool live) {
    if constexpr (PLAIN) {
      const unsigned dead = live ? 0u : kDmaOob;      // scalar
#pragma unroll
      for (int i = 0; i < GI; ++i) {
        const unsigned vy = pl_offy[i] | pl_killy[i] | dead;
        const unsigned vx = pl_offx[i] | pl_killx[i] | dead;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lptr_t)(smem[buf][0] + (wid * GI + i) * 1024), 16, (int)vy, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)(smem[buf][1] + (wid * GI + i) * 1024), 16, (int)vx, 0, 0, 0);
        pl_offy[i] += pl_stepy;
        pl_offx[i] += pl_stepx;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const bool mok = live && c_m[i] < p.M;
      const bool yok = mok && c_yok[i];
      const bool xok = mok && c_xok[i] && ((unsigned)c_hi[i] < (unsigned)p.H) && ((unsigned)c_wi[i] < (unsigned)p.W);
      const int offx = c_offx[i];
#ifdef MXDET_WG_L2TEST   /* diagnostic: every workgroup streams the same 1 MiB of dy / x (wrong results, timing only) */
      const unsigned vy = yok ? (2u * (unsigned)c_offy[i]) & 0xfffffu : kDmaOob;
      const unsigned vx = xok ? (2u * (unsigned)offx) & 0xfffffu : kDmaOob;
#else
      const unsigned vy = yok ? 2u * (unsigned)c_offy[i] : kDmaOob;
      const unsigned vx = xok ? 2u * (unsigned)offx : kDmaOob;
#endif
#ifndef MXDET_ABL_NOLOAD
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lptr_t)(smem[buf][0] + (wid * GI + i) * 1024), 16, (int)vy, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)(smem[buf][1] + (wid * GI + i) * 1024), 16, (int)vx, 0, 0, 0);
#else
      asm volatile("" ::"v"(vy), "v"(vx));
#endif
      // advance this row by BKP pixels
      c_m[i] += BKP;
      c_offy[i] += stepy;
      const int wi = c_wi[i] + d_wo_s;
      const bool cw = wi >= wi_lim;
      c_wi[i] = wi - (cw ? wrap_w : 0);
      const int hi = c_hi[i] + d_ho_s + (cw ? p.stride : 0);
      const bool ch = hi >= hi_lim;
      c_hi[i] = hi - (ch ? wrap_h : 0);
      c_offx[i] += stepx0 + (cw ? stepx_w : 0) + (ch ? stepx_h : 0);
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

  static_assert(BKP == 32 || BKP == 64, "one or two 32-pixel MFMA k-steps per ring stage");
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
    const unsigned sbase = smem_addr + (unsigned)cur * (2u * BKP * 256u);
    wgrad_half<BKP, 0>(sbase, offy, offx, acc);
    if constexpr (BKP == 64) wgrad_half<BKP, 1>(sbase, offy, offx, acc);
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


struct WgradG {
  WgradP p;                 // slab / bslab hold byte offsets into the workspace
  int block0, nblocks;      // this layer's workgroups: [block0, block0 + nblocks), block0 a multiple of 8
  int bblock0, bnblocks;    // the same in the three-tap kernel's numbering
  int rblock0, wblocks, bblocks;   // fold kernel: first workgroup, workgroups over dw, workgroups over db
  int fold_ksplit;                 // slabs the fold adds up (all items that share this item's dw write into one run)
  long long nparams;
};

}  // namespace mxdet
