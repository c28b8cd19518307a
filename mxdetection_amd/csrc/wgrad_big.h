// wgrad_big.h -- the 256(co) x 256(ci) weight-gradient tile: 8 waves (2 x 4, 128 x 64 accumulators each), 64-pixel
// steps, 128 KiB of LDS (one workgroup per CU).
//
// Why a second tile: the 128 x 128 tile of wgrad_tile.h moves (128 + 128) x 2 B of operands from L2 to LDS per
// 128 x 128 x 2 flop of one pixel, 64 flop/B; the chip sustains ~17 TB/s on that path (all CUs gathering rows through
// LDS-DMA), which caps the tile at ~1.1 PFLOP/s and measured 0.71. This tile is 128 flop/B: the L2 path stops being
// the bound, and a tile's nine taps re-read the same dy rows half as often.
//
// Per step the LDS holds dy [64 px][256 co] and x [64 px][256 ci] (bf16, 512-B rows, 32-B granules XOR-swizzled by
// (row&3)|((row>>3)&1)<<2 on the low three granule bits -- the same involution on the DMA source chunk and on the
// transposed reads, conflict-free for ds_read_b64_tr_b16: the eight rows a 32-lane half touches land in eight
// different 32-B bank slots). Fragments come from ds_read_b64_tr_b16 (two per 16 x 32 fragment), in inline asm with
// register-tied waits as in wgrad_tile.h (behind the builtin hipcc drains the LDS-DMA ring before every step).
#pragma once
#include "wgrad_tile.h"

namespace mxdet {

constexpr int kBigPx = 64;                         // pixels per ring stage
constexpr int kBigStage = 2 * kBigPx * 512;        // bytes per stage: dy image + x image
constexpr int kBigNS = 2;
constexpr int kBigLds = kBigNS * kBigStage;        // 128 KiB

#ifdef MXDET_WGB_STAMP
// Diagnostic build only (tools/build_stamp.sh): cycles per schedule segment of workgroup 0, waves 0 and 4, summed over
// the K loop. [group][phase*5 + {issue, vmcnt wait, reads + barrier + lgkmcnt, MFMAs, second barrier}]. Nothing in the
// kernel reads these; stamps sit only where no LDS read is outstanding (s_memtime returns through lgkmcnt).
__device__ unsigned long long g_wgb_stamp[2][24];
#define MXDET_WGB_T(k)                                         \
  do {                                                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    st_acc[k] += t_ - st_prev;                                 \
    st_prev = t_;                                              \
  } while (0)
#else
#define MXDET_WGB_T(k) do { } while (0)
#endif

// b: tile index inside the item (main tiles only; bias partial sums stay with the 256-thread kernel)
__device__ __forceinline__ void wgrad_big_tile(const WgradP& p, int b, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  {
    const int nwg = p.big_nwg;
    int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntaps = p.KH * p.KW;
  const int tap = b % ntaps; b /= ntaps;
  const int ci_t = b % p.big_ci_tiles; b /= p.big_ci_tiles;
  const int co_t = b % p.big_co_tiles; b /= p.big_co_tiles;
  const int ks = b;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = co_t * 256, ci0 = ci_t * 256;

  const int step0 = ks * p.big_steps_per_split;
  int nsteps = ceil_div(p.M, kBigPx) - step0;
  nsteps = nsteps > p.big_steps_per_split ? p.big_steps_per_split : nsteps;

  // ---- staging -----------------------------------------------------------------------------------------------------
  // A stage (64 pixels) is staged as four half-tiles of 16 KiB: x and dy rows 0..31 (k-half 0), x and dy rows 32..63
  // (k-half 1). A half-tile is two LDS-DMA instructions per wave (2 pixel rows x 512 B each): wave w, instruction i fills
  // rows h*32 + 2*(2w+i), +1; lane l covers row (l>>5), physical 16-B slot (l&31), i.e. logical chunk
  // ((slot>>1) ^ f(row)) * 2 + (slot & 1).
  const int lrow = lane >> 5, lslot = lane & 31;
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(p.dy, 2u * (unsigned)p.M * (unsigned)p.Cout);
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x, 2u * (unsigned)(p.N * p.H * p.W) * (unsigned)p.Cin);
  const int HW = p.Ho * p.Wo;
  const int d_img = kBigPx / HW, d_rem = kBigPx - d_img * HW;
  const int d_ho = d_rem / p.Wo, d_wo = d_rem - d_ho * p.Wo;
  // walker state of this lane's four pixel rows, k = h*2 + i; x and dy halves of one stage are issued in different
  // phases, so each keeps its own pixel counter
  int c_ho[4], c_wo[4], c_mx[4], c_my[4], c_offy[4], c_offx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = (k >> 1) * 32 + (wid * 2 + (k & 1)) * 2 + lrow;
    const int m = step0 * kBigPx + row;
    c_mx[k] = c_my[k] = m;
    const int img = m / HW;
    const int rem = m - img * HW;
    c_ho[k] = rem / p.Wo;
    c_wo[k] = rem - c_ho[k] * p.Wo;
    c_offy[k] = m * p.Cout + co0;
    // element offset of the tap's source pixel (may lie outside the map: then the lane reads nothing)
    c_offx[k] = ((img * p.H + c_ho[k] * p.stride - p.pad + kh) * p.W + c_wo[k] * p.stride - p.pad + kw) * p.Cin + ci0;
  }
  // the source chunk of this lane per instruction i (the swizzle function of a row does not change with +32 rows)
  int chunk_i[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wid * 2 + i) * 2 + lrow;                       // (+32 for k-half 1: same f)
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    chunk_i[i] = ((((lslot >> 1) ^ f) << 1) | (lslot & 1)) * 8;     // first channel of this lane's 16 bytes
  }
  const int stepy = kBigPx * p.Cout;
  // d(offx) for d(img, ho, wo) = (d_img + ch, d_ho + cw - ch*Ho, d_wo - cw*Wo)
  const int stepx0 = ((d_img * p.H + d_ho * p.stride) * p.W + d_wo * p.stride) * p.Cin;
  const int stepx_w = (p.stride * p.W - p.Wo * p.stride) * p.Cin;
  const int stepx_h = (p.H * p.W - p.Ho * p.stride * p.W) * p.Cin;
  auto issue_y = [&](int h, int buf, bool live) {
    unsigned char* sy = smem + (size_t)buf * kBigStage + h * 16384;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = h * 2 + i;
      const bool ok = live && c_my[k] < p.M && (co0 + chunk_i[i]) < p.Cout;
      // buffer form (descriptor in SGPRs + 32-bit byte offset per lane): a fifth of the issue cost of the 64-bit
      // per-lane address form (tools/micro/dma_rate.hip), and lanes past the end write zeros by themselves
      const unsigned vo = ok ? 2u * (unsigned)(c_offy[k] + chunk_i[i]) : kDmaOob;
#ifndef MXDET_WGB_ABL_NODMA
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lptr_t)(sy + (wid * 2 + i) * 1024), 16, (int)vo, 0, 0, 0);
#else
      asm volatile("" ::"v"(vo));
#endif
      c_my[k] += kBigPx;
      c_offy[k] += stepy;
    }
  };
  auto issue_x = [&](int h, int buf, bool live) {
    unsigned char* sx = smem + (size_t)buf * kBigStage + kBigPx * 512 + h * 16384;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = h * 2 + i;
      const int hi = c_ho[k] * p.stride - p.pad + kh, wi = c_wo[k] * p.stride - p.pad + kw;
      const bool ok = live && c_mx[k] < p.M && (ci0 + chunk_i[i]) < p.Cin && ((unsigned)hi < (unsigned)p.H) &&
                      ((unsigned)wi < (unsigned)p.W);
      const unsigned vo = ok ? 2u * (unsigned)(c_offx[k] + chunk_i[i]) : kDmaOob;
#ifndef MXDET_WGB_ABL_NODMA
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)(sx + (wid * 2 + i) * 1024), 16, (int)vo, 0, 0, 0);
#else
      asm volatile("" ::"v"(vo));
#endif
      // advance 64 pixels: (img, ho, wo) by an exact carry chain, the source offset by the matching constants
      c_mx[k] += kBigPx;
      c_wo[k] += d_wo;
      const bool cw = c_wo[k] >= p.Wo;
      c_wo[k] -= cw ? p.Wo : 0;
      c_ho[k] += d_ho + (cw ? 1 : 0);
      const bool ch = c_ho[k] >= p.Ho;
      c_ho[k] -= ch ? p.Ho : 0;
      c_offx[k] += stepx0 + (cw ? stepx_w : 0) + (ch ? stepx_h : 0);
    }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // transposed-read geometry: lane 16g + 4q + pp addresses pixel row 8g+q (second read: +4 rows = +2048 B; second
  // 32-pixel half: +32 rows = +16384 B), channels 4pp..4pp+3 of a 16-channel granule; f(row) is the same for all four
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int rowa = 8 * g + q;
  const int fa = q | ((g & 1) << 2);
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned offy[8], offx[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) offy[i] = (unsigned)(rowa * 512 + (((wm * 8 + i) ^ fa) << 5) + pp * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) offx[j] = (unsigned)(kBigPx * 512 + rowa * 512 + (((wn * 4 + j) ^ fa) << 5) + pp * 8);

  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  // ---- the schedule ------------------------------------------------------------------------------------------------
  // Four phases per 64-pixel step, 16 MFMAs each: (k-half 0, co rows 0..63), (k-half 0, rows 64..127), (k-half 1, ...).
  // A phase is  R: [issue one half-tile of a later step | counted wait]                                  barrier
  //             M: [16 MFMAs of THIS phase, the fragment reads of the NEXT phase in their gaps | lgkmcnt(0)] barrier.
  // Waves 4..7 run one barrier interval behind waves 0..3 (one extra barrier before their first phase, one after the
  // last phase of waves 0..3): on every SIMD one wave is in M while its partner is in R, so the matrix pipe sees an
  // MFMA cluster in every interval while the other wave pays the LDS-DMA issue and the barrier skew. The transposed
  // reads move 8 B per lane (48 per wave and step): issued back to back by the one wave of a SIMD that is not
  // multiplying they took as long as the MFMAs themselves (measured: the step did not get shorter without either);
  // between the MFMAs of the same wave they are free.
  // Staging runs 7 phases ahead (half-tile k = 4*step + {x-k0, dy-k0, x-k1, dy-k1} is issued in phase k - 7): phase 0
  // issues dy-k1 of step s+1, phase 1 x-k0(s+2), phase 2 dy-k0(s+2), phase 3 x-k1(s+2). Rules (the partner is one
  // interval off): a half-tile is read in the M of the phase AFTER the R whose counted wait retires it; a slot is
  // re-filled two phases after the M that read it last.
  //   reads in M0: dy-k0 rows 64..127 | M1: x-k1, dy-k1 rows 0..63 | M2: dy-k1 rows 64..127 | M3: x-k0, dy-k0(0..63) of s+1
  //   wait in R0: x-k1, dy-k1 of this step landed (four younger half-tiles may be in flight: vmcnt(8));
  //   wait in R2: x-k0, dy-k0 of the next step.
  s16x8_t bxA[4], bxB[4], ayA[4], ayB[4];
  s16x4_t rlo[8], rhi[8];
#define MXDET_WGB_RD(k, ad)                                                        \
  do {                                                                             \
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(rlo[k]) : "v"(ad));            \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(rhi[k]) : "v"(ad)); \
  } while (0)
#define MXDET_WGB_PACK(dst, k) \
  dst = (s16x8_t){rlo[k][0], rlo[k][1], rlo[k][2], rlo[k][3], rhi[k][0], rhi[k][1], rhi[k][2], rhi[k][3]}
#ifndef MXDET_WGB_ABL_NOMFMA
#define MXDET_WGB_MFMA1(ay, bx, mh, i_, j_)                                                                   \
  acc[(mh) * 4 + (i_)][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ay[i_]),     \
                                                                     __builtin_bit_cast(bf16x8_t, bx[j_]),     \
                                                                     acc[(mh) * 4 + (i_)][j_], 0, 0, 0)
#else
#define MXDET_WGB_MFMA1(ay, bx, mh, i_, j_) asm volatile("" ::"v"(ay[i_]), "v"(bx[j_]))
#endif
  // M segment: 16 MFMAs of (ay, bx) into rows mh*4.., with NR (8 or 16 -> 4 or 8 fragments) reads of the next phase
  // spread over the gaps. rd(k) issues the two reads of fragment k; done() packs them into their registers.
#define MXDET_WGB_M(ay, bx, mh, NF, RDK, DONE)                                          \
  do {                                                                                   \
    _Pragma("unroll") for (int t_ = 0; t_ < 16; ++t_) {                                  \
      MXDET_WGB_MFMA1(ay, bx, mh, t_ >> 2, t_ & 3);                                      \
      if (t_ < (NF)) RDK(t_);      /* front-loaded: the last read has 8+ MFMAs (128+ cycles) to land */ \
      __builtin_amdgcn_sched_barrier(0);                                                 \
    }                                                                                    \
    DONE;                                                                                \
  } while (0)
#ifndef MXDET_WGB_ABL_NOBAR
#define MXDET_WGB_BAR()                 \
  do {                                  \
    __builtin_amdgcn_sched_barrier(0);  \
    __builtin_amdgcn_s_barrier();       \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)
#else
#define MXDET_WGB_BAR() asm volatile("" ::: "memory")
#endif
#define MXDET_WGB_LGKM0_4(r)  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]))

  const bool late = wid >= 4;           // wave-uniform (wid comes from readfirstlane)
  unsigned sb = smem_addr;              // LDS byte address of the stage being multiplied
  // fragment k of a read set: k < 4 -> dy fragment (rows mh*4 + k), k >= 4 -> x fragment k - 4
#define MXDET_WGB_RD_Y(k, h, mh) MXDET_WGB_RD(k, sbn + offy[(mh) * 4 + (k)] + (unsigned)((h) * 16384))
#define MXDET_WGB_RD_XY(k, h) \
  do { if ((k) < 4) { MXDET_WGB_RD(k, sbn + offy[k] + (unsigned)((h) * 16384)); } else { MXDET_WGB_RD(k, sbn + offx[(k) - 4] + (unsigned)((h) * 16384)); } } while (0)

  // ---- prologue: "phase -2" = R (seven half-tiles, x-k0 / dy-k0 of step 0 retired) + empty M; "phase -1" = empty R +
  // M with reads only (operands of phase 0)
  issue_x(0, 0, 0 < nsteps);
  issue_y(0, 0, 0 < nsteps);
  issue_x(1, 0, 0 < nsteps);
  issue_y(1, 0, 0 < nsteps);
  issue_x(0, 1, 1 < nsteps);
  issue_y(0, 1, 1 < nsteps);
  issue_x(1, 1, 1 < nsteps);
  if (late) MXDET_WGB_BAR();
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  MXDET_WGB_BAR();
  MXDET_WGB_BAR();
  MXDET_WGB_BAR();
  {
    const unsigned sbn = sb;
#ifndef MXDET_WGB_ABL_NOREAD
#pragma unroll
    for (int k = 0; k < 8; ++k) MXDET_WGB_RD_XY(k, 0);
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) { MXDET_WGB_PACK(ayA[k], k); MXDET_WGB_PACK(bxA[k], k + 4); }
    MXDET_WGB_LGKM0_4(ayA);
    MXDET_WGB_LGKM0_4(bxA);
  }
  MXDET_WGB_BAR();
#ifdef MXDET_WGB_ABL_NOREAD
#define MXDET_WGB_RDK_Y01(k) do { } while (0)
#define MXDET_WGB_RDK_XY1(k) do { } while (0)
#define MXDET_WGB_RDK_Y11(k) do { } while (0)
#define MXDET_WGB_RDK_XY0N(k) do { } while (0)
#else
#define MXDET_WGB_RDK_Y01(k) MXDET_WGB_RD_Y(k, 0, 1)
#define MXDET_WGB_RDK_XY1(k) MXDET_WGB_RD_XY(k, 1)
#define MXDET_WGB_RDK_Y11(k) MXDET_WGB_RD_Y(k, 1, 1)
#define MXDET_WGB_RDK_XY0N(k) MXDET_WGB_RD_XY(k, 0)
#endif
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1, nxt = cur ^ 1;
    // ---- phase 0: k-half 0, co rows 0..63 (ayA, bxA); reads dy-k0 rows 64..127 -> ayB ----
    issue_y(1, nxt, st + 1 < nsteps);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    MXDET_WGB_BAR();
    {
      const unsigned sbn = sb;
      MXDET_WGB_M(ayA, bxA, 0, 4, MXDET_WGB_RDK_Y01, ({
        _Pragma("unroll") for (int k = 0; k < 4; ++k) MXDET_WGB_PACK(ayB[k], k);
        MXDET_WGB_LGKM0_4(ayB); }));
    }
    MXDET_WGB_BAR();
    // ---- phase 1: k-half 0, co rows 64..127 (ayB, bxA); reads x-k1 -> bxB, dy-k1 rows 0..63 -> ayA ----
    issue_x(0, cur, st + 2 < nsteps);
    MXDET_WGB_BAR();
    {
      const unsigned sbn = sb;
      MXDET_WGB_M(ayB, bxA, 1, 8, MXDET_WGB_RDK_XY1, ({
        _Pragma("unroll") for (int k = 0; k < 4; ++k) { MXDET_WGB_PACK(ayA[k], k); MXDET_WGB_PACK(bxB[k], k + 4); }
        MXDET_WGB_LGKM0_4(ayA); MXDET_WGB_LGKM0_4(bxB); }));
    }
    MXDET_WGB_BAR();
    // ---- phase 2: k-half 1, co rows 0..63 (ayA, bxB); reads dy-k1 rows 64..127 -> ayB ----
    issue_y(0, cur, st + 2 < nsteps);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    MXDET_WGB_BAR();
    {
      const unsigned sbn = sb;
      MXDET_WGB_M(ayA, bxB, 0, 4, MXDET_WGB_RDK_Y11, ({
        _Pragma("unroll") for (int k = 0; k < 4; ++k) MXDET_WGB_PACK(ayB[k], k);
        MXDET_WGB_LGKM0_4(ayB); }));
    }
    MXDET_WGB_BAR();
    // ---- phase 3: k-half 1, co rows 64..127 (ayB, bxB); reads x-k0, dy-k0 rows 0..63 of the NEXT step -> bxA, ayA ----
    issue_x(1, cur, st + 2 < nsteps);
    MXDET_WGB_BAR();
    {
      const unsigned sbn = smem_addr + (unsigned)nxt * (unsigned)kBigStage;
      MXDET_WGB_M(ayB, bxB, 1, 8, MXDET_WGB_RDK_XY0N, ({
        _Pragma("unroll") for (int k = 0; k < 4; ++k) { MXDET_WGB_PACK(ayA[k], k); MXDET_WGB_PACK(bxA[k], k + 4); }
        MXDET_WGB_LGKM0_4(ayA); MXDET_WGB_LGKM0_4(bxA); }));
      sb = sbn;
    }
    MXDET_WGB_BAR();
  }
  if (!late) MXDET_WGB_BAR();
#undef MXDET_WGB_RD
#undef MXDET_WGB_PACK
#undef MXDET_WGB_MFMA1
#undef MXDET_WGB_M
#undef MXDET_WGB_BAR
#undef MXDET_WGB_LGKM0_4
#undef MXDET_WGB_RD_Y
#undef MXDET_WGB_RD_XY
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                     // the ring becomes the epilogue's staging area

  // Epilogue. D layout: col = lane&15 -> ci, row = (lane>>4)*4 + r -> co. Each wave stages 16 co rows x 64 ci floats
  // at a time in its own LDS patch and writes them back as whole 256-B rows (four rows per store instruction).
  const size_t Ktot = (size_t)p.KH * p.KW * p.Cin;
  const bool single = p.big_ksplit == 1 && !p.force_slab;
  float* out = single ? p.dw : p.slab + (size_t)ks * p.Cout * Ktot;
  const bool add_old = single && p.accumulate;
  constexpr int EPS = 68;                               // floats per staged row (64 + pad: conflict-free float4 reads)
  float* ep = (float*)smem + wid * 16 * EPS;
  const int er = lane >> 4, ec = (lane & 15) * 4;       // read-back: row er (+4 per pass), 4 floats at column ec
  const int ci = ci0 + wn * 64 + ec;
  const bool ciok = ci < p.Cin;                         // Cin % 8 == 0: a float4 is in or out as a whole
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[((lane >> 4) * 4 + r) * EPS + j * 16 + (lane & 15)] = acc[i][j][r];
    // wave-private staging: the LDS operations of one wave execute in order
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int row = ps * 4 + er;
      const int co = co0 + wm * 128 + i * 16 + row;
      float4 v = *(const float4*)(ep + row * EPS + ec);
      if (co < p.Cout && ciok) {
        float* dst = out + (size_t)co * Ktot + (size_t)tap * p.Cin + ci;
        if (add_old) {
          const float4 o = *(const float4*)dst;
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *(float4*)dst = v;
      }
    }
  }
}

}  // namespace mxdet
