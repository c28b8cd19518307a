// wgrad3_tile.h -- the three-tap weight-gradient tile for 3x3 / stride 1 / pad 1 layers (74 % of the weight-gradient
// FLOPs of the step): one workgroup = 128 (co) x 64 (ci) x the THREE kw taps of one kh, over a range of 64-pixel steps.
//
// Why: with one tap per workgroup (wgrad_tile.h) a 3x3 layer reads its dy tile nine times and its x tile nine times
// through L2 -> LDS (the taps of a tile only share through the XCD's L2), and every 16 MFMAs of a wave pay the whole
// per-step bookkeeping (~50 VALU instructions + 4 LDS-DMA issues on a loop that is vector-issue bound). The taps of one
// filter row read the SAME pixels shifted by one: here a stage holds dy [64 px][128 co] and ONE x image of 66 pixel rows
// [-1 .. 64] x 64 ci; the B fragments of tap kw are transposed reads that start kw rows further down. Per 64-pixel step a
// wave issues 6-7 DMA pieces and ~60 VALU for 48 MFMAs (before: 8 pieces and ~100 VALU for 32), the L2 -> LDS bytes per
// flop drop 2x (128 flop/B), the workgroups per layer 2.25x.
//
// Validity without masks: the reduction runs over VIRTUAL pixels -- every image row gets one pad pixel behind it
// (virtual row width W + 1; v -> row R = v / (W+1), column wv = v % (W+1)). A DMA lane whose column is the pad, or whose
// source row (h + kh - 1) lies outside the map, writes zeros (buffer out-of-range offset). Tap kw = 0 at wo = 0 then
// multiplies the previous row's pad, tap kw = 2 at wo = W-1 its own row's pad, and the pad's own dy row is zero: every
// invalid product vanishes by itself, for any W, at 1 / W extra reduction length. (Masking the dy fragments in the
// half-steps that contain a row end was built first: the branch around the MFMAs doubled the accumulator registers.)
//
// LDS: dy rows are 256 B with the 32-B granule swizzle of wgrad_tile.h; x rows are 128 B (64 ci), granule
// c -> c ^ (((row>>1)&1) | ((row>>3)&1)<<1): the eight rows {r..r+3, r+8..r+11} a 32-lane half of ds_read_b64_tr_b16
// touches land on eight different 32-B bank slots for every start row r (so also for the shifted taps).
#pragma once
#include "wgrad_tile.h"

namespace mxdet {

constexpr int kT3Px = 64;                          // pixels per ring stage
constexpr int kT3DyBytes = kT3Px * 256;            // [64 px][128 co] bf16
constexpr int kT3XRows = 72;                       // 66 used (pixels -1 .. 64), nine 8-row DMA pieces
constexpr int kT3XBytes = kT3XRows * 128;          // [72 px][64 ci] bf16
constexpr int kT3Stage = kT3DyBytes + kT3XBytes;   // 25,600 B (a multiple of 1024)

__host__ __device__ inline bool wgrad3_eligible(int KH, int KW, int stride, int pad, int H, int W) {
  // (the row carry of a 64-pixel advance must fit one wrap of the image height)
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && kT3Px / (W + 1) + 1 <= H;
}
// virtual pixels of a layer (one pad pixel per image row)
__host__ __device__ inline long long wgrad3_vpixels(int N, int H, int W) { return (long long)N * H * (W + 1); }

typedef __attribute__((ext_vector_type(8))) short t3_s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned t3_u32x4_t;

// one 32-pixel half of a stage: 20 transposed reads, 24 MFMAs. sb = LDS byte address of the stage.
template <int HH>
__device__ __forceinline__ void wgrad3_half(unsigned sb, const unsigned (&offy)[4], const unsigned (&offxb)[3][2],
                                            f32x4_t (&acc)[3][4][2]) {
  t3_s16x8_t ay[4], bx[3][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned ad = sb + offy[i];
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ad), "n"(HH * 8192));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ad), "n"(HH * 8192 + 1024));
    ay[i] = (t3_s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const unsigned a0 = sb + offxb[kw][0], a1 = sb + offxb[kw][1];   // rows +0..3 / +4..7 of this tap (own swizzles)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned ad0 = j ? (a0 ^ 32u) : a0, ad1 = j ? (a1 ^ 32u) : a1;   // the neighbouring 16-channel granule
      s16x4_t lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ad0), "n"(kT3DyBytes + HH * 4096));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ad1), "n"(kT3DyBytes + HH * 4096));
      bx[kw][j] = (t3_s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  }
  // LDS returns in issue order: the first fence releases the dy fragments and tap 0 (12 of 20 reads), and so on, so
  // the reads of the later taps are still in flight under the MFMAs of the earlier ones. The fences name the registers
  // they release.
#define MXDET_T3_MFMA8(KW)                                                                                           \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                         \
    acc[KW][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ay[i]),                      \
                                                            __builtin_bit_cast(bf16x8_t, bx[KW][j]), acc[KW][i][j], 0, 0, 0)
  asm volatile("s_waitcnt lgkmcnt(8)"
               : "+v"(ay[0]), "+v"(ay[1]), "+v"(ay[2]), "+v"(ay[3]), "+v"(bx[0][0]), "+v"(bx[0][1]));
#ifndef MXDET_ABL_NOMFMA
  MXDET_T3_MFMA8(0);
#endif
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bx[1][0]), "+v"(bx[1][1]));
#ifndef MXDET_ABL_NOMFMA
  MXDET_T3_MFMA8(1);
#endif
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bx[2][0]), "+v"(bx[2][1]));
#ifndef MXDET_ABL_NOMFMA
  MXDET_T3_MFMA8(2);
#endif
  __builtin_amdgcn_sched_barrier(0);
#undef MXDET_T3_MFMA8
}

// b: tile index inside the item. smem: NS * kT3Stage bytes, 1024-aligned.
template <int NS>
__device__ __forceinline__ void wgrad3_tile(const WgradP& p, int b, unsigned char* smem) {
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  {
    const int nwg = p.t3_nwg;
    int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // kh fastest: the three workgroups of a (dy tile, x tile) sit next to each other in one XCD's queue; then the ci
  // tiles (they share the dy tile), then the co tiles, the pixel range slowest
  const int kh = b % 3; b /= 3;
  const int ci_t = b % p.t3_ci_tiles; b /= p.t3_ci_tiles;
  const int co_t = b % p.co_tiles; b /= p.co_tiles;
  const int ks = b;
  const int co0 = co_t * 128, ci0 = ci_t * 64;
  const int W = p.W, H = p.H, Wp = W + 1;
  const int NH = p.N * H;
  const int Mv = NH * Wp;                           // virtual pixels (one pad pixel behind every image row)

  const int step0 = ks * p.t3_steps;
  int nsteps = ceil_div(Mv, kT3Px) - step0;
  nsteps = nsteps > p.t3_steps ? p.t3_steps : nsteps;
  const int vb0 = step0 * kT3Px;                    // first virtual pixel of this workgroup's range

  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(p.dy, 2u * (unsigned)p.M * (unsigned)p.Cout);
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x, 2u * (unsigned)p.M * (unsigned)p.Cin);

  // ---- DMA geometry ------------------------------------------------------------------------------------------------
  // A lane owns one image row of a piece and walks it 64 virtual pixels per step: column wv (+ d_wv, carry into the
  // row), source byte offset (+ a constant, minus one pixel per carry: the pad is not stored). Rows past the tensor need
  // no test (the descriptor's range check writes zeros), channels past Cout / Cin neither (they only feed outputs that
  // are never stored).
  const int d_R = kT3Px / Wp, d_wv = kT3Px - d_R * Wp;
  // dy: 16 pieces of 4 rows x 256 B; wave w issues pieces 4w .. 4w+3. Lane: row (l>>4) of the piece, 16-B slot (l&15).
  int y_wv[4];
  unsigned y_off[4];
  {
    const int lrow = lane >> 4, lslot = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wid * 4 + i) * 4 + lrow;
      const int f = (row & 3) | (((row >> 3) & 1) << 2);
      const int chunk = ((((lslot >> 1) ^ f) << 1) | (lslot & 1)) * 8;
      const int v = vb0 + row;
      const int R = v / Wp;
      y_wv[i] = v - R * Wp;
      y_off[i] = 2u * (unsigned)((R * W + y_wv[i]) * p.Cout + co0 + chunk);
    }
  }
  const unsigned stepy = 2u * (unsigned)((d_R * W + d_wv) * p.Cout), carryy = 2u * (unsigned)p.Cout;
  // x: 9 pieces of 8 rows x 128 B, image row j <-> virtual pixel vb - 1 + j; wave w issues pieces 2w, 2w+1, wave 3 also
  // piece 8 (rows 64, 65 of it are used). Lane: row (l>>3) of the piece, 16-B slot (l&7). hi = h + kh - 1 is the source row.
  int x_wv[3], x_hi[3];
  unsigned x_off[3];
  const int hi_lim = H + kh - 1;                    // hi of the first row of the NEXT image
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int pj = k < 2 ? wid * 2 + k : 8;
    const int row = pj * 8 + (lane >> 3), slot = lane & 7;
    const int g2 = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    const int chunk = (((slot >> 1) ^ g2) << 4) | ((slot & 1) << 3);     // first channel of this lane's 16 bytes
    const int v = vb0 - 1 + row;
    int R, wv;
    if (v < 0) { R = -1; wv = Wp - 1; }             // virtual pixel -1 = the pad of row -1: the carry chain stays exact
    else { R = v / Wp; wv = v - R * Wp; }
    const int img = R < 0 ? -1 : R / H;
    x_wv[k] = wv;
    x_hi[k] = R - img * H + kh - 1;
    x_off[k] = 2u * (unsigned)(((R + kh - 1) * W + wv) * p.Cin + ci0 + chunk);
  }
  const unsigned stepx = 2u * (unsigned)((d_R * W + d_wv) * p.Cin), carryx = 2u * (unsigned)p.Cin;
  auto issue_stage = [&](int buf) {
    unsigned char* sy = smem + buf * kT3Stage;
    unsigned char* sx = sy + kT3DyBytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#ifdef MXDET_WG_L2TEST   /* diagnostic: every workgroup streams the same 1 MiB (wrong results, timing only) */
      const unsigned vy = y_wv[i] < W ? (y_off[i] & 0xfffffu) : kDmaOob;
#else
      const unsigned vy = y_wv[i] < W ? y_off[i] : kDmaOob;
#endif
#ifndef MXDET_ABL_NOLOAD
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lptr_t)(sy + (wid * 4 + i) * 1024), 16, (int)vy, 0, 0, 0);
#else
      asm volatile("" ::"v"(vy));
#endif
      const int w2 = y_wv[i] + d_wv;
      const bool c = w2 >= Wp;
      y_wv[i] = w2 - (c ? Wp : 0);
      y_off[i] += stepy - (c ? carryy : 0u);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k == 2 && wid != 3) break;                // wave-uniform
      const bool ok = (x_wv[k] < W) && ((unsigned)x_hi[k] < (unsigned)H);
#ifdef MXDET_WG_L2TEST
      const unsigned vx = ok ? (x_off[k] & 0xfffffu) : kDmaOob;
#else
      const unsigned vx = ok ? x_off[k] : kDmaOob;
#endif
      const int pj = k < 2 ? wid * 2 + k : 8;
#ifndef MXDET_ABL_NOLOAD
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)(sx + pj * 1024), 16, (int)vx, 0, 0, 0);
#else
      asm volatile("" ::"v"(vx));
#endif
      const int w2 = x_wv[k] + d_wv;
      const bool c = w2 >= Wp;
      x_wv[k] = w2 - (c ? Wp : 0);
      x_off[k] += stepx - (c ? carryx : 0u);
      const int h2 = x_hi[k] + d_R + (c ? 1 : 0);
      x_hi[k] = h2 - (h2 >= hi_lim ? H : 0);
    }
  };

  f32x4_t acc[3][4][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- transposed-read geometry: lane 16g + 4q + pp addresses pixel row 8g+q (second read: +4 rows), channels
  // 4pp..4pp+3 of a 16-channel granule, and receives channel (lane&15) of those rows: reduction positions 8g .. 8g+7
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  unsigned offy[4], offxb[3][2];
  {
    const int rowa = 8 * g + q;
    const int fa = q | ((g & 1) << 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) offy[i] = (unsigned)(rowa * 256 + (((wm * 4 + i) ^ fa) << 5) + pp * 8);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int row = rowa + kw + 4 * s2;         // image row of (pixel + kw - 1); +32 rows (second half) keeps g2
        const int g2 = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
        offxb[kw][s2] = (unsigned)(row * 128 + (((wn * 2) ^ g2) << 5) + pp * 8);
      }
  }
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- NS-deep ring: stages st+1 .. st+NS-1 are in flight while stage st is multiplied ---------------------------------
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) issue_stage(s0);
  int cur = 0, nxt = NS - 1;
  for (int st = 0; st < nsteps; ++st) {
    // this wave's pieces of stage st have landed (wave 3 issues 7 per stage, the others 6) ...
    if constexpr (NS == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (NS == 3) {
      if (wid == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      if (wid == 3) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                     // ... and everyone's; everyone is done with the slot refilled next
    asm volatile("" ::: "memory");
    issue_stage(nxt);
    const unsigned sb = smem_addr + (unsigned)cur * (unsigned)kT3Stage;
    wgrad3_half<0>(sb, offy, offxb, acc);
    wgrad3_half<1>(sb, offy, offxb, acc);
    cur = (cur + 1 == NS) ? 0 : cur + 1;
    nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // D layout: col = lane&15 -> ci, row = (lane>>4)*4 + r -> co (Cout % 8 == 0: the four rows of a lane are in or out
  // together)
  const int Ktot = 9 * p.Cin;
  const bool single = p.t3_ksplit == 1 && !p.force_slab;
  float* out = single ? p.dw : p.slab + (size_t)ks * p.Cout * (size_t)Ktot;
  const bool add_old = single && p.accumulate;
  const int co_l = co0 + wm * 64 + (lane >> 4) * 4, ci_l = ci0 + wn * 32 + (lane & 15);
  out += (size_t)co_l * Ktot + (size_t)(kh * 3) * p.Cin + ci_l;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (co_l + i * 16 < p.Cout && ci_l + j * 16 < p.Cin) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* o = out + (i * 16 + r) * Ktot + kw * p.Cin + j * 16;
            *o = add_old ? *o + acc[kw][i][j][r] : acc[kw][i][j][r];
          }
      }
    }
}

}  // namespace mxdet
