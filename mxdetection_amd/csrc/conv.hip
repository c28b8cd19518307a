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
// fragment (8 consecutive k per lane) is one ds_read_b128. Tiles: BM x BN x 64 per stage, 4 or 8 waves
// of 64x64 (or 32x64) accumulator tiles, v_mfma_f32_16x16x32_bf16, fp32 accumulators. LDS tiles are
// [row][64] bf16 with the 16-B chunk index XOR-swizzled by (row>>1)&7 (conflict-free ds_read_b128, see
// DESIGN.md section 5). Global->LDS is an NS-deep ring of LDS-DMA loads (global_load_lds_dwordx4) with a
// counted s_waitcnt vmcnt and one raw s_barrier per K-step, so 2-3 stages are always in flight.
// Epilogue: accumulators -> LDS (fp32) -> rows of 8 channels per lane: + bias, + residual
// (optionally nearest-upsampled: FPN top-down), ReLU / ReLU-mask, bf16 pack, 16-B coalesced stores.
#include <stdlib.h>

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
  // stride-2 data gradient, parity-grouped rows (PAR): class c = 2*ph + pw with ph = (hd+pad)&1, pw = (wd+pad)&1
  int par_tile0[5];          // first m-tile of each class (prefix sums), [4] = total
  int par_hc[2], par_wc[2];  // number of rows / columns of each parity
  int par_h0[2], par_w0[2];  // first coordinate of each parity
  int m_begin;     // first output row of this launch (a layer may be covered by two launches with different tiles)
  int ksplit;      // > 1 (static 1x1 path only): the grid is ksplit x tiles, split ks multiplies channel slices
                   // [ks * per, (ks + 1) * per) and writes raw fp32 sums to partial[ks][M][Ncols]
  float* partial;
  const void* pf;  // prefetch hint (mxdet_conv_desc_t.prefetch): read, never used
  long long pf_bytes;
  // 1-bit ReLU masks (mxdet_conv_desc_t.relu_bits): byte [pixel][col / 8], bit k = (value of column col + k) > 0.
  // forward: written next to y; data gradient: read instead of the 16-bit mask operand
  unsigned char* bits_out;
  const unsigned char* bits_in;
  int force_cfg;   // 0 = heuristic; 1..4 = a specific tile configuration (tuning / tests)
  // chained 1x1 (CHAIN template parameter: mxdet_conv2d_fwd_chain): y2 = relu?(relu(conv3x3(x) + bias) * w2 + bias2 + res2)
  const uint16_t* chain_w;     // [CHAIN][Ncols] (Ncols == BN == 64: a wave holds every mid channel of its rows)
  const float* chain_bias;     // [CHAIN] or null
  const uint16_t* chain_res;   // [M][CHAIN] or null
  uint16_t* chain_y;           // [M][CHAIN]
  int chain_relu;
  // ... and optionally a second chained 1x1 on the CHAIN-column result (the next block's conv1): y3 = relu?(y2 * w3 + bias3)
  const uint16_t* chain3_w;    // [64][CHAIN] or null
  const float* chain3_bias;    // [64] or null
  uint16_t* chain3_y;          // [M][64]
  int chain3_relu;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifdef MXDET_CONV_STAMP
// diagnostic build (tools/build_conv_stamp.sh): cycle stamps of one workgroup's life, wave 0 of blocks 0 and 300:
// [start, geometry done, first stage landed, K loop done, end]
__device__ unsigned long long g_conv_stamp[2][8];
#define MXDET_CS(k) do { if (cs_on) cs[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MXDET_CS(k) do { } while (0)
#endif

// Tile BM x BN x 64 per stage, WM x WN waves each owning a (BM/WM) x (BN/WN) accumulator tile.
// Global -> LDS goes through an NS-deep ring filled by global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave
// instruction = 8 tile rows): the loads of stage t+NS-1 are issued while stage t is being multiplied, so
// NS-1 stages (48-96 KiB per CU) are in flight and the global-load latency is off the critical path.
// One counted s_waitcnt vmcnt + one raw s_barrier per K-step; the LDS image is lane-linear, so the
// conflict-avoiding XOR swizzle is applied to the per-lane SOURCE chunk and again on the ds_read side.
// TAPS = KH*KW when the caller guarantees a 1x1 (stride 1, pad 0) or a 3x3 (pad 1; stride 1, or any stride in the
// forward direction) layer; 0 = any geometry, all bookkeeping at run time. With TAPS known the K loop is unrolled over the taps: the tap of every stage is a constant, its
// displacement and filter column are loop-invariant scalars, and the ~50 dependent scalar instructions per step that
// walked (kh, kw, channel slice) at run time disappear -- on the 64-row tiles, at 1-3 waves per SIMD, that serial
// bookkeeping cost as many cycles per step as the step's eight MFMAs.
template <int BM, int BN, int WM, int WN, int NS, bool DGRAD, bool PAR = false, int TAPS = 0, int CHAIN = 0>
__device__ __forceinline__ void conv_igemm_tile(const ConvP& p, int bid, const int nwg) {
  constexpr int NW = WM * WN, NTHR = 64 * NW;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int NBUF = NS;
  constexpr int RPI = 8;                              // tile rows per LDS-DMA instruction (8 rows x 128 B)
  constexpr int GA = BM / RPI / NW, GB = BN / RPI / NW;   // LDS-DMA instructions per wave per stage
  constexpr int STAGE = (BM + BN) * 64;               // bf16 elements per stage
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "rows must split evenly over the waves");
  static_assert(MT % 2 == 0, "epilogue stages two m-tiles at a time");
  static_assert(NS >= 2 && (NS - 2) * (GA + GB) <= 63, "vmcnt is a 6-bit counter");
  static_assert(TAPS == 0 || ((TAPS == 1 || TAPS == 9) && !PAR), "static taps: 1x1 or 3x3");
  constexpr int EP_STRIDE = WTN + 4;
  constexpr int EP_BYTES = NW * 32 * EP_STRIDE * 4;
  constexpr int MAIN_BYTES = NBUF * STAGE * 2;
  // CHAIN > 0: a 1x1 convolution to CHAIN columns follows in the same workgroup (see after the K loop); its whole filter
  // (CHAIN x 64 bf16) sits behind the ring / epilogue area for the workgroup's life
  static_assert(CHAIN == 0 || (TAPS == 9 && !DGRAD && WN == 1 && BN == 64 && CHAIN % (8 * NW) == 0),
                "chain: forward 3x3 whose waves own all 64 mid channels of their rows");
  constexpr int CHAIN_OFF = MAIN_BYTES > EP_BYTES ? MAIN_BYTES : EP_BYTES;
  constexpr int SMEM_BYTES = CHAIN_OFF + CHAIN * 128;
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[SMEM_BYTES];
  uint16_t* smem = (uint16_t*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
#ifdef MXDET_CONV_STAMP
  const bool cs_on = (blockIdx.x == 0 || blockIdx.x == 300) && wid == 0;
  unsigned long long cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  MXDET_CS(0);

  // XCD-aware tile order: blocks that share an XCD (bid % 8) walk consecutive tiles, and consecutive
  // tiles share their A rows (n fastest), so the A tile is re-read from that XCD's L2.
  const int pf_bid = bid, pf_nwg = nwg;            // as launched: the prefetch hint's slices are dealt by this index
  int ks = 0, sl_begin = 0, sl_end = p.C >> 6;     // split-K (static 1x1 path): this workgroup's range of channel slices
  int nwg_ = nwg;
  if constexpr (TAPS == 1) {
    if (p.ksplit > 1) {
      const int tiles = p.tiles_m * p.tiles_n;       // a multiple of 8 keeps every tile on its XCD (the host checks)
      ks = bid / tiles;
      bid -= ks * tiles;
      nwg_ = tiles;
      const int per = ((p.C >> 6) + p.ksplit - 1) / p.ksplit;
      sl_begin = ks * per;
      sl_end = sl_begin + per < sl_end ? sl_begin + per : sl_end;
      sl_end = sl_end < sl_begin ? sl_begin : sl_end;
    }
  }
  if constexpr (!PAR) {   // PAR: classes differ 4:2:2:1 in work and are laid out one after the other -- giving each XCD
                          // a contiguous range would put all the heavy tiles on two of the eight; keep the hardware's
                          // round-robin instead
    const int nwg = nwg_;
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  int m0 = p.m_begin + tile_m * BM;
  const int n0 = tile_n * BN;
  // PAR (stride-2 data gradient): rows are grouped by the parity class of (hd+pad, wd+pad). Inside a class every
  // row has the same set of filter taps that land on a source pixel -- (KH-ph+1)/2 x (KW-pw+1)/2 of them, 9 over
  // the four classes of a 3x3 instead of 36 -- so the K loop runs over exactly those taps (none at all for three
  // of the four classes of a 1x1/2 shortcut: such tiles go straight to the epilogue).
  int par_ph = 0, par_pw = 0, par_rows = 0, par_hc = 1, par_wc = 1, nkh = p.KH, nkw = p.KW;
  if constexpr (PAR) {
    int c = 0;
    while (c < 3 && tile_m >= p.par_tile0[c + 1]) ++c;
    par_ph = c >> 1; par_pw = c & 1;
    par_hc = p.par_hc[par_ph]; par_wc = p.par_wc[par_pw];
    par_rows = p.N * par_hc * par_wc;
    m0 = (tile_m - p.par_tile0[c]) * BM;            // row index inside the class
    nkh = (p.KH - par_ph + 1) >> 1; nkw = (p.KW - par_pw + 1) >> 1;
    nkh = nkh < 0 ? 0 : nkh; nkw = nkw < 0 ? 0 : nkw;
  }

  // ---- per-lane gather geometry: lane l of DMA instruction j fills row 8j + (l>>3), 16-B slot l&7 -------
  // The address math is hoisted out of the K loop (the loop was VALU-issue bound on it): per row a base
  // element offset and a bitmask of the taps that fall inside the image are computed once; inside the loop
  // a load costs one add + one bit test + one select, the per-tap displacement being a uniform SALU value.
  const int lrow = lane >> 3, lslot = lane & 7;
  constexpr int SWZ = 7;
  // LDS-DMA in the buffer form (descriptor in SGPRs + 32-bit byte offset per lane): a fifth of the per-wave issue cost
  // of the 64-bit-address form (tools/micro/dma_rate.hip: 20 vs 116 cycles per 1-KiB piece), and a lane whose offset is
  // out of range writes zeros into its LDS slot by itself (padding taps need no zero page and no pointer select)
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x, 2u * (unsigned)(p.N * p.Hs * p.Ws) * (unsigned)p.C);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(p.w, 2u * (unsigned)p.Ncols * (unsigned)(p.KH * p.KW * p.C));
  const int ntaps = p.KH * p.KW;
  const int hw_d = p.Hd * p.Wd;
  const bool smallm = p.M < (1 << 24);
  const float rcp_hw = 1.0f / (float)hw_d, rcp_w = 1.0f / (float)p.Wd;
  const int Ktot = ntaps * p.C;
  const int KT = PAR ? (nkh * nkw * p.C) >> 6 : Ktot >> 6;
  int wrow[GB];          // element offset of this lane's chunk in filter row n
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    int rb = (wid * GB + i) * RPI + lrow;
    int n = n0 + rb;
    n = n < p.Ncols ? n : p.Ncols - 1;   // rows past Ncols read a valid row; their columns are never stored
    wrow[i] = n * Ktot + ((lslot ^ ((rb >> 1) & SWZ)) << 3);
  }
  // static-tap path: the filter pieces of the prologue stages go out right here, before the per-row gather geometry
  // (two divisions and the tap masks per row, ~2 us of a 3x3 workgroup's life) -- the weights of a layer are cold in L2
  // at its first touch, so their latency runs underneath that arithmetic
  auto issue_b = [&](int buf, int tap, int c0, bool live) {
    const int so_b = live ? 2 * (tap * p.C + c0) : 0;
    unsigned char* sbase = smem_raw + (size_t)buf * (STAGE * 2);
#pragma unroll
    for (int i = 0; i < GB; ++i)
#if !defined(MXDET_ABL_NOLOAD) && !defined(MXDET_ABL_NOBLOAD)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(sbase + BM * 128 + (wid * GB + i) * 1024), 16,
                                               (int)(2u * (unsigned)wrow[i]), so_b, 0, 0);
#else
      asm volatile("" ::"v"(wrow[i]), "s"(so_b));
#endif
  };
  if constexpr (TAPS > 0) {
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
      issue_b(s0, s0 % TAPS, (sl_begin + s0 / TAPS) * 64, s0 < (sl_end - sl_begin) * TAPS);
  }
  if constexpr (CHAIN > 0) {
    // the chained filter: CHAIN rows x 64 channels in the B-stage image (8 rows per piece), older than every ring piece of
    // this wave, so the K loop's first counted wait covers it
    const __amdgpu_buffer_rsrc_t rsrc_c = make_rsrc(p.chain_w, 2u * (unsigned)CHAIN * 64u);
#pragma unroll
    for (int i = 0; i < CHAIN / 8 / NW; ++i) {
      const int r = (wid * (CHAIN / 8 / NW) + i) * 8 + lrow;
      const unsigned vo = 2u * (unsigned)(r * 64 + ((lslot ^ ((r >> 1) & 7)) << 3));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lptr_t)(smem_raw + CHAIN_OFF + (wid * (CHAIN / 8 / NW) + i) * 1024), 16,
                                               (int)vo, 0, 0, 0);
    }
  }
  int a_off[GA];
  unsigned a_mask[GA];
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    int ra = (wid * GA + i) * RPI + lrow;
    int m = m0 + ra;
    int coff = (lslot ^ ((ra >> 1) & SWZ)) << 3;
    unsigned mask = 0;
    int off = 0;
    if constexpr (PAR) {
      if (m < par_rows) {
        int img = m / (par_hc * par_wc);
        int rem = m - img * (par_hc * par_wc);
        int ih = rem / par_wc, iw = rem - ih * par_wc;
        int hd = p.par_h0[par_ph] + 2 * ih, wd = p.par_w0[par_pw] + 2 * iw;
        // first valid tap is (kh, kw) = (ph, pw); tap (ph + 2a, pw + 2b) reads source (hs0 - a, ws0 - b)
        int hs0 = (hd + p.pad - par_ph) >> 1, ws0 = (wd + p.pad - par_pw) >> 1;
        off = ((img * p.Hs + hs0) * p.Ws + ws0) * p.C + coff;
        for (int a = 0; a < nkh; ++a)
          for (int b = 0; b < nkw; ++b) {
            int hs = hs0 - a, ws = ws0 - b;
            if (hs >= 0 && ws >= 0 && hs < p.Hs && ws < p.Ws) mask |= 1u << (a * nkw + b);
          }
      }
    } else if (ntaps == 1 && p.stride == 1 && p.pad == 0) {
      // 1x1 / stride 1 (half of the layers of a bottleneck stack, every FC): the source pixel IS the destination
      // pixel -- no divisions, one tap (the general path below is ~40 instructions per division, six per row)
      if (m < p.M) { off = m * p.C + coff; mask = 1u; }
    } else
    if (m < p.M) {
      int img, hd, wd, rem;
      if (smallm) {       // M < 2^24: exact division by a float reciprocal (wave-uniform choice)
        img = fast_divmod(m, hw_d, rcp_hw, &rem);
        hd = fast_divmod(rem, p.Wd, rcp_w, &wd);
      } else {
        img = m / hw_d; rem = m - img * hw_d;
        hd = rem / p.Wd; wd = rem - hd * p.Wd;
      }
      int h0, w0;   // source position of tap (0,0)
      if (DGRAD) {
        if (p.stride == 1) { h0 = hd + p.pad; w0 = wd + p.pad; }
        else { h0 = (hd + p.pad) / p.stride; w0 = (wd + p.pad) / p.stride; }
      } else { h0 = hd * p.stride - p.pad; w0 = wd * p.stride - p.pad; }
      off = ((img * p.Hs + h0) * p.Ws + w0) * p.C + coff;
      if (DGRAD && p.stride != 1) {
        for (int kh = 0; kh < p.KH; ++kh)
          for (int kw = 0; kw < p.KW; ++kw) {
            int th = hd + p.pad - kh, tw = wd + p.pad - kw;
            int hs = th / p.stride, ws = tw / p.stride;
            bool ok = th >= 0 && tw >= 0 && th == hs * p.stride && tw == ws * p.stride && hs < p.Hs && ws < p.Ws;
            if (ok) mask |= 1u << (kh * p.KW + kw);
          }
      } else if (TAPS == 9) {
        // 3x3, pad 1 (forward: any stride; data gradient: stride 1): the middle tap always exists
        const unsigned rlo = (DGRAD ? h0 < p.Hs : h0 >= 0) ? 1u : 0u, rhi = (DGRAD ? h0 - 2 >= 0 : h0 + 2 < p.Hs) ? 4u : 0u;
        const unsigned clo = (DGRAD ? w0 < p.Ws : w0 >= 0) ? 1u : 0u, chi = (DGRAD ? w0 - 2 >= 0 : w0 + 2 < p.Ws) ? 4u : 0u;
        const unsigned rowm = rlo | 2u | rhi, colm = clo | 2u | chi;
        mask = ((rowm & 1u) ? colm : 0u) | (colm << 3) | ((rowm & 4u) ? colm << 6 : 0u);
      } else {
        // a tap is inside the map iff its row and its column are: two small bit masks instead of KH*KW tests
        unsigned rowm = 0, colm = 0;
        for (int kh = 0; kh < p.KH; ++kh) {
          const int hs = DGRAD ? h0 - kh : h0 + kh;
          if (hs >= 0 && hs < p.Hs) rowm |= 1u << kh;
        }
        for (int kw = 0; kw < p.KW; ++kw) {
          const int ws = DGRAD ? w0 - kw : w0 + kw;
          if (ws >= 0 && ws < p.Ws) colm |= 1u << kw;
        }
        for (int kh = 0; kh < p.KH; ++kh)
          if ((rowm >> kh) & 1u) mask |= colm << (kh * p.KW);
      }
    }
    a_off[i] = off;
    a_mask[i] = mask;
  }

  // one stage at an explicit position (static-tap path: kh, kw are constants after unrolling)
  auto issue_at = [&](int buf, int kh, int kw, int c0, bool live) {
    unsigned char* sbase = smem_raw + (size_t)buf * (STAGE * 2);
    const int tap = live ? kh * p.KW + kw : 31;
    const int delta = DGRAD ? c0 - (kh * p.Ws + kw) * p.C : c0 + (kh * p.Ws + kw) * p.C;     // stride 1
    const int koff = live ? (kh * p.KW + kw) * p.C + c0 : 0;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const unsigned vo = ((a_mask[i] >> tap) & 1u) ? 2u * (unsigned)(a_off[i] + delta) : kDmaOob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)(sbase + (wid * GA + i) * 1024), 16, (int)vo, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const unsigned wo = 2u * (unsigned)(wrow[i] + koff);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(sbase + BM * 128 + (wid * GB + i) * 1024), 16, (int)wo, 0, 0, 0);
    }
  };
  // running (channel-slice, tap) position of the NEXT stage to load: uniform scalars, no divisions
  int ld_kt = 0, ld_kh = 0, ld_kw = 0, ld_c0 = 0;
  auto issue_stage = [&](int buf) {
    unsigned char* sbase = smem_raw + (size_t)buf * (STAGE * 2);
    const bool live = ld_kt < KT;
    int tap, delta, koff;   // bit of a_mask, uniform displacement of this tap, column of (tap, slice) in the filter rows
    if constexpr (PAR) {
      tap = live ? ld_kh * nkw + ld_kw : 31;                           // bit 31 is never set in a_mask
      delta = ld_c0 - (ld_kh * p.Ws + ld_kw) * p.C;
      koff = live ? ((par_ph + 2 * ld_kh) * p.KW + (par_pw + 2 * ld_kw)) * p.C + ld_c0 : 0;
    } else {
      tap = live ? ld_kh * p.KW + ld_kw : 31;
      if (DGRAD) delta = ld_c0 - ((ld_kh / p.stride) * p.Ws + (ld_kw / p.stride)) * p.C;
      else delta = ld_c0 + (ld_kh * p.Ws + ld_kw) * p.C;
      koff = live ? (ld_kh * p.KW + ld_kw) * p.C + ld_c0 : 0;
    }
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      unsigned vo = ((a_mask[i] >> tap) & 1u) ? 2u * (unsigned)(a_off[i] + delta) : kDmaOob;
#ifdef MXDET_ABL_ZEROSRC
      vo = kDmaOob;
#endif
#ifndef MXDET_ABL_NOLOAD
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)(sbase + (wid * GA + i) * 1024), 16, (int)vo, 0, 0, 0);
#else
      asm volatile("" ::"v"(vo));
#endif
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      unsigned wo = 2u * (unsigned)(wrow[i] + koff);
#ifdef MXDET_ABL_ZEROSRC
      wo = kDmaOob;
#endif
#ifndef MXDET_ABL_NOLOAD
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(sbase + BM * 128 + (wid * GB + i) * 1024), 16, (int)wo, 0, 0, 0);
#else
      asm volatile("" ::"v"(wo));
#endif
    }
    // K order = (channel slice, kh, kw) with the TAP fastest: the KH*KW shifted re-reads of one 128-B
    // activation chunk happen in consecutive steps, while that chunk is still in this XCD's L2
    ++ld_kt;
    if (++ld_kw == nkw) {
      ld_kw = 0;
      if (++ld_kh == nkh) { ld_kh = 0; ld_c0 += 64; }
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  MXDET_CS(1);

  const int frow = lane & 15, fq = lane >> 4;
  if constexpr (TAPS > 0) {
  // ---- static taps: stride-1 1x1 / 3x3, everything per step that can be a constant is one --------------------------
  constexpr int TKW = TAPS == 9 ? 3 : 1;
  constexpr int SL = (NS % TAPS == 0 || TAPS % NS == 0) ? (TAPS % NS == 0 ? 1 : NS) : NS;   // slices per unrolled block
  static_assert((SL * TAPS) % NS == 0, "the ring position must repeat with the unrolled block");
  // The uniform part of a piece's address (tap displacement + channel slice) rides in the instruction's scalar offset,
  // which the range check ignores; the per-lane part must then be non-negative on its own, so the descriptor starts
  // `bias` elements before the tensor (padding rows / columns of the first image give negative tap-(0,0) offsets).
  // Data gradient: offsets are taken from the LAST tap's source pixel so that every displacement is >= 0.
  const int bias_el = (p.pad * p.Ws + p.pad) * p.C;
  const int rebase = DGRAD ? ((p.KH - 1) * p.Ws + (p.KW - 1)) * p.C : 0;
  const __amdgpu_buffer_rsrc_t rsrc_xs =
      make_rsrc(p.x - bias_el, 2u * (unsigned)(p.N * p.Hs * p.Ws) * (unsigned)p.C + 2u * (unsigned)bias_el);
  unsigned voa[GA];
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    voa[i] = 2u * (unsigned)(a_off[i] - rebase + bias_el);
    if (TAPS == 1) voa[i] = (a_mask[i] & 1u) ? voa[i] : kDmaOob;
  }
  auto issue_a = [&](int buf, int tap, int c0, bool live) {           // buf, tap: constants after unrolling
    const int kh = tap / TKW, kw = tap % TKW;
    const int disp = DGRAD ? ((p.KH - 1 - kh) * p.Ws + (p.KW - 1 - kw)) * p.C : (kh * p.Ws + kw) * p.C;
    const int so_a = live ? 2 * (disp + c0) : 0;                       // dummy stages past the end: see below
    unsigned char* sbase = smem_raw + (size_t)buf * (STAGE * 2);
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      unsigned vo = voa[i];
      if (TAPS != 1) vo = ((a_mask[i] >> tap) & 1u) ? vo : kDmaOob;
      // a dummy stage has no displacement: its tap-(0,0) address may lie in front of the tensor -- it loads nothing
      if (!live) vo = kDmaOob;
#ifndef MXDET_ABL_NOLOAD
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_xs, (lptr_t)(sbase + (wid * GA + i) * 1024), 16, (int)vo, so_a, 0, 0);
#else
      asm volatile("" ::"v"(vo), "s"(so_a));
#endif
    }
  };
  // fragment addresses: two per operand (the two 32-deep halves differ in the swizzled chunk), rows i / j at +2 KiB each
  const uint16_t* fa0 = smem + lds_off(wm * WTM + frow, fq);
  const uint16_t* fa1 = smem + lds_off(wm * WTM + frow, 4 + fq);
  const uint16_t* fb0 = smem + BM * 64 + lds_off(wn * WTN + frow, fq);
  const uint16_t* fb1 = smem + BM * 64 + lds_off(wn * WTN + frow, 4 + fq);
  const int nslices = sl_end;                     // (split-K: this workgroup's slices are [sl_begin, sl_end))
  // (the prologue's filter pieces went out before the geometry; its gather pieces follow them, so stage 0 is complete
  // once only the later stages' gather pieces are outstanding)
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    issue_a(s0, s0 % TAPS, (sl_begin + s0 / TAPS) * 64, s0 < (sl_end - sl_begin) * TAPS);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * GA) : "memory");
#ifdef MXDET_ABL_NOBFRAG
  bf16x8_t bf0[NT], bf1[NT];     // ablation: the filter fragments are read once (stale LDS contents) and kept
#pragma unroll
  for (int j = 0; j < NT; ++j) { bf0[j] = *(const bf16x8_t*)(fb0 + j * 1024); bf1[j] = *(const bf16x8_t*)(fb1 + j * 1024); }
#endif
#ifdef MXDET_ABL_NOKLOOP
  for (int cs = sl_begin; cs < sl_begin; cs += SL) {
#else
  for (int cs = sl_begin; cs < nslices; cs += SL) {
#endif
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      if (SL > 1 && cs + sl >= nslices) break;                        // wave-uniform
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp) {
        const int step = sl * TAPS + tp;                              // constants after unrolling
        const int cur = step % NS, nxt = (step + NS - 1) % NS;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (GA + GB)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // all fragment reads of the step first (immediate offsets), then the next stage's loads, then the MFMAs behind
        // counted lgkmcnt waits: one exposed LDS latency per step, underneath the DMA issue
#ifdef MXDET_ABL_NOBFRAG
        bf16x8_t af0[MT], af1[MT];
#else
        bf16x8_t af0[MT], bf0[NT], af1[MT], bf1[NT];
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i) af0[i] = *(const bf16x8_t*)(fa0 + cur * STAGE + i * 1024);
#ifndef MXDET_ABL_NOBFRAG
#pragma unroll
        for (int j = 0; j < NT; ++j) bf0[j] = *(const bf16x8_t*)(fb0 + cur * STAGE + j * 1024);
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i) af1[i] = *(const bf16x8_t*)(fa1 + cur * STAGE + i * 1024);
#ifndef MXDET_ABL_NOBFRAG
#pragma unroll
        for (int j = 0; j < NT; ++j) bf1[j] = *(const bf16x8_t*)(fb1 + cur * STAGE + j * 1024);
#endif
        __builtin_amdgcn_sched_barrier(0);
        {
          const int ahead = tp + NS - 1;
          const int atap = ahead % TAPS, aslice = ahead / TAPS;
          issue_a(nxt, atap, (cs + sl + aslice) * 64, cs + sl + aslice < nslices);
          issue_b(nxt, atap, (cs + sl + aslice) * 64, cs + sl + aslice < nslices);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#ifndef MXDET_ABL_NOMFMA
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0[i], bf0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1[i], bf1[j], acc[i][j], 0, 0, 0);
#else
#pragma unroll
        for (int i = 0; i < MT; ++i) { asm volatile("" ::"v"(af0[i])); asm volatile("" ::"v"(af1[i])); }
#pragma unroll
        for (int j = 0; j < NT; ++j) { asm volatile("" ::"v"(bf0[j])); asm volatile("" ::"v"(bf1[j])); }
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  } else {
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) issue_stage(s0);

  int cur = 0, nxt = NS - 1;
  for (int kt = 0; kt < KT; ++kt) {
    // stage kt has landed once all but the (NS-2) youngest stages' loads of this wave are done ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (GA + GB)) : "memory");
    // ... in every wave; the same barrier says every wave is done reading the buffer refilled next
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef MXDET_CONV_STAMP
    if (kt == 0) MXDET_CS(2);
#endif
    // Software pipeline inside the step: the fragment reads of BOTH 32-deep halves are issued first (their LDS
    // latency overlaps the address math / DMA issue of the next stage), then the 2*MT*NT MFMAs run back to back.
    const uint16_t* sa = smem + cur * STAGE;
    const uint16_t* sb = sa + BM * 64;
    bf16x8_t af0[MT], bf0[NT], af1[MT], bf1[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af0[i] = *(const bf16x8_t*)(sa + lds_off(wm * WTM + i * 16 + frow, fq));
#pragma unroll
    for (int j = 0; j < NT; ++j) bf0[j] = *(const bf16x8_t*)(sb + lds_off(wn * WTN + j * 16 + frow, fq));
    issue_stage(nxt);
    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT; ++i) af1[i] = *(const bf16x8_t*)(sa + lds_off(wm * WTM + i * 16 + frow, 4 + fq));
#pragma unroll
    for (int j = 0; j < NT; ++j) bf1[j] = *(const bf16x8_t*)(sb + lds_off(wn * WTN + j * 16 + frow, 4 + fq));
#ifndef MXDET_ABL_NOMFMA
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0[i], bf0[j], acc[i][j], 0, 0, 0);
    // interleave: the second half's fragment reads (issued above in program order) are spread between the
    // first half's MFMAs, 1 ds_read per 2 MFMAs, so LDS traffic and matrix work overlap inside one wave
#pragma unroll
    for (int g = 0; g < MT + NT; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, (MT * NT) / (MT + NT), 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       // DS read
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1[i], bf1[j], acc[i][j], 0, 0, 0);
#else
#pragma unroll
    for (int i = 0; i < MT; ++i) { asm volatile("" ::"v"(af0[i])); asm volatile("" ::"v"(af1[i])); }
#pragma unroll
    for (int j = 0; j < NT; ++j) { asm volatile("" ::"v"(bf0[j])); asm volatile("" ::"v"(bf1[j])); }
#endif
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    cur = (cur + 1 == NS) ? 0 : cur + 1;
    nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
  }
  }
  MXDET_CS(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (dummy) tail loads before LDS is re-used
  __syncthreads();
  MXDET_CS(4);

  // ---- next layer's filter: every workgroup touches a slice while its epilogue runs (the loads are consumed only at the
  // very end; they are older than the epilogue's own operand loads, whose wait therefore covers them) ----
  uint4 pfv[4] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  if (p.pf != nullptr) {
    const long long pf_stride = (long long)pf_nwg * NTHR * 16;       // (workgroup index inside this launch / grouped item)
    long long off = ((long long)pf_bid * NTHR + tid) * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k, off += pf_stride)
      if (off + 16 <= p.pf_bytes) pfv[k] = *(const uint4*)((const char*)p.pf + off);
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  float* ep = (float*)smem_raw + wid * 32 * EP_STRIDE;
  constexpr int LPR = WTN / 8;          // lanes per row on read-back
  constexpr int RPP = 64 / LPR;         // rows per pass
  constexpr int PASSES = 32 / RPP;
  const int rl = lane / LPR, cg = lane - rl * LPR;
  // CHAIN: the tile just computed (BM rows x 64 mid channels) is the A operand of a 1x1 convolution to CHAIN columns. A wave
  // owns all 64 channels of its rows (WN == 1), so nothing crosses waves: accumulators -> wave-private fp32 staging ->
  // rows read back in the A-fragment layout, + bias, ReLU, rounded to bf16 exactly as the unfused layer stores them; then
  // per 64-column chunk 2 * MT * NT MFMAs against the filter image in LDS (two 32-deep halves in the order of the
  // stand-alone 1x1 kernel: bit-identical sums) and the ordinary epilogue on the chunk.
  bf16x8_t af2[CHAIN > 0 ? MT : 1][2];
  if constexpr (CHAIN > 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ep[(i * 16 + fq * 4 + r) * EP_STRIDE + j * 16 + frow] = acc[i][j][r];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
      if (p.bias) {
        c0 = *(const float4*)(p.bias + hf * 32 + fq * 8);
        c1 = *(const float4*)(p.bias + hf * 32 + fq * 8 + 4);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float4 v0 = *(const float4*)(ep + (i * 16 + frow) * EP_STRIDE + hf * 32 + fq * 8);
        const float4 v1 = *(const float4*)(ep + (i * 16 + frow) * EP_STRIDE + hf * 32 + fq * 8 + 4);
        float v[8] = {v0.x + c0.x, v0.y + c0.y, v0.z + c0.z, v0.w + c0.w, v1.x + c1.x, v1.y + c1.y, v1.z + c1.z, v1.w + c1.w};
        if (p.relu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]);
        o.w = pack_bf16x2(v[6], v[7]);
        af2[i][hf] = __builtin_bit_cast(bf16x8_t, o);
      }
    }
  }
  // epilogue operands: the layer's own, or the chained convolution's
  // Second chained 1x1 (chain3_w != null): its 64 output columns accumulate over the chunks -- chunk c's STORED values (the
  // bf16 words the epilogue writes) are put back into the wave's staging rows as a bf16 tile, read as A fragments and
  // multiplied with columns [64 c, 64 c + 64) of the filter, whose fragments come straight from L2; one more trip through
  // the loop then runs the ordinary epilogue on that accumulator (K order = chunk, half: that of the stand-alone kernel).
  static_assert(CHAIN == 0 || MT == 2, "chain: one epilogue half covers the wave's rows");
  constexpr int NCH = CHAIN > 0 ? CHAIN / BN : 1;
  const bool c3 = CHAIN > 0 && p.chain3_w != nullptr;
  f32x4_t acc3[CHAIN > 0 ? MT : 1][CHAIN > 0 ? NT : 1];
  if constexpr (CHAIN > 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc3[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll 1      // (unrolled: no faster; the filter fragments straight from L2 instead of LDS, 48 KiB per workgroup: no faster)
  for (int chunk = 0; chunk < (CHAIN > 0 ? NCH + (c3 ? 1 : 0) : 1); ++chunk) {
  const bool last3 = CHAIN > 0 && chunk == NCH;        // the trip that stores the second chained convolution
  const int e_ncols = CHAIN > 0 ? (last3 ? BN : CHAIN) : p.Ncols;
  const float* const e_bias = CHAIN > 0 ? (last3 ? p.chain3_bias : p.chain_bias) : p.bias;
  const uint16_t* const e_res = CHAIN > 0 ? (last3 ? nullptr : p.chain_res) : p.res;
  uint16_t* const e_y = CHAIN > 0 ? (last3 ? p.chain3_y : p.chain_y) : p.y;
  const int e_relu = CHAIN > 0 ? (last3 ? p.chain3_relu : p.chain_relu) : p.relu;
  const int n0c = CHAIN > 0 ? (last3 ? 0 : chunk * BN) : n0;
  if constexpr (CHAIN > 0) {
    if (last3) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = acc3[i][j];
    }
  }
  // the second chained filter's fragments of this chunk: from L2, issued here so that their latency runs under the chunk
  bf16x8_t b3[2][CHAIN > 0 ? NT : 1];
  if constexpr (CHAIN > 0) {
    if (c3 && !last3) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          b3[hf][j] = *(const bf16x8_t*)(p.chain3_w + (size_t)(j * 16 + frow) * CHAIN + chunk * BN + hf * 32 + fq * 8);
    }
  }
  if (CHAIN > 0 && !last3) {
    const uint16_t* wl = (const uint16_t*)(smem_raw + CHAIN_OFF);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      bf16x8_t bw[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bw[j] = *(const bf16x8_t*)(wl + lds_off(n0c + j * 16 + frow, hf * 4 + fq));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af2[i][hf], bw[j], acc[i][j], 0, 0, 0);
    }
  }
  // bias: one pair of loads per lane for the whole epilogue (the column does not depend on the pass)
  const int col = n0c + wn * WTN + cg * 8;
  const bool colok = col < e_ncols;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (e_bias) {
    const int cb = colok ? col : 0;
    b0 = *(const float4*)(e_bias + cb);
    b1 = *(const float4*)(e_bias + cb + 4);
  }
#pragma unroll
  for (int h = 0; h < MT / 2; ++h) {
    // Residual / mask operands of ALL passes of this half are fetched first, unconditionally (rows outside the
    // tensor read element 0): a load inside the per-row `if` makes hipcc branch around it and wait for each one
    // separately -- PASSES dependent round trips per half, most of the run time of a short-K 1x1 layer.
    size_t pixs[PASSES];
    bool oks[PASSES];
    uint4 rres[PASSES], rmsk[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int row = ps * RPP + rl;
      const int m = m0 + wm * WTM + h * 32 + row;
      bool rowok = m < p.M;
      size_t pix = (size_t)m;          // linear output pixel of this row
      if constexpr (PAR) {
        rowok = m < par_rows;
        int img = m / (par_hc * par_wc);
        int rem = m - img * (par_hc * par_wc);
        int ih = rem / par_wc, iw = rem - ih * par_wc;
        pix = (size_t)(img * p.Hd + p.par_h0[par_ph] + 2 * ih) * p.Wd + (p.par_w0[par_pw] + 2 * iw);
      }
      oks[ps] = rowok && colok;
      pixs[ps] = oks[ps] ? pix : 0;
    }
    if (e_res) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        size_t ri = pixs[ps] * e_ncols + (oks[ps] ? col : 0);
        if (CHAIN == 0 && p.res_up) {
          const int m = (int)pixs[ps];
          int img = m / (p.Hd * p.Wd);
          int rem = m - img * (p.Hd * p.Wd);
          int hd = rem / p.Wd, wd = rem - hd * p.Wd;
          int Hc = (p.Hd + 1) >> 1, Wc = (p.Wd + 1) >> 1;
          ri = ((size_t)(img * Hc + (hd >> 1)) * Wc + (wd >> 1)) * e_ncols + (oks[ps] ? col : 0);
        }
        rres[ps] = *(const uint4*)(e_res + ri);
      }
    }
    if (CHAIN == 0 && p.mask) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps)
        rmsk[ps] = *(const uint4*)(p.mask + pixs[ps] * e_ncols + (oks[ps] ? col : 0));
    }
    unsigned rbits[PASSES];
    if (CHAIN == 0 && p.bits_in) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps)
        rbits[ps] = p.bits_in[(pixs[ps] * e_ncols + (oks[ps] ? col : 0)) >> 3];
    }
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
      const int row = ps * RPP + rl;
      float4 v0 = *(const float4*)(ep + row * EP_STRIDE + cg * 8);
      float4 v1 = *(const float4*)(ep + row * EP_STRIDE + cg * 8 + 4);
      if constexpr (TAPS == 1) {
        if (p.ksplit > 1) {                          // split-K: raw sums; bias / residual / ReLU happen in the fold kernel
          if (oks[ps]) {
            float* dst = p.partial + ((size_t)ks * p.M + pixs[ps]) * e_ncols + col;
            *(float4*)dst = v0;
            *(float4*)(dst + 4) = v1;
          }
          continue;
        }
      }
      float v[8] = {v0.x + b0.x, v0.y + b0.y, v0.z + b0.z, v0.w + b0.w, v1.x + b1.x, v1.y + b1.y, v1.z + b1.z, v1.w + b1.w};
      if (e_res) {
        const uint4 rv = rres[ps];
        v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
        v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
        v[4] += __uint_as_float(rv.z << 16); v[5] += __uint_as_float(rv.z & 0xffff0000u);
        v[6] += __uint_as_float(rv.w << 16); v[7] += __uint_as_float(rv.w & 0xffff0000u);
      }
      if (CHAIN == 0 && p.mask) {
        // bf16 > 0  <=>  sign clear and magnitude non-zero
        const unsigned mm[4] = {rmsk[ps].x, rmsk[ps].y, rmsk[ps].z, rmsk[ps].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          unsigned lo = mm[k] & 0xffffu, hi = mm[k] >> 16;
          if (!(lo != 0u && lo < 0x8000u)) v[2 * k] = 0.0f;
          if (!(hi != 0u && hi < 0x8000u)) v[2 * k + 1] = 0.0f;
        }
      }
      if (CHAIN == 0 && p.bits_in) {
        const unsigned mb = rbits[ps];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (!((mb >> k) & 1u)) v[k] = 0.0f;
      }
      if (e_relu && (CHAIN > 0 || (!p.mask && !p.bits_in))) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
      }
      uint4 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      o.z = pack_bf16x2(v[4], v[5]);
      o.w = pack_bf16x2(v[6], v[7]);
      if (oks[ps]) *(uint4*)(e_y + pixs[ps] * e_ncols + col) = o;
      if constexpr (CHAIN > 0) {
        // (this pass's fp32 rows have been read: the bf16 row takes the first 128 bytes of the same staging row)
        if (c3 && !last3) *(uint4*)((unsigned char*)ep + row * (EP_STRIDE * 4) + ((cg ^ (row & 7)) << 4)) = o;
      }
      if (CHAIN == 0 && p.bits_out && oks[ps]) {
        // the mask of the STORED values: bf16 > 0 <=> sign clear and magnitude non-zero (as the 16-bit mask test reads it)
        const unsigned w4[4] = {o.x, o.y, o.z, o.w};
        unsigned mb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned lo = w4[k] & 0xffffu, hi = w4[k] >> 16;
          mb |= ((lo != 0u && lo < 0x8000u) ? 1u : 0u) << (2 * k);
          mb |= ((hi != 0u && hi < 0x8000u) ? 1u : 0u) << (2 * k + 1);
        }
        p.bits_out[(pixs[ps] * e_ncols + col) >> 3] = (unsigned char)mb;
      }
    }
  }
  if constexpr (CHAIN > 0) {
    if (c3 && !last3) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        bf16x8_t a3[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
          a3[i] = *(const bf16x8_t*)((const unsigned char*)ep + (i * 16 + frow) * (EP_STRIDE * 4) +
                                     (((hf * 4 + fq) ^ ((i * 16 + frow) & 7)) << 4));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[i], b3[hf][j], acc3[i][j], 0, 0, 0);
      }
    }
  }
  }   // chunk
  if (p.pf != nullptr) asm volatile("" ::"v"(pfv[0].x), "v"(pfv[1].x), "v"(pfv[2].x), "v"(pfv[3].x));
#ifdef MXDET_CONV_STAMP
  if (cs_on) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cs[5] = __builtin_amdgcn_s_memtime();
    if (lane == 0)
      for (int k = 0; k < 6; ++k) g_conv_stamp[blockIdx.x == 0 ? 0 : 1][k] = cs[k];
  }
#endif
}

template <int BM, int BN, int WM, int WN, int NS, bool DGRAD, bool PAR = false, int TAPS = 0, int CHAIN = 0>
__global__ void __launch_bounds__(64 * WM * WN)
conv_igemm_kernel(ConvP p) {
  conv_igemm_tile<BM, BN, WM, WN, NS, DGRAD, PAR, TAPS, CHAIN>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Grouped form: independent convolutions that share one tile configuration (the 3x3 of every pyramid level of an RPN
// / RetinaNet head, the FPN output convs, ...) as ONE grid. The small levels (a few dozen workgroups, latency-bound
// on their own) ride in the shadow of the large ones. Table in device memory, built once per group by
// mxdet_conv2d_grouped_plan; block0 is a multiple of 8 so the XCD-aware tile order of every item stays aligned.
struct ConvG {
  ConvP p;
  int block0, nblocks;
};

template <int BM, int BN, int WM, int WN, int NS, bool DGRAD, int TAPS = 0>
__global__ void __launch_bounds__(64 * WM * WN)
conv_igemm_grouped_kernel(const ConvG* __restrict__ table, int n) {
  int lo = 0, hi = n - 1;
  const int bid = (int)blockIdx.x;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (table[mid].block0 <= bid) lo = mid; else hi = mid - 1;
  }
  const int b = bid - table[lo].block0;
  const int nb = table[lo].nblocks;
  if (b >= nb) return;                      // alignment padding between items
  const ConvP p = table[lo].p;
  conv_igemm_tile<BM, BN, WM, WN, NS, DGRAD, false, TAPS>(p, b, nb);
}

static thread_local int g_force_cfg = 0;   // tuning hook (mxdet_debug_force_conv_cfg), 0 = heuristic

// Tile-choice thresholds (mxdet_debug_set_tuning overrides them for sweeps: they only tune correctly on whole-step
// A/B runs -- a layer replayed alone keeps its filter in L2 and owns the chip).
static int thr_t64() { return (int)tuning(MXDET_TUNE_T64); }
static int thr_t128() { return (int)tuning(MXDET_TUNE_T128); }
static int thr_par64() { return (int)tuning(MXDET_TUNE_PAR64); }

template <int BM, int BN, int WM, int WN, int NS, bool DGRAD, bool PAR = false, int TAPS = 0, int CHAIN = 0>
static int launch_cfg(ConvP& p, hipStream_t s) {
  if (PAR) {   // rows grouped by parity class: tiles never straddle two classes
    int t = 0;
    for (int c = 0; c < 4; ++c) {
      p.par_tile0[c] = t;
      t += ceil_div(p.N * p.par_hc[c >> 1] * p.par_wc[c & 1], BM);
    }
    p.par_tile0[4] = t;
    p.tiles_m = t;
  }
  if (p.tiles_m <= 0) p.tiles_m = ceil_div(p.M - p.m_begin, BM);   // caller may restrict the row range
  p.tiles_n = ceil_div(p.Ncols, BN);
  long long nwg = (long long)p.tiles_m * p.tiles_n;
  if (TAPS == 1 && p.ksplit > 1) nwg *= p.ksplit;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, NS, DGRAD, PAR, TAPS, CHAIN>), dim3((unsigned)nwg),
                     dim3(64 * WM * WN), 0, s, p);
  return check_launch("conv2d");
}

// Tile choice (measured on MI355X, tools/bench_one_conv.py sweeps, profiles/r01_conv_cfg_sweep.txt): what matters
// most is having 2-3 workgroups resident per CU so that one workgroup's barrier / LDS-latency / DMA-issue phases
// overlap another's MFMA phase; a 2-stage ring (48-64 KiB of LDS) allows that, deeper rings do not pay.
template <bool DGRAD>
static int launch(ConvP& p, hipStream_t s) {
  const int force = g_force_cfg;
  const long long t128 = (long long)ceil_div(p.M, 128) * ceil_div(p.Ncols, 128);
  const long long t64 = (long long)ceil_div(p.M, 64) * ceil_div(p.Ncols, 128);
  const int K = p.KH * p.KW * p.C;
  // forced tile configurations (mxdet_debug_force_conv_cfg: tools/sweep_cfg_all.py, tools/check_cfg.py). 3..8, 15: the
  // run-time-geometry loop; 40..49: the static-tap loop (the caller passes a stride-1 1x1 / 3x3 layer)
#define MXDET_FORCE_ST(BM, BN, WM, WN, NS)                                                  \
  (p.KH * p.KW == 1 ? launch_cfg<BM, BN, WM, WN, NS, DGRAD, false, 1>(p, s) : launch_cfg<BM, BN, WM, WN, NS, DGRAD, false, 9>(p, s))
  switch (force) {
    case 3: return launch_cfg<128, 128, 2, 2, 4, DGRAD>(p, s);
    case 5: return launch_cfg<64, 64, 2, 2, 3, DGRAD>(p, s);
    case 6: return launch_cfg<64, 128, 2, 2, 2, DGRAD>(p, s);
    case 7: return launch_cfg<128, 128, 2, 2, 2, DGRAD>(p, s);
    case 8: return launch_cfg<128, 64, 4, 1, 2, DGRAD>(p, s);
    case 15: return launch_cfg<256, 256, 2, 4, 2, DGRAD>(p, s);   // 8 waves, 128x64 per wave, 128 KiB
    case 40: return MXDET_FORCE_ST(64, 64, 2, 2, 3);
    case 41: return MXDET_FORCE_ST(64, 128, 2, 2, 2);
    case 45: return MXDET_FORCE_ST(128, 128, 2, 2, 2);
    case 46: return MXDET_FORCE_ST(128, 64, 4, 1, 2);
    case 49: return MXDET_FORCE_ST(128, 128, 2, 4, 2);            // 8 waves, 64x32 per wave
    default: break;
  }
#undef MXDET_FORCE_ST
  // stride-1 1x1 / 3x3 layers (all but the four stride-2 convolutions and the stem): the unrolled static-tap K loop
  const int taps = p.KH * p.KW;
  // (a forward 3x3 may be strided: its tap displacements do not depend on the stride; a strided data gradient may not)
  const int st = ((p.stride != 1 && (DGRAD || taps == 1)) || tuning(MXDET_TUNE_STATIC_TAPS) == 0) ? 0
                 : (taps == 1 && p.pad == 0) ? 1 : (p.KH == 3 && p.KW == 3 && p.pad == 1) ? 9 : 0;
#define MXDET_LAUNCH_ST(BM, BN, WM, WN, NS)                                                        \
  (st == 1 ? launch_cfg<BM, BN, WM, WN, NS, DGRAD, false, 1>(p, s)                                 \
           : st == 9 ? launch_cfg<BM, BN, WM, WN, NS, DGRAD, false, 9>(p, s)                       \
                     : launch_cfg<BM, BN, WM, WN, NS, DGRAD>(p, s))
  if constexpr (DGRAD) {
    if (p.stride == 2 && force == 0) {
      // stride-2 data gradient: parity-grouped rows, only the taps that exist (a quarter of the MACs)
      for (int par = 0; par < 2; ++par) {
        // coordinates hd in [0,Hd) with (hd + pad) & 1 == par
        int h0 = ((par - p.pad) % 2 + 2) % 2, w0 = h0;
        p.par_h0[par] = h0; p.par_w0[par] = w0;
        p.par_hc[par] = p.Hd > h0 ? (p.Hd - h0 + 1) / 2 : 0;
        p.par_wc[par] = p.Wd > w0 ? (p.Wd - w0 + 1) / 2 : 0;
      }
      if (p.Ncols <= 64) return launch_cfg<128, 64, 4, 1, 2, true, true>(p, s);
      if (t64 >= thr_par64()) return launch_cfg<64, 128, 2, 2, 2, true, true>(p, s);
      return launch_cfg<64, 64, 2, 2, 3, true, true>(p, s);
    }
  }
  if (p.Ncols <= 64) return MXDET_LAUNCH_ST(128, 64, 4, 1, 2);
  if (t128 >= thr_t128() && K > 256) {
    // Largest layers: 256x256 tiles (one 8-wave workgroup per CU, half the LDS-DMA pieces per MFMA of the 128x128
    // tile) for as many whole rounds of the chip's 256 CUs as the layer has; the remaining rows -- a partial round
    // would leave most CUs idle for a whole tile time -- go to a second launch with 128x128 tiles.
    const int tn = ceil_div(p.Ncols, 256);
    const long long full = (long long)(p.M / 256) * tn;
    const long long rounds = full / 256;
    if (rounds >= 1 && p.Ncols % 256 == 0) {
      ConvP big = p;
      big.tiles_m = (int)(rounds * 256 / tn);
      int rc = launch_cfg<256, 256, 2, 4, 2, DGRAD>(big, s);
      if (rc) return rc;
      p.m_begin = big.tiles_m * 256;
      if (p.m_begin >= p.M) return rc;
    }
    // the remaining rows (a partial round of 256x256 tiles would idle most CUs for a whole tile time): small tiles,
    // so that every CU gets a share of the tail (measured: 64x64 tiles 208 workgroups, vs 128x128 tiles 52 workgroups)
    switch ((int)tuning(MXDET_TUNE_TAIL)) {
      case 1: return MXDET_LAUNCH_ST(64, 128, 2, 2, 2);
      case 2: return MXDET_LAUNCH_ST(64, 64, 2, 2, 3);
      default: return MXDET_LAUNCH_ST(128, 128, 2, 2, 2);
    }
  }
  // one partial round of 256 x 256 tiles that still covers most of the chip, with a long enough reduction to pay for the
  // tile's prologue / epilogue (the box head's first FC, data gradient: 1,024 x 12,544 x 1,024 -> 196 tiles, 69 -> 33 us)
  {
    const long long t256 = (long long)(p.M / 256) * (p.Ncols / 256);
    if (p.M % 256 == 0 && p.Ncols % 256 == 0 && t256 >= 160 && t256 <= 256 && K >= 1024 && p.m_begin == 0)
      return launch_cfg<256, 256, 2, 4, 2, DGRAD>(p, s);
  }
  // many rows, short reduction: 128 x 128 tiles of eight waves (64 x 32 per wave) -- half the LDS-DMA bytes per MFMA of
  // the 64 x 128 tile (the small tiles' K loop is bound by what a CU's vector-memory path takes in, ~64 B/clk)
  if (st != 0 && p.Ncols >= 128 && t128 >= tuning(MXDET_TUNE_T128W)) return MXDET_LAUNCH_ST(128, 128, 2, 4, 2);
  if (t64 >= thr_t64()) return MXDET_LAUNCH_ST(64, 128, 2, 2, 2);
  return MXDET_LAUNCH_ST(64, 64, 2, 2, 3);
#undef MXDET_LAUNCH_ST
}

static int validate(const mxdet_conv_desc_t* d, const char* who) {
  MXDET_REQUIRE(d != nullptr, MXDET_EINVAL, "%s: null descriptor", who);
  MXDET_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 &&
                    d->stride > 0 && d->pad >= 0,
                MXDET_ESHAPE, "%s: non-positive dimension", who);
  MXDET_REQUIRE(d->KH * d->KW <= 25, MXDET_ESHAPE, "%s: at most 25 filter taps (use the stem kernel for 7x7)", who);
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

#ifdef MXDET_CONV_STAMP
extern "C" int mxdet_debug_read_conv_stamps(unsigned long long* out /* host, 16 */) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_stamp), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -4;
}
#endif

extern "C" int mxdet_debug_force_conv_cfg(int32_t cfg) {
  g_force_cfg = cfg;
  return MXDET_OK;
}

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
  p.pf = d->prefetch; p.pf_bytes = d->prefetch ? d->prefetch_bytes : 0;
  p.bits_out = (unsigned char*)d->relu_bits;
  MXDET_REQUIRE(!d->relu_bits || d->Cout % 8 == 0, MXDET_ESHAPE, "conv2d_fwd: relu_bits needs Cout %% 8 == 0");
  return launch<false>(p, as_stream(stream));
}

// ---- chained 3x3 -> 1x1 forward (the tail of a frozen C2 bottleneck: conv2 + ReLU, conv3 + shortcut + ReLU) --------------
// One launch: the 3x3's 128-row x 64-channel tile never leaves the workgroup (accumulators -> bf16 A fragments, rounded as the
// unfused layer would store them), the 1x1's 256 x 64 filter sits in LDS, and the 1x1's epilogue (bias, residual, ReLU)
// writes the block output. Saves the intermediate map's write and read and one launch; results are bit-identical to the
// two launches (same reduction order in both convolutions). Optionally a THIRD convolution rides along: the next block's
// conv1 (1x1, 256 -> 64, + bias, ReLU) accumulated over the 64-column chunks of the block output as they are stored -- the
// block output is then not read back for it.
extern "C" int mxdet_conv2d_fwd_chain(const mxdet_conv_desc_t* d, const uint16_t* x, const uint16_t* w, const float* bias,
                                      const uint16_t* w2, const float* bias2, int32_t cout2, int32_t relu2,
                                      const uint16_t* residual2, uint16_t* y2, const uint16_t* w3, const float* bias3,
                                      int32_t cout3, int32_t relu3, uint16_t* y3, mxdet_stream_t stream) {
  clear_error();
  int rc = validate(d, "conv2d_fwd_chain");
  if (rc) return rc;
  MXDET_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, MXDET_ESHAPE,
                "conv2d_fwd_chain: the first convolution must be 3x3, stride 1, pad 1");
  MXDET_REQUIRE(d->Cin % 64 == 0 && d->Cout == 64 && cout2 == 256, MXDET_ESHAPE,
                "conv2d_fwd_chain: Cin %d must be a multiple of 64, Cout %d must be 64 and cout2 %d must be 256", d->Cin,
                d->Cout, cout2);
  MXDET_REQUIRE(!d->res_upsample && !d->relu_bits, MXDET_EINVAL, "conv2d_fwd_chain: no upsampled residual / relu_bits");
  MXDET_REQUIRE(x && w && w2 && y2, MXDET_EINVAL, "conv2d_fwd_chain: null pointer");
  MXDET_REQUIRE((long long)d->N * d->Ho * d->Wo * cout2 < (1ll << 31), MXDET_ESHAPE, "conv2d_fwd_chain: output exceeds 2^31");
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w; p.bias = bias; p.y = nullptr;
  p.N = d->N; p.Hs = d->H; p.Ws = d->W; p.C = d->Cin;
  p.Hd = d->Ho; p.Wd = d->Wo; p.Ncols = d->Cout;
  p.KH = 3; p.KW = 3; p.stride = 1; p.pad = 1;
  p.relu = d->relu;
  p.M = d->N * d->Ho * d->Wo;
  p.pf = d->prefetch; p.pf_bytes = d->prefetch ? d->prefetch_bytes : 0;
  p.chain_w = w2; p.chain_bias = bias2; p.chain_res = residual2; p.chain_y = y2; p.chain_relu = relu2;
  if (w3 != nullptr) {
    MXDET_REQUIRE(cout3 == 64 && y3 != nullptr, MXDET_ESHAPE, "conv2d_fwd_chain: the third convolution needs cout3 == 64 (got %d) and y3",
                  cout3);
    p.chain3_w = w3; p.chain3_bias = bias3; p.chain3_y = y3; p.chain3_relu = relu3;
  }
  return launch_cfg<128, 64, 4, 1, 2, false, false, 9, 256>(p, as_stream(stream));
}

// ---- split-K forward for long reductions on few rows (FC6: 1,024 rois x 12,544 features x 1,024 outputs) ----------------
// 64 x 64 tiles give such a layer 256 workgroups -- one wave per SIMD, 196 latency-bound steps each (118 us in the step, the
// chip mostly idle). With the reduction cut in `ksplit` ranges the grid is ksplit times larger; the raw fp32 tiles go to
// the caller's workspace and a fold kernel adds them in split order (deterministic) and applies bias / residual / ReLU.
namespace mxdet {
__global__ void __launch_bounds__(256)
conv_splitk_fold_kernel(const float* __restrict__ partial, int ksplit, long long M, int Ncols, const float* __restrict__ bias,
                        const uint16_t* __restrict__ res, int relu, uint16_t* __restrict__ y) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;     // element index, 8 channels per lane
  const long long total = M * Ncols;
  if (i >= total) return;
  const int col = (int)(i % Ncols);
  float v[8];
  {
    const float4 a = *(const float4*)(partial + i), b = *(const float4*)(partial + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  for (int k = 1; k < ksplit; ++k) {
    const float4 a = *(const float4*)(partial + (size_t)k * total + i), b = *(const float4*)(partial + (size_t)k * total + i + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  if (bias) {
    const float4 a = *(const float4*)(bias + col), b = *(const float4*)(bias + col + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  if (res) {
    const uint4 rv = *(const uint4*)(res + i);
    v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
    v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
    v[4] += __uint_as_float(rv.z << 16); v[5] += __uint_as_float(rv.z & 0xffff0000u);
    v[6] += __uint_as_float(rv.w << 16); v[7] += __uint_as_float(rv.w & 0xffff0000u);
  }
  if (relu) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
  }
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  *(uint4*)(y + i) = o;
}
}  // namespace mxdet

extern "C" size_t mxdet_conv2d_fwd_splitk_workspace_bytes(const mxdet_conv_desc_t* d, int32_t ksplit) {
  if (!d || ksplit < 1) return 0;
  return (size_t)ksplit * (size_t)d->N * d->Ho * d->Wo * d->Cout * sizeof(float);
}

extern "C" int mxdet_conv2d_fwd_splitk(const mxdet_conv_desc_t* d, const uint16_t* x, const uint16_t* w, const float* bias,
                                       const uint16_t* residual, uint16_t* y, int32_t ksplit, void* workspace,
                                       size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  int rc = validate(d, "conv2d_fwd_splitk");
  if (rc) return rc;
  MXDET_REQUIRE(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !d->res_upsample, MXDET_ESHAPE,
                "conv2d_fwd_splitk: 1x1 / stride 1 / pad 0 layers (fully connected) only");
  MXDET_REQUIRE(d->Cin % 64 == 0 && d->Cout % 8 == 0, MXDET_ESHAPE, "conv2d_fwd_splitk: Cin %% 64, Cout %% 8");
  MXDET_REQUIRE(ksplit >= 1 && ksplit <= d->Cin / 64, MXDET_ESHAPE, "conv2d_fwd_splitk: 1 <= ksplit <= Cin / 64");
  MXDET_REQUIRE(x && w && y, MXDET_EINVAL, "conv2d_fwd_splitk: null pointer");
  if (ksplit == 1) return mxdet_conv2d_fwd(d, x, w, bias, residual, y, stream);
  const size_t need = mxdet_conv2d_fwd_splitk_workspace_bytes(d, ksplit);
  MXDET_REQUIRE(workspace && workspace_bytes >= need, MXDET_EWORKSPACE, "conv2d_fwd_splitk: workspace %zu < %zu", workspace_bytes, need);
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w; p.bias = nullptr; p.res = nullptr; p.mask = nullptr; p.y = y;
  p.N = d->N; p.Hs = d->H; p.Ws = d->W; p.C = d->Cin;
  p.Hd = d->Ho; p.Wd = d->Wo; p.Ncols = d->Cout;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
  p.M = d->N * d->Ho * d->Wo;
  p.ksplit = ksplit; p.partial = (float*)workspace;
  // tile: 64 x 64 (three workgroups per CU), or 128 x 128 (half the L2 -> LDS bytes per flop: the K loop of the small
  // tile is bound by that path) when the layer still fills the chip with them (tuning key SPLITK_TILE: 0 / 1 / 2 = 128 x 128
  // tiles of four / eight waves)
  const int big = (int)tuning(MXDET_TUNE_SPLITK_TILE);
  const int BT = big ? 128 : 64;
  const long long tiles = (long long)ceil_div(p.M, BT) * ceil_div(p.Ncols, BT);
  MXDET_REQUIRE(tiles % 8 == 0, MXDET_ESHAPE, "conv2d_fwd_splitk: the tile count (%lld) must be a multiple of 8", tiles);
  hipStream_t s = as_stream(stream);
  if (big == 1) rc = launch_cfg<128, 128, 2, 2, 2, false, false, 1>(p, s);
  else if (big == 2) rc = launch_cfg<128, 128, 2, 4, 2, false, false, 1>(p, s);
  else rc = launch_cfg<64, 64, 2, 2, 3, false, false, 1>(p, s);
  if (rc) return rc;
  const long long total = (long long)p.M * p.Ncols;
  hipLaunchKernelGGL(conv_splitk_fold_kernel, dim3((unsigned)ceil_div<long long>(total / 8, 256)), dim3(256), 0, s,
                     (const float*)workspace, ksplit, (long long)p.M, p.Ncols, bias, residual, d->relu, y);
  return check_launch("conv2d_fwd_splitk");
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
  p.mask = (d->relu && !d->relu_bits) ? relu_mask : nullptr;
  p.bits_in = d->relu ? (const unsigned char*)d->relu_bits : nullptr;
  MXDET_REQUIRE(!d->relu || relu_mask || d->relu_bits, MXDET_EINVAL, "conv2d_dgrad: relu set without relu_mask / relu_bits");
  p.N = d->N; p.Hs = d->Ho; p.Ws = d->Wo; p.C = d->Cout;
  p.Hd = d->H; p.Wd = d->W; p.Ncols = d->Cin;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.relu = 0; p.res_up = 0;
  p.M = d->N * d->H * d->W;
  p.pf = d->prefetch; p.pf_bytes = d->prefetch ? d->prefetch_bytes : 0;
  return launch<true>(p, as_stream(stream));
}

// ---- grouped convolutions -------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NS, bool DGRAD>
static void launch_grouped_cfg(const ConvG* table, int n, int grid, hipStream_t s, int tapclass) {
  // tapclass: 0 = any geometry, 1 = every item a stride-1 1x1, 2 = every item a stride-1 3x3 (static-tap K loop)
  if (tapclass == 1)
    hipLaunchKernelGGL((conv_igemm_grouped_kernel<BM, BN, WM, WN, NS, DGRAD, 1>), dim3((unsigned)grid), dim3(64 * WM * WN),
                       0, s, table, n);
  else if (tapclass == 2)
    hipLaunchKernelGGL((conv_igemm_grouped_kernel<BM, BN, WM, WN, NS, DGRAD, 9>), dim3((unsigned)grid), dim3(64 * WM * WN),
                       0, s, table, n);
  else
    hipLaunchKernelGGL((conv_igemm_grouped_kernel<BM, BN, WM, WN, NS, DGRAD>), dim3((unsigned)grid), dim3(64 * WM * WN),
                       0, s, table, n);
}

static const int kGroupedTiles[4][2] = {{128, 64}, {128, 128}, {64, 128}, {64, 64}};   // cfg -> BM, BN

extern "C" size_t mxdet_conv2d_grouped_table_bytes(int32_t n) { return n > 0 ? (size_t)n * sizeof(ConvG) : 0; }

extern "C" int mxdet_conv2d_grouped_plan(const mxdet_conv_item_t* items, int32_t n, int32_t kind, void* table_host,
                                         size_t table_bytes, int32_t* cfg_out, int32_t* grid_out) {
  clear_error();
  MXDET_REQUIRE(items && n > 0 && table_host && cfg_out && grid_out, MXDET_EINVAL, "conv2d_grouped_plan: null pointer");
  MXDET_REQUIRE(kind == 0 || kind == 1, MXDET_EINVAL, "conv2d_grouped_plan: kind must be 0 (forward) or 1 (dgrad)");
  MXDET_REQUIRE(table_bytes >= (size_t)n * sizeof(ConvG), MXDET_EWORKSPACE, "conv2d_grouped_plan: table too small");
  ConvG* t = (ConvG*)table_host;
  long long t64 = 0, t128_max = 0;
  int max_cols = 0, kmax = 0;
  int tapclass = -1;          // -1: not decided, 0: mixed / strided
  for (int i = 0; i < n; ++i) {
    const mxdet_conv_desc_t* d = &items[i].desc;
    int rc = validate(d, "conv2d_grouped_plan");
    if (rc) return rc;
    ConvP& p = t[i].p;
    memset(&t[i], 0, sizeof(ConvG));
    MXDET_REQUIRE(items[i].src && items[i].filt && items[i].dst, MXDET_EINVAL, "conv2d_grouped_plan: item %d: null pointer", i);
    if (kind == 0) {
      MXDET_REQUIRE(d->Cin % 64 == 0 && d->Cout % 8 == 0, MXDET_ESHAPE, "conv2d_grouped_plan: item %d: Cin %% 64, Cout %% 8", i);
      p.x = (const uint16_t*)items[i].src; p.w = (const uint16_t*)items[i].filt; p.bias = items[i].bias;
      p.res = (const uint16_t*)items[i].residual; p.mask = nullptr; p.y = (uint16_t*)items[i].dst;
      p.bits_out = (unsigned char*)d->relu_bits;
      p.N = d->N; p.Hs = d->H; p.Ws = d->W; p.C = d->Cin;
      p.Hd = d->Ho; p.Wd = d->Wo; p.Ncols = d->Cout;
      p.relu = d->relu; p.res_up = d->res_upsample;
      p.M = d->N * d->Ho * d->Wo;
    } else {
      MXDET_REQUIRE(d->Cout % 64 == 0 && d->Cin % 8 == 0, MXDET_ESHAPE, "conv2d_grouped_plan: item %d: Cout %% 64, Cin %% 8", i);
      MXDET_REQUIRE(d->stride == 1, MXDET_ESHAPE, "conv2d_grouped_plan: item %d: strided data gradients are not grouped", i);
      MXDET_REQUIRE(!d->relu || items[i].relu_mask || d->relu_bits, MXDET_EINVAL, "conv2d_grouped_plan: item %d: relu without mask", i);
      p.x = (const uint16_t*)items[i].src; p.w = (const uint16_t*)items[i].filt; p.bias = nullptr;
      p.y = (uint16_t*)items[i].dst;
      p.res = items[i].residual ? (const uint16_t*)items[i].residual : (d->accumulate ? (const uint16_t*)items[i].dst : nullptr);
      p.mask = (d->relu && !d->relu_bits) ? (const uint16_t*)items[i].relu_mask : nullptr;
      p.bits_in = d->relu ? (const unsigned char*)d->relu_bits : nullptr;
      p.N = d->N; p.Hs = d->Ho; p.Ws = d->Wo; p.C = d->Cout;
      p.Hd = d->H; p.Wd = d->W; p.Ncols = d->Cin;
      p.relu = 0; p.res_up = 0;
      p.M = d->N * d->H * d->W;
    }
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    p.pf = d->prefetch; p.pf_bytes = d->prefetch ? d->prefetch_bytes : 0;
    {
      const int tc = (d->stride != 1 && (kind == 1 || d->KH * d->KW == 1)) ? 0
                     : (d->KH == 1 && d->KW == 1 && d->pad == 0) ? 1 : (d->KH == 3 && d->KW == 3 && d->pad == 1) ? 2 : 0;
      tapclass = tapclass < 0 ? tc : (tapclass == tc ? tc : 0);
    }
    t64 += (long long)ceil_div(p.M, 64) * ceil_div(p.Ncols, 128);
    long long t128 = (long long)ceil_div(p.M, 128) * ceil_div(p.Ncols, 128);
    t128_max = t128 > t128_max ? t128 : t128_max;
    max_cols = p.Ncols > max_cols ? p.Ncols : max_cols;
    kmax = p.KH * p.KW * p.C > kmax ? p.KH * p.KW * p.C : kmax;
  }
  // one tile configuration for the whole group, by the same rule as single launches (on the group's totals)
  int cfg;
  if (max_cols <= 64) cfg = 0;
  else if (t128_max >= thr_t128() && kmax > 256) cfg = 1;
  else if (t64 >= thr_t64()) cfg = 2;
  else cfg = 3;
  const int BM = kGroupedTiles[cfg][0], BN = kGroupedTiles[cfg][1];
  long long blocks = 0;
  for (int i = 0; i < n; ++i) {
    ConvP& p = t[i].p;
    p.tiles_m = ceil_div(p.M, BM);
    p.tiles_n = ceil_div(p.Ncols, BN);
    t[i].nblocks = p.tiles_m * p.tiles_n;
    t[i].block0 = (int)blocks;
    blocks += (long long)align_up((size_t)t[i].nblocks, 8);
    MXDET_REQUIRE(blocks < (1ll << 30), MXDET_ESHAPE, "conv2d_grouped_plan: group too large");
  }
  if (tuning(MXDET_TUNE_STATIC_TAPS) == 0 || tapclass < 0) tapclass = 0;
  *cfg_out = cfg + 4 * tapclass;          // tile configuration + 4 x tap class
  *grid_out = (int32_t)blocks;
  return MXDET_OK;
}

extern "C" int mxdet_conv2d_grouped(const void* table_dev, int32_t n, int32_t kind, int32_t cfg, int32_t grid,
                                    mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(table_dev && n > 0 && grid > 0, MXDET_EINVAL, "conv2d_grouped: empty group");
  MXDET_REQUIRE((kind == 0 || kind == 1) && cfg >= 0 && cfg <= 11, MXDET_EINVAL, "conv2d_grouped: bad kind / cfg");
  const ConvG* t = (const ConvG*)table_dev;
  hipStream_t s = as_stream(stream);
  const int tc = cfg >> 2;
  cfg &= 3;
  if (kind == 0) {
    switch (cfg) {
      case 0: launch_grouped_cfg<128, 64, 4, 1, 2, false>(t, n, grid, s, tc); break;
      case 1: launch_grouped_cfg<128, 128, 2, 2, 2, false>(t, n, grid, s, tc); break;
      case 2: launch_grouped_cfg<64, 128, 2, 2, 2, false>(t, n, grid, s, tc); break;
      default: launch_grouped_cfg<64, 64, 2, 2, 3, false>(t, n, grid, s, tc); break;
    }
  } else {
    switch (cfg) {
      case 0: launch_grouped_cfg<128, 64, 4, 1, 2, true>(t, n, grid, s, tc); break;
      case 1: launch_grouped_cfg<128, 128, 2, 2, 2, true>(t, n, grid, s, tc); break;
      case 2: launch_grouped_cfg<64, 128, 2, 2, 2, true>(t, n, grid, s, tc); break;
      default: launch_grouped_cfg<64, 64, 2, 2, 3, true>(t, n, grid, s, tc); break;
    }
  }
  return check_launch("conv2d_grouped");
}
