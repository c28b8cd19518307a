// conv1x1_fs.h -- filter-stationary kernel for 1x1 / stride-1 convolutions with a SHORT reduction (K <= 512) and many
// output columns: every bottleneck's expand layer forward (conv3: Cmid -> 4 Cmid, + shortcut, ReLU, 1-bit mask) and the
// data gradient of its reduce layer (conv1: Cmid -> 4 Cmid, + shortcut gradient, masked), forward and data gradient of
// the FPN laterals. Included by conv.hip (uses ConvP).
//
// Why a second kernel: these layers are streaming kernels with a thin slice of arithmetic -- the C3 expand layer moves
// 80 MB for 4.4 GFLOP -- and the general tile (conv_igemm_tile) spends 2-8 K-steps between a prologue and an LDS-staged
// epilogue per 64 x 128 tile: every workgroup pays launch -> first loads -> epilogue operand loads -> stores as a chain
// of exposed latencies (measured: 7.9 of the 13.4 us of the C4 expand layer remain with the K loop removed). Here
//  * the filter slice of a wave (32 or 64 columns x K) lives in REGISTERS for the workgroup's life, loaded once straight
//    from L2 in MFMA-operand order -- LDS carries activations only;
//  * a workgroup is persistent over row tiles: while tile t is multiplied the LDS-DMA of tile t+1 and the residual /
//    mask loads of tile t+1 are in flight (one exposed latency per workgroup, not per tile);
//  * the MFMA operands are SWAPPED (filter rows = MFMA rows, pixels = MFMA columns) and the filter rows of an n-block
//    are permuted so that a lane ends up with 8 (NB = 2) or 16 (NB = 4) CONSECUTIVE channels of one pixel in its
//    accumulators: bias / residual / mask / ReLU / 1-bit mask happen in registers and the result leaves as 16-byte
//    stores (64 or 128 contiguous bytes per pixel across the wave) -- no LDS round trip.
// Same products, same summation order (32-deep k blocks, ascending) as the general kernel: bit-identical results.
#pragma once

namespace mxdet {

template <int NB, int K, int BM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv1x1_fs_kernel(ConvP p, int ngroups, int tiles_m) {
  constexpr int KS = K / 32;              // 32-deep MFMA steps
  constexpr int NSL = K / 64;             // 64-channel slices of the LDS image
  constexpr int MB = BM / 16;             // 16-pixel m-blocks per tile
  constexpr int STAGE = BM * K * 2;       // bytes per ring stage
  constexpr int PIECES = NSL * (BM / 8);  // 1-KiB LDS-DMA pieces per stage (8 rows x 128 B)
  constexpr int PPW = PIECES / 4;
  constexpr int WCOLS = 16 * NB;          // columns per wave
  static_assert(PIECES % 4 == 0 && K % 64 == 0 && BM % 16 == 0, "tile shape");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order: the workgroups of one row chunk (they read the same activation tiles) sit next to each other on one XCD
  int bid = (int)blockIdx.x;
  {
    const int nwg = (int)gridDim.x;
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cg = bid % ngroups, rc = bid / ngroups;
  const int P = (int)gridDim.x / ngroups;          // row chunks (the host launches ngroups * P workgroups)
  const int n0 = cg * (4 * WCOLS) + wid * WCOLS;   // first column of this wave
  const int fq = lane >> 4, fr = lane & 15;

  // ---- the wave's filter slice, in MFMA A-operand order with the rows of n-block j permuted: MFMA row i = 4q + r holds
  // column n0 + q * (4 NB) + 4 j + r, so that D row group q of all n-blocks is one run of 4 NB consecutive columns
  bf16x8_t wf[NB][KS];
  {
    const int q2 = fr >> 2, r2 = fr & 3;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int c = n0 + q2 * (4 * NB) + 4 * j + r2;
      c = c < p.Ncols ? c : p.Ncols - 1;           // (the host only routes Ncols % (64 NB) == 0 here)
      const uint16_t* wr = p.w + (size_t)c * K + fq * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[j][ks] = *(const bf16x8_t*)(wr + ks * 32);
    }
  }
  const int c0 = n0 + fq * (4 * NB);               // first of this lane's 4 NB output columns
  float bv[4 * NB];
#pragma unroll
  for (int k = 0; k < 4 * NB; ++k) bv[k] = p.bias ? p.bias[c0 + k] : 0.0f;

  // ---- LDS-DMA geometry: piece pi = wid * PPW + i covers slice pi / (BM/8), rows 8 * (pi % (BM/8)) .. +7
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x, 2u * (unsigned)p.M * (unsigned)K);
  unsigned dma_off[PPW];                           // byte offset of this lane's 16 bytes inside tile 0
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pi = wid * PPW + i;
    const int sl = pi / (BM / 8), rb = pi % (BM / 8);
    const int row = rb * 8 + (lane >> 3), slot = lane & 7;
    dma_off[i] = 2u * (unsigned)(row * K + sl * 64 + ((slot ^ ((row >> 1) & 7)) << 3));
  }
  auto issue_tile = [&](int t, int buf) {
    const unsigned base = 2u * (unsigned)t * (unsigned)(BM * K);      // rows past M: out of range -> zeros
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)(smem + buf * STAGE + (wid * PPW + i) * 1024), 16,
                                               (int)(base + dma_off[i]), 0, 0, 0);
  };
  // epilogue operands of a tile, one tile ahead: residual (4 NB bf16 per pixel and lane) and the 1-bit mask
  uint4 rres[MB][NB / 2];
  unsigned rbit[MB];
  auto fetch_operands = [&](int t) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = t * BM + mb * 16 + fr;
      const size_t o = (size_t)(m < p.M ? m : 0) * p.Ncols + c0;
      if (p.res) {
#pragma unroll
        for (int h = 0; h < NB / 2; ++h) rres[mb][h] = *(const uint4*)(p.res + o + h * 8);
      }
      if (p.bits_in) {
        if constexpr (NB == 2) rbit[mb] = p.bits_in[o >> 3];
        else rbit[mb] = *(const unsigned short*)(p.bits_in + (o >> 3));
      }
    }
  };

  // element offsets of the B-operand (activation) fragments: pixel row mb * 16 + fr, 16-byte chunk (ks & 1) * 4 + fq of slice ks >> 1
  const uint16_t* lds = (const uint16_t*)smem;
  int t = rc;
  if (t < tiles_m) {
    issue_tile(t, 0);
    fetch_operands(t);
  }
  int buf = 0;
  for (; t < tiles_m; t += P, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile t and its epilogue operands have landed (this wave's part)
    __builtin_amdgcn_s_barrier();                        // ... everyone's; everyone is done reading the other buffer
    asm volatile("" ::: "memory");
    // keep the landed operands of THIS tile (the compiler must place its own wait for them here, where nothing newer is
    // outstanding), then put the next tile in flight
    uint4 cres[MB][NB / 2];
    unsigned cbit[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int h = 0; h < NB / 2; ++h) {
        cres[mb][h] = rres[mb][h];
        asm volatile("" : "+v"(cres[mb][h].x), "+v"(cres[mb][h].y), "+v"(cres[mb][h].z), "+v"(cres[mb][h].w));
      }
      cbit[mb] = rbit[mb];
      asm volatile("" : "+v"(cbit[mb]));
    }
    if (t + P < tiles_m) {
      issue_tile(t + P, buf ^ 1);
      fetch_operands(t + P);
    }
    const uint16_t* st = lds + buf * (STAGE / 2);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int row = mb * 16 + fr;
      bf16x8_t af[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        af[ks] = *(const bf16x8_t*)(st + (ks >> 1) * (BM * 64) + row * 64 + ((((ks & 1) * 4 + fq) ^ ((row >> 1) & 7)) << 3));
      f32x4_t acc[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], af[ks], acc[j], 0, 0, 0);
      // ---- epilogue in registers: D row 4 fq + r of n-block j = column c0 + 4 j + r, D column fr = pixel row
      const int m = t * BM + row;
      float v[4 * NB];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[j][r] + bv[4 * j + r];
      if (p.res) {
#pragma unroll
        for (int h = 0; h < NB / 2; ++h) {
          const uint4 rv = cres[mb][h];
          v[8 * h + 0] += __uint_as_float(rv.x << 16); v[8 * h + 1] += __uint_as_float(rv.x & 0xffff0000u);
          v[8 * h + 2] += __uint_as_float(rv.y << 16); v[8 * h + 3] += __uint_as_float(rv.y & 0xffff0000u);
          v[8 * h + 4] += __uint_as_float(rv.z << 16); v[8 * h + 5] += __uint_as_float(rv.z & 0xffff0000u);
          v[8 * h + 6] += __uint_as_float(rv.w << 16); v[8 * h + 7] += __uint_as_float(rv.w & 0xffff0000u);
        }
      }
      if (p.bits_in) {
        const unsigned mbits = cbit[mb];
#pragma unroll
        for (int k = 0; k < 4 * NB; ++k)
          if (!((mbits >> k) & 1u)) v[k] = 0.0f;
      } else if (p.relu) {
#pragma unroll
        for (int k = 0; k < 4 * NB; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
      }
      if (m < p.M) {
        unsigned mb_out = 0;
#pragma unroll
        for (int h = 0; h < NB / 2; ++h) {
          uint4 o;
          o.x = pack_bf16x2(v[8 * h + 0], v[8 * h + 1]);
          o.y = pack_bf16x2(v[8 * h + 2], v[8 * h + 3]);
          o.z = pack_bf16x2(v[8 * h + 4], v[8 * h + 5]);
          o.w = pack_bf16x2(v[8 * h + 6], v[8 * h + 7]);
          *(uint4*)(p.y + (size_t)m * p.Ncols + c0 + 8 * h) = o;
          if (p.bits_out) {
            // the mask of the STORED values: bf16 > 0 <=> sign clear and magnitude non-zero
            const unsigned w4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const unsigned lo = w4[k] & 0xffffu, hi = w4[k] >> 16;
              mb_out |= ((lo != 0u && lo < 0x8000u) ? 1u : 0u) << (8 * h + 2 * k);
              mb_out |= ((hi != 0u && hi < 0x8000u) ? 1u : 0u) << (8 * h + 2 * k + 1);
            }
          }
        }
        if (p.bits_out) {
          const size_t ob = ((size_t)m * p.Ncols + c0) >> 3;
          if constexpr (NB == 2) p.bits_out[ob] = (unsigned char)mb_out;
          else *(unsigned short*)(p.bits_out + ob) = (unsigned short)mb_out;
        }
      }
    }
  }
}

// 0 = not routed here. The caller guarantees a 1x1 / stride 1 / pad 0 layer (static tap path).
static bool fs1x1_ok(const ConvP& p) {
  if (tuning(MXDET_TUNE_FS1X1) == 0) return false;
  if (p.res_up || p.ksplit > 1 || p.mask != nullptr || p.m_begin != 0) return false;
  if (p.C != 128 && p.C != 256 && p.C != 512) return false;
  const int wg_cols = p.C == 128 ? 256 : 128;
  if (p.Ncols < 256 || p.Ncols % wg_cols != 0) return false;
  if (p.M < 1024) return false;
  return true;
}

template <int NB, int K, int BM>
static int launch_fs1x1_cfg(ConvP& p, hipStream_t s) {
  const int ngroups = p.Ncols / (64 * NB);
  const int tiles_m = ceil_div(p.M, BM);
  // two workgroups per CU resident (registers); every workgroup walks tiles_m / P row tiles
  int P = 512 / ngroups;
  P = P < 1 ? 1 : (P > tiles_m ? tiles_m : P);
  hipLaunchKernelGGL((conv1x1_fs_kernel<NB, K, BM>), dim3((unsigned)(ngroups * P)), dim3(256), 0, s, p, ngroups, tiles_m);
  return check_launch("conv2d (1x1 filter-stationary)");
}

static int launch_fs1x1(ConvP& p, hipStream_t s) {
  if (p.C == 128) return launch_fs1x1_cfg<4, 128, 64>(p, s);
  if (p.C == 256) return launch_fs1x1_cfg<2, 256, 64>(p, s);
  return launch_fs1x1_cfg<2, 512, 32>(p, s);
}

}  // namespace mxdet
