// LDS-DMA (global_load_lds_dwordx4) issue / throughput microbenchmark for gfx950. One workgroup per CU, W waves, each wave
// loops: issue G pieces of 1 KiB (L2-resident source), keep at most K in flight. Reports cycles per piece per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_rate.hip -o gpurun_out/dma_rate && gpurun_out/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int MODE, int INFLIGHT>
__global__ void __launch_bounds__(512) k(const unsigned char* __restrict__ src, size_t span, int iters, int nwaves_load, int nwaves_mfma,
                  unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[128 * 1024];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long t0 = 0, t1 = 0;
  if (wid < nwaves_load) {
    // each wave walks its own 1-KiB pieces; the footprint per workgroup is 64 KiB, shared by every workgroup (L2 hits)
    size_t off = ((size_t)wid * 8 * 1024 + (size_t)lane * 16);
    if (MODE == 1) off = ((size_t)wid * 8 * 1024 + (size_t)((lane ^ ((lane >> 3) & 7))) * 16);   // permuted inside the piece
    if (MODE == 2) off = ((size_t)wid * 8 * 1024 + (size_t)(lane >> 5) * 512 + (size_t)((((lane & 31) >> 1) ^ 5) * 32 + (lane & 1) * 16));
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)span, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(64u << 20), 0x00020000);
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (MODE >= 5 && MODE <= 9) {   // buffer form, STREAMING: groups of `share` workgroups walk the same 2-MiB region once
          const int share = MODE == 6 ? 1 : 9;
          // the workgroups of a group sit on ONE XCD (blocks b and b + 8 share an XCD): group = (xcd, index-in-xcd / share)
          const unsigned xcd = blockIdx.x & 7, inx = blockIdx.x >> 3;
          const unsigned grp = share == 1 ? blockIdx.x : xcd * 4 + inx / share;
          const unsigned base = (unsigned)((grp * (2u << 20)) % (60u << 20));
          // skew between the workgroups of a group: none (5), 1/9 of the region (7), 16 KiB (8), 64 KiB (9)
          const unsigned skew = MODE == 7 ? (inx % share) * ((2u << 20) / 9 / 1024) : MODE == 8 ? (inx % share) * 16u
                                : MODE == 9 ? (inx % share) * 64u : 0u;
          const unsigned q = (unsigned)((it * 8 + g) * nwaves_load + wid) + skew;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc2, (lptr_t)(lds + wid * 16384 + (g & 7) * 1024), 16,
                                                   (int)(base + (q * 1024u) % (2u << 20) + lane * 16), 0, 0, 0);
        } else if (MODE == 3) {          // buffer_load ... lds: SRD in SGPRs + 32-bit per-lane offset
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + wid * 16384 + (g & 7) * 1024), 16,
                                               (int)(off + g * 1024), 0, 0, 0);
        } else if (MODE == 4) {   // global_load_lds, saddr form: uniform 64-bit base + 32-bit per-lane offset
          const unsigned voff = (unsigned)(off + g * 1024);
          const unsigned ldsa = (unsigned)(size_t)(lptr_t)(lds + wid * 16384 + (g & 7) * 1024);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(ldsa) : "memory", "m0");
        } else {
          const unsigned char* p = src + ((off + (size_t)g * 1024 + (size_t)blockIdx.x * 0) % span);
          __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + wid * 16384 + (g & 7) * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
  } else if (wid < nwaves_load + nwaves_mfma) {
    __syncthreads();
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    f32x4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0, 0, 0, 0};
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters * 4; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) sink[0] = s;
  } else {
    __syncthreads();
  }
  if (lane == 0) out[blockIdx.x * 16 + wid] = t1 - t0;
}

template <int MODE, int INFLIGHT>
void run(const char* name, int nload, int nmfma, unsigned char* src, unsigned long long* out, float* sink) {
  const int iters = 200, grid = 256;
  hipMemset(out, 0, grid * 16 * 8);
  hipLaunchKernelGGL((k<MODE, INFLIGHT>), dim3(grid), dim3(512), 0, 0, src, (size_t)(64 * 1024), iters, nload, nmfma, out, sink);
  hipLaunchKernelGGL((k<MODE, INFLIGHT>), dim3(grid), dim3(512), 0, 0, src, (size_t)(64 * 1024), iters, nload, nmfma, out, sink);
  hipDeviceSynchronize();
  unsigned long long h[256 * 16];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double sl = 0, sm = 0;
  for (int b = 0; b < grid; ++b) {
    for (int w = 0; w < nload; ++w) sl += (double)h[b * 16 + w];
    for (int w = nload; w < nload + nmfma; ++w) sm += (double)h[b * 16 + w];
  }
  const double per_wave = nload ? sl / (grid * nload) : 0;                   // cycles a loader wave spent
  const double pieces_cu = (double)iters * 8 * nload;
  printf("%-34s loaders %d mfma-waves %d inflight %2d: %7.1f cyc/piece/wave  %6.1f cyc/piece/CU = %5.1f B/clk/CU", name, nload, nmfma,
         INFLIGHT, per_wave / (iters * 8), nload ? per_wave / pieces_cu : 0, nload ? 1024.0 * pieces_cu / per_wave : 0);
  if (nmfma) printf("   mfma %5.1f cyc each", sm / (grid * nmfma) / (iters * 4 * 8));
  printf("\n");
}

int main() {
  unsigned char* src; unsigned long long* out; float* sink;
  hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20);
  hipMalloc(&out, 256 * 16 * 8); hipMalloc(&sink, 64);
  run<0, 0>("linear, serial", 1, 0, src, out, sink);
  run<0, 7>("linear", 1, 0, src, out, sink);
  run<0, 7>("linear", 4, 0, src, out, sink);
  run<0, 7>("linear", 8, 0, src, out, sink);
  run<0, 3>("linear", 8, 0, src, out, sink);
  run<0, 1>("linear", 8, 0, src, out, sink);
  run<1, 7>("chunks permuted in 128-B lines", 8, 0, src, out, sink);
  run<2, 7>("2 rows x 512 B, 32-B granules xor", 8, 0, src, out, sink);
  run<2, 7>("2 rows x 512 B, 32-B granules xor", 4, 0, src, out, sink);
  run<3, 7>("buffer_load lds (SRD + voffset)", 1, 0, src, out, sink);
  run<3, 7>("buffer_load lds (SRD + voffset)", 4, 0, src, out, sink);
  run<3, 7>("buffer_load lds (SRD + voffset)", 8, 0, src, out, sink);
  run<4, 7>("global_load_lds saddr + voffset", 1, 0, src, out, sink);
  run<4, 7>("global_load_lds saddr + voffset", 4, 0, src, out, sink);
  run<4, 7>("global_load_lds saddr + voffset", 8, 0, src, out, sink);
  run<5, 7>("buffer, streaming, 9 WGs share", 4, 0, src, out, sink);
  run<5, 7>("buffer, streaming, 9 WGs share", 8, 0, src, out, sink);
  run<5, 7>("buffer, streaming, 9 share + mfma", 4, 4, src, out, sink);
  run<7, 7>("9 share, skew 1/9 region", 8, 0, src, out, sink);
  run<8, 7>("9 share, skew 16 KiB", 8, 0, src, out, sink);
  run<9, 7>("9 share, skew 64 KiB", 8, 0, src, out, sink);
  run<9, 7>("9 share, skew 64 KiB", 4, 0, src, out, sink);
  run<6, 7>("buffer, streaming, private regions", 4, 0, src, out, sink);
  run<6, 7>("buffer, streaming, private regions", 8, 0, src, out, sink);
  run<6, 3>("buffer, streaming, private regions", 8, 0, src, out, sink);
  run<3, 7>("buffer lds + mfma partners", 4, 4, src, out, sink);
  run<0, 7>("linear + mfma partners", 4, 4, src, out, sink);
  run<2, 7>("xor + mfma partners", 4, 4, src, out, sink);
  run<0, 7>("mfma only", 0, 4, src, out, sink);
  run<0, 7>("mfma only x8", 0, 8, src, out, sink);
  return 0;
}
