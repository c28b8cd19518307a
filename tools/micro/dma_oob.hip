// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros into its LDS slot? (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const unsigned char* src, int bytes, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) ((unsigned*)lds)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // lanes 0..31 in range (offset lane*16), lanes 32..47: offset = bytes (first byte past the end), 48..63: 0xfffffff0
  unsigned voff = lane < 32 ? lane * 16 : (lane < 48 ? (unsigned)bytes + (lane - 32) * 16 : 0xfffffff0u);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, (int)voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = ((unsigned*)lds)[i];
}
int main() {
  unsigned char* src; unsigned* out;
  hipMalloc(&src, 4096); hipMemset(src, 0x11, 4096);
  hipMalloc(&out, 2048);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 1024, out);
  unsigned h[512];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane 0: %08x  lane 31: %08x  lane 32 (first past end): %08x  lane 47: %08x  lane 48 (huge): %08x  lane 63: %08x  slot 64 (untouched): %08x\n",
         h[0], h[31 * 4], h[32 * 4], h[47 * 4 + 3], h[48 * 4], h[63 * 4 + 3], h[64 * 4]);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { unsigned want = l < 32 ? 0x11111111u : 0u; if (h[l * 4 + j] != want) ++bad; }
  printf("mismatches vs (in-range: data, out-of-range: zeros): %d\n", bad);
  return 0;
}
