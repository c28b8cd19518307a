// common.h -- shared host/device helpers for libmxdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mxdet.h"
#include "../../include/mxdet_debug.h"
#include "../../include/mxdet_math.h"

namespace mxdet {

// thread-local error string (the only mutable global state of the library)
void set_error(const char* fmt, ...);
void clear_error();

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MXDET_EHIP;
  }
  return MXDET_OK;
}

#define MXDET_REQUIRE(cond, code, ...)  \
  do {                                  \
    if (!(cond)) {                      \
      ::mxdet::set_error(__VA_ARGS__);  \
      return (code);                    \
    }                                   \
  } while (0)

// plan-time thresholds (mxdet_debug_set_tuning; defaults in capi.hip). The library never reads the environment.
long long tuning(int which);

static inline hipStream_t as_stream(mxdet_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;
// byte offset that is out of range for every buffer descriptor of the library (tensors are < 2^31 elements): an LDS-DMA
// lane given this offset writes 16 zero bytes into its LDS slot (measured, tools/micro/dma_oob.hip)
constexpr unsigned kDmaOob = 0xfffffff0u;

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Zero `bytes` (a multiple of 4) at `p` with a kernel. Used instead of hipMemsetAsync everywhere: memset NODES of a
// hipGraph did not order reliably against the kernels behind them once a graph started with them (a step captured
// with the RPN branch as its own graph lost the zeroing of the anchor-sampling counters: garbage counts, GPU fault).
__global__ static void zero_u32_kernel(unsigned* __restrict__ p, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t s) {
  long long n = (long long)((bytes + 3) / 4);
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (unsigned*)p, n);
  return hipGetLastError();
}

// ---- device helpers ----------------------------------------------------------------------------
// raw buffer descriptor (stride 0, `bytes` records) over a tensor, built from provably wave-uniform words so that hipcc
// keeps it in SGPRs (a descriptor it cannot prove uniform gets a waterfall loop around every buffer instruction)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) { return mxdet_f32_to_bf16(f); }
// two floats -> packed bf16 pair by the hardware conversion (v_cvt_pk_bf16_f32, round to nearest even: the same bits as
// mxdet_f32_to_bf16 for every finite value; NaNs keep being NaNs). One instruction instead of ~12 integer ones: used
// by the dense epilogues, whose results are tolerance-checked; the bit-exact detection ops keep the shared helper.
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t p = (bf16x2_t){(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, p);
}
// exact floor(n / d) and remainder for 0 <= n < 2^24 through one float multiply + a +-1 correction (an integer division
// is ~40 instructions): the tile prologues do two per row
__device__ __forceinline__ int fast_divmod(int n, int d, float rcp, int* rem) {
  int q = (int)((float)n * rcp);
  int r = n - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
  *rem = r;
  return q;
}

__device__ __forceinline__ float load_as_f32(const void* p, int64_t i, int dtype) {
  if (dtype == MXDET_DTYPE_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
  return ((const float*)p)[i];
}
__device__ __forceinline__ void store_from_f32(void* p, int64_t i, int dtype, float v) {
  if (dtype == MXDET_DTYPE_BF16)
    ((uint16_t*)p)[i] = f32_to_bf16_bits(v);
  else
    ((float*)p)[i] = v;
}

// wave-level inclusive/exclusive helpers (wave = 64)
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ int wave_excl_count(bool pred, int* total) {
  unsigned long long m = __ballot(pred);
  int lane = lane_id();
  unsigned long long below = (lane == 0) ? 0ull : (m & (~0ull >> (64 - lane)));
  *total = __popcll(m);
  return __popcll(below);
}

// Block-wide exclusive prefix count of a predicate in thread order; returns this thread's rank and
// the block total. `scratch` must hold (blockDim.x/64 + 1) ints. Contains __syncthreads.
__device__ inline int block_excl_count(bool pred, int* scratch, int* total) {
  int wtot;
  int r = wave_excl_count(pred, &wtot);
  int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();  // protect scratch reuse
  if (lane_id() == 0) scratch[wid] = wtot;
  __syncthreads();
  int base = 0, all = 0;
  for (int i = 0; i < nw; ++i) {
    int v = scratch[i];
    if (i < wid) base += v;
    all += v;
  }
  *total = all;
  return base + r;
}

}  // namespace mxdet
