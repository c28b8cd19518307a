// mask.hip -- Mask R-CNN specific operators for gfx950: mask-target generation, 2x pixel shuffle (the data
// movement half of the 2x2 stride-2 deconvolution) and the per-pixel sigmoid BCE mask loss.
//
// Slots: core/mask (/root/reference/README.md:18) and models/mask_heads (README.md:30); MXNet roles: the lineage's
// mask-target CustomOp (numpy/cv2 crop + resize on the host), Deconvolution(kernel=2, stride=2) and a sigmoid BCE
// on the ground-truth class channel (README.md:37). The deconvolution's arithmetic runs on the MFMA conv kernel as a
// 1x1 convolution 256 -> 4*256 (one output group per (dy,dx)); this file only interleaves the four groups.
#include "common.h"

namespace mxdet {

// ---- mask targets: crop the matched instance bitmask to the roi, resample to SxS, threshold at 0.5 -------------
// target[r][py][px] = bilinear(mask_g, y = y1 + (py+0.5)*h/S - 0.5?, ...) >= 0.5 with the RoIAlign(aligned=False)
// geometry: sample point = roi start + (p + 0.5) * bin, clamped bilinear taps, out-of-image samples = 0.
__global__ void __launch_bounds__(256)
mask_target_kernel(const float* __restrict__ rois, const int32_t* __restrict__ matched_gt,
                   const int32_t* __restrict__ labels, const uint8_t* __restrict__ gt_masks, int R, int G_max,
                   int H, int W, int S, uint8_t* __restrict__ targets, int32_t* __restrict__ cls_out) {
  const int r = blockIdx.x;
  const float* q = rois + (long long)r * 5;
  const int n = (int)q[0];
  const int g = matched_gt[r];
  const int lab = labels[r];
  const bool fg = lab > 0 && g >= 0 && g < G_max;
  if (threadIdx.x == 0) cls_out[r] = fg ? lab : -1;
  const float x1 = q[1], y1 = q[2];
  float rw = q[3] - q[1], rh = q[4] - q[2];
  rw = rw > 1.0f ? rw : 1.0f;
  rh = rh > 1.0f ? rh : 1.0f;
  const float bw = rw / (float)S, bh = rh / (float)S;
  const uint8_t* m = gt_masks + ((long long)n * G_max + (fg ? g : 0)) * H * W;
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) {
    uint8_t t = 0;
    if (fg) {
      int py = i / S, px = i - py * S;
      float y = y1 + ((float)py + 0.5f) * bh;
      float x = x1 + ((float)px + 0.5f) * bw;
      if (!(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W)) {
        if (y <= 0.0f) y = 0.0f;
        if (x <= 0.0f) x = 0.0f;
        int yl = (int)y, xl = (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
        float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
        float v = hy * hx * (float)m[(long long)yl * W + xl];
        v = v + hy * lx * (float)m[(long long)yl * W + xh];
        v = v + ly * hx * (float)m[(long long)yh * W + xl];
        v = v + ly * lx * (float)m[(long long)yh * W + xh];
        t = v >= 0.5f ? 1 : 0;
      }
    }
    targets[(long long)r * S * S + i] = t;
  }
}

// ---- pixel shuffle: x [R,H,W,4*C] with channel = (dy*2+dx)*C + c  ->  y [R,2H,2W,C]; and its inverse ------------
__global__ void pixel_shuffle2_kernel(const uint4* __restrict__ x, int R, int H, int W, int C8, int inverse,
                                      uint4* __restrict__ y) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)R * H * W * 4 * C8;
  if (idx >= total) return;
  int c8 = (int)(idx % C8);
  long long t = idx / C8;
  int d = (int)(t & 3);
  t >>= 2;
  int w = (int)(t % W);
  t /= W;
  int h = (int)(t % H);
  int r = (int)(t / H);
  long long packed = idx;   // [r][h][w][d][c8]
  long long spread = ((((long long)r * 2 * H + (2 * h + (d >> 1))) * 2 * W) + (2 * w + (d & 1))) * C8 + c8;
  if (inverse) y[packed] = x[spread]; else y[spread] = x[packed];
}

// backward of (1x1 conv -> ReLU -> pixel shuffle): gather the upsampled gradient back into the packed layout and apply
// the ReLU mask of the packed activation in the same pass (the two separate passes moved the tensor twice more)
__global__ void pixel_shuffle2_inv_relu_kernel(const uint4* __restrict__ dy_up, const uint4* __restrict__ act, int R,
                                               int H, int W, int C8, uint4* __restrict__ dx) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)R * H * W * 4 * C8;
  if (idx >= total) return;
  int c8 = (int)(idx % C8);
  long long t = idx / C8;
  int d = (int)(t & 3);
  t >>= 2;
  int w = (int)(t % W);
  t /= W;
  int h = (int)(t % H);
  int r = (int)(t / H);
  long long spread = ((((long long)r * 2 * H + (2 * h + (d >> 1))) * 2 * W) + (2 * w + (d & 1))) * C8 + c8;
  const uint4 g = dy_up[spread], a = act[idx];
  const unsigned gg[4] = {g.x, g.y, g.z, g.w}, aa[4] = {a.x, a.y, a.z, a.w};
  unsigned o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {     // bf16 > 0  <=>  sign clear and magnitude non-zero
    const unsigned lo = aa[k] & 0xffffu, hi = aa[k] >> 16;
    o[k] = ((lo != 0u && lo < 0x8000u) ? (gg[k] & 0xffffu) : 0u) | ((hi != 0u && hi < 0x8000u) ? (gg[k] & 0xffff0000u) : 0u);
  }
  dx[idx] = make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- mask loss: sigmoid BCE on the ground-truth class channel, fused forward + backward -------------------------
// logits bf16 [R,S,S,Cpad]; cls[r] in 1..num_classes (or -1: ignored roi); targets u8 [R,S,S].
// grad (bf16, same shape) is written completely (zeros off the class channel). partial[blockIdx] = block loss sum.
__global__ void __launch_bounds__(256)
mask_loss_kernel(const uint16_t* __restrict__ logits, const int32_t* __restrict__ cls,
                 const uint8_t* __restrict__ targets, int R, int SS, int Cpad, const int* __restrict__ num_fg_dev,
                 float loss_scale, uint16_t* __restrict__ grad, float* __restrict__ partial) {
  __shared__ float red[8];
  // A workgroup owns 256 consecutive pixels (the partial-sum layout is unchanged) and walks their Cpad/8 16-byte
  // gradient chunks with consecutive lanes on consecutive chunks: one lane per pixel writing its whole 176-byte row
  // left every store instruction scattered over 64 rows.
  const long long pix0 = (long long)blockIdx.x * 256;
  const long long total = (long long)R * SS;
  const int nfg = *num_fg_dev;   // number of foreground rois, counted on device
  const float norm = 1.0f / (float)((nfg > 0 ? nfg : 1) * SS);
  const int CH = Cpad >> 3;
  float l = 0.0f;
  for (int it = threadIdx.x; it < 256 * CH; it += 256) {
    const int pl = it / CH, ck = it - pl * CH;
    const long long pix = pix0 + pl;
    if (pix >= total) break;
    const int r = (int)(pix / SS);
    const int c = cls[r];
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (c > 0 && c <= Cpad && ((c - 1) >> 3) == ck) {
      float z = bf16_bits_to_f32(logits[pix * Cpad + (c - 1)]);
      float t = (float)targets[pix];
      float az = z < 0.0f ? -z : z;
      float sp = mxdet_logf(1.0f + mxdet_expf(-az));
      l += (z > 0.0f ? z : 0.0f) - z * t + sp;
      float p = z >= 0.0f ? 1.0f / (1.0f + mxdet_expf(-z)) : mxdet_expf(z) / (1.0f + mxdet_expf(z));
      const unsigned gb = (unsigned)f32_to_bf16_bits((p - t) * norm * loss_scale) << (16 * ((c - 1) & 1));
      const int w = ((c - 1) & 7) >> 1;
      o.x = w == 0 ? gb : 0u; o.y = w == 1 ? gb : 0u; o.z = w == 2 ? gb : 0u; o.w = w == 3 ? gb : 0u;
    }
    *(uint4*)(grad + pix * Cpad + ck * 8) = o;
  }
  for (int off = 32; off > 0; off >>= 1) l += __shfl_down(l, off);
  int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wid] = l;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (((red[0] + red[1]) + red[2]) + red[3]) * norm;
}

__global__ void count_pos_kernel(const int32_t* __restrict__ cls, int R, int* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool p = i < R && cls[i] > 0;
  unsigned long long m = __ballot(p);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, __popcll(m));
}

__global__ void __launch_bounds__(256)
mask_finalize_kernel(const float* __restrict__ partial, int count, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < count; i += blockDim.x) acc += partial[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- mask paste-back (inference): class channel -> probability map -> bilinear resize into the detection box -> threshold
// prob[r][i] = sigmoid(logits[r, i, cls-1]) for the detection's class (rows without a valid class give zeros)
__global__ void __launch_bounds__(256)
mask_prob_kernel(const uint16_t* __restrict__ logits, const float* __restrict__ dets, int R, int SS, int Cpad,
                 float* __restrict__ prob) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)R * SS) return;
  const int r = (int)(i / SS);
  const int c = (int)dets[r * 6 + 5];
  float pv = 0.0f;
  if (c > 0 && c <= Cpad) {
    const float z = bf16_bits_to_f32(logits[i * Cpad + (c - 1)]);
    pv = z >= 0.0f ? 1.0f / (1.0f + mxdet_expf(-z)) : mxdet_expf(z) / (1.0f + mxdet_expf(z));
  }
  prob[i] = pv;
}

// source coordinate of destination index d when S samples are stretched over `len` pixels (OpenCV float bilinear:
// half-pixel centres, clamped at both ends)
__device__ __forceinline__ void paste_coef(int d, int len, int S, int& s0, int& s1, float& f) {
  float x = (float)(((double)d + 0.5) * ((double)S / (double)len) - 0.5);
  int s = (int)floorf(x);
  x -= (float)s;
  if (s < 0) { x = 0.0f; s = 0; }
  if (s >= S - 1) { x = 0.0f; s = S - 1; }
  s0 = s;
  s1 = s + 1 < S ? s + 1 : S - 1;
  f = x;
}

// one lane = 4 consecutive pixels of one row of one detection's full-frame mask (one 32-bit store)
__global__ void __launch_bounds__(256)
mask_paste_kernel(const float* __restrict__ prob, const float* __restrict__ dets, int S, int H, int W, float thresh,
                  uint8_t* __restrict__ out) {
  const int r = blockIdx.z, y = blockIdx.y;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= W) return;
  const float* d = dets + r * 6;
  // integer box, inclusive corners, as the lineage rounds it before resizing the mask into it
  const int bx1 = (int)rintf(d[0]), by1 = (int)rintf(d[1]), bx2 = (int)rintf(d[2]), by2 = (int)rintf(d[3]);
  const int bw = bx2 - bx1 + 1, bh = by2 - by1 + 1;
  unsigned wbits = 0u;
  if ((int)d[5] > 0 && bw > 0 && bh > 0 && y >= by1 && y <= by2) {
    int sy0, sy1;
    float fy;
    paste_coef(y - by1, bh, S, sy0, sy1, fy);
    const float* p = prob + (long long)r * S * S;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = x0 + j;
      if (x < bx1 || x > bx2) continue;
      int sx0, sx1;
      float fx;
      paste_coef(x - bx1, bw, S, sx0, sx1, fx);
      const float top = __fadd_rn(__fmul_rn(p[sy0 * S + sx0], __fsub_rn(1.0f, fx)), __fmul_rn(p[sy0 * S + sx1], fx));
      const float bot = __fadd_rn(__fmul_rn(p[sy1 * S + sx0], __fsub_rn(1.0f, fx)), __fmul_rn(p[sy1 * S + sx1], fx));
      const float v = __fadd_rn(__fmul_rn(top, __fsub_rn(1.0f, fy)), __fmul_rn(bot, fy));
      if (v > thresh) wbits |= 1u << (8 * j);
    }
  }
  *(unsigned*)(out + ((long long)r * H + y) * W + x0) = wbits;
}

}  // namespace mxdet

using namespace mxdet;

extern "C" int mxdet_mask_target(const float* rois, const int32_t* matched_gt, const int32_t* labels,
                                 const uint8_t* gt_masks, int64_t R, int32_t G_max, int32_t H, int32_t W,
                                 int32_t S, uint8_t* targets, int32_t* cls_out, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(R >= 0 && G_max > 0 && H > 0 && W > 0 && S > 0, MXDET_ESHAPE, "mask_target: bad shape");
  if (R == 0) return MXDET_OK;
  MXDET_REQUIRE(rois && matched_gt && labels && gt_masks && targets && cls_out, MXDET_EINVAL,
                "mask_target: null pointer");
  hipLaunchKernelGGL(mask_target_kernel, dim3((unsigned)R), dim3(256), 0, as_stream(stream), rois, matched_gt,
                     labels, gt_masks, (int)R, G_max, H, W, S, targets, cls_out);
  return check_launch("mask_target");
}

extern "C" int mxdet_pixel_shuffle2(const uint16_t* x, int64_t R, int32_t H, int32_t W, int32_t C,
                                    int32_t inverse, uint16_t* y, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(R > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MXDET_ESHAPE, "pixel_shuffle2: bad shape");
  MXDET_REQUIRE(x && y, MXDET_EINVAL, "pixel_shuffle2: null pointer");
  long long total = (long long)R * H * W * 4 * (C / 8);
  hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3((unsigned)ceil_div<long long>(total, 256)), dim3(256), 0,
                     as_stream(stream), (const uint4*)x, (int)R, H, W, C / 8, inverse, (uint4*)y);
  return check_launch("pixel_shuffle2");
}

extern "C" int mxdet_pixel_shuffle2_inv_relu(const uint16_t* dy_up, const uint16_t* act, int64_t R, int32_t H, int32_t W,
                                             int32_t C, uint16_t* dx, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(R > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MXDET_ESHAPE, "pixel_shuffle2_inv_relu: bad shape");
  MXDET_REQUIRE(dy_up && act && dx, MXDET_EINVAL, "pixel_shuffle2_inv_relu: null pointer");
  long long total = (long long)R * H * W * 4 * (C / 8);
  hipLaunchKernelGGL(pixel_shuffle2_inv_relu_kernel, dim3((unsigned)ceil_div<long long>(total, 256)), dim3(256), 0,
                     as_stream(stream), (const uint4*)dy_up, (const uint4*)act, (int)R, H, W, C / 8, (uint4*)dx);
  return check_launch("pixel_shuffle2_inv_relu");
}

extern "C" size_t mxdet_mask_loss_workspace_bytes(int64_t R, int32_t S) {
  long long blocks = ((long long)(R > 0 ? R : 0) * S * S + 255) / 256;
  return (size_t)blocks * sizeof(float) + 256;
}

extern "C" int mxdet_mask_loss(const uint16_t* logits, const int32_t* cls, const uint8_t* targets, int64_t R,
                               int32_t S, int32_t Cpad, float loss_scale, float* loss_out, uint16_t* grad,
                               void* workspace, size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(R > 0 && S > 0 && Cpad > 0 && Cpad % 8 == 0, MXDET_ESHAPE, "mask_loss: bad shape");
  MXDET_REQUIRE(logits && cls && targets && loss_out && grad, MXDET_EINVAL, "mask_loss: null pointer");
  MXDET_REQUIRE(workspace && workspace_bytes >= mxdet_mask_loss_workspace_bytes(R, S), MXDET_EWORKSPACE,
                "mask_loss: workspace too small");
  hipStream_t s = as_stream(stream);
  int* cnt = (int*)workspace;
  float* partial = (float*)((char*)workspace + 256);
  hipError_t e = zero_async(cnt, 256, s);
  MXDET_REQUIRE(e == hipSuccess, MXDET_EHIP, "mask_loss: memset failed");
  hipLaunchKernelGGL(count_pos_kernel, dim3((unsigned)ceil_div<long long>(R, 256)), dim3(256), 0, s, cls, (int)R, cnt);
  int blocks = (int)(((long long)R * S * S + 255) / 256);
  hipLaunchKernelGGL(mask_loss_kernel, dim3(blocks), dim3(256), 0, s, logits, cls, targets, (int)R, S * S, Cpad,
                     (const int*)cnt, loss_scale, grad, partial);
  hipLaunchKernelGGL(mask_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, blocks, loss_out);
  return check_launch("mask_loss");
}

extern "C" size_t mxdet_mask_paste_workspace_bytes(int64_t R, int32_t S) {
  return (size_t)(R > 0 ? R : 0) * S * S * sizeof(float) + 256;
}

extern "C" int mxdet_mask_paste(const uint16_t* logits, const float* dets, int64_t R, int32_t S, int32_t Cpad, int32_t H,
                                int32_t W, float thresh, uint8_t* masks, void* workspace, size_t workspace_bytes,
                                mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(R >= 0 && S > 0 && Cpad > 0 && H > 0 && W > 0 && W % 4 == 0, MXDET_ESHAPE,
                "mask_paste: bad shape (W must be a multiple of 4)");
  if (R == 0) return MXDET_OK;
  MXDET_REQUIRE(R <= 65535 && H <= 65535, MXDET_ESHAPE, "mask_paste: R=%lld or H=%d exceeds the grid", (long long)R, H);
  MXDET_REQUIRE(logits && dets && masks, MXDET_EINVAL, "mask_paste: null pointer");
  MXDET_REQUIRE(workspace && workspace_bytes >= mxdet_mask_paste_workspace_bytes(R, S), MXDET_EWORKSPACE,
                "mask_paste: workspace too small");
  hipStream_t s = as_stream(stream);
  float* prob = (float*)workspace;
  const long long n = (long long)R * S * S;
  hipLaunchKernelGGL(mask_prob_kernel, dim3((unsigned)ceil_div<long long>(n, 256)), dim3(256), 0, s, logits, dets, (int)R,
                     S * S, Cpad, prob);
  hipLaunchKernelGGL(mask_paste_kernel, dim3((unsigned)ceil_div(W, 1024), (unsigned)H, (unsigned)R), dim3(256), 0, s, prob,
                     dets, S, H, W, thresh, masks);
  return check_launch("mask_paste");
}
