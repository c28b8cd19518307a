// losses.hip -- fused forward+backward detection losses for gfx950.
//
// Slot: core/loss (/root/reference/README.md:19); MXNet roles smooth_l1(scalar=sigma), SoftmaxOutput
// (use_ignore, normalization='valid') and the RetinaNet sigmoid focal loss (README.md:37). Each loss
// reads its logits once and writes the gradient in the same pass (HBM-bound: 2 B read + 2 B written
// per bf16 logit). Scalar losses are reduced in a fixed order (per-block tree, then an index-ordered
// single-workgroup pass), so repeated runs give identical bits.
#include "common.h"

namespace mxdet {

// deterministic block sum: fixed shuffle tree inside each wave, then wave partials added in order
__device__ inline float block_sum_fixed(float v, float* smem /* >= blockDim/64 floats */) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane_id() == 0) smem[wid] = v;
  __syncthreads();
  float s = 0.0f;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) s += smem[i];
  return s;  // valid on thread 0
}

__device__ __forceinline__ float softplus_neg_abs(float z) {
  // log(1 + exp(-|z|))
  float az = z < 0.0f ? -z : z;
  return mxdet_logf(1.0f + mxdet_expf(-az));
}
// both at once from ONE exp(-|z|): bit-identical to the two functions above (they evaluate the same exponential)
__device__ __forceinline__ void sigmoid_softplus(float z, float& p, float& sp) {
  const float az = z < 0.0f ? -z : z;
  const float t = mxdet_expf(-az);
  const float d = 1.0f + t;
  sp = mxdet_logf(d);
  p = z >= 0.0f ? 1.0f / d : t / d;
}
__device__ __forceinline__ float sigmoidf_det(float z) {
  // 1/(1+exp(-z)) evaluated on the stable side
  if (z >= 0.0f) return 1.0f / (1.0f + mxdet_expf(-z));
  float e = mxdet_expf(z);
  return e / (1.0f + e);
}

// ---------------------------------------------------------------------------------------------
__global__ void smooth_l1_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                     const float* __restrict__ w, long long n, float sigma2,
                                     float* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = p[i] - t[i];
  if (w) d = d * w[i];
  out[i] = mxdet_smooth_l1(d, sigma2);
}
__global__ void smooth_l1_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                     const float* __restrict__ w, const float* __restrict__ go,
                                     long long n, float sigma2, int accumulate, float* __restrict__ gp) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ww = w ? w[i] : 1.0f;
  float d = (p[i] - t[i]) * ww;
  float g = mxdet_smooth_l1_grad(d, sigma2) * ww;
  if (go) g = g * go[i];
  gp[i] = accumulate ? gp[i] + g : g;
}

// ---------------------------------------------------------------------------------------------
// focal loss
__global__ void count_fg_kernel(const int32_t* __restrict__ labels, long long n, int* __restrict__ cnt) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool fg = (i < n) && labels[i] > 0;
  unsigned long long m = __ballot(fg);
  if (lane_id() == 0 && m) atomicAdd(cnt, __popcll(m));
}

__global__ void __launch_bounds__(256)
focal_kernel(const void* __restrict__ logits, int dtype, const int32_t* __restrict__ labels,
             long long n, int C, float alpha, float gamma, float grad_scale,
             const int* __restrict__ num_fg, void* __restrict__ grad, float* __restrict__ partial) {
  __shared__ float red[8];
  const long long total = n * C;
  int nf = *num_fg;
  float inv_norm = 1.0f / (float)(nf > 1 ? nf : 1);
  float acc = 0.0f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i / C;
    int c = (int)(i - r * C);
    int lab = labels[r];
    float z = load_as_f32(logits, i, dtype);
    float g = 0.0f;
    if (lab >= 0) {
      float p, sp;
      sigmoid_softplus(z, p, sp);
      // log(p) = -(max(-z,0) + sp); log(1-p) = -(max(z,0) + sp)
      float logp = -((z < 0.0f ? -z : 0.0f) + sp);
      float log1mp = -((z > 0.0f ? z : 0.0f) + sp);
      if (lab == c + 1) {
        float q = 1.0f - p;
        float mod = (gamma == 2.0f) ? q * q : mxdet_expf(gamma * mxdet_logf(q > 1e-30f ? q : 1e-30f));
        acc += -alpha * mod * logp;
        g = -alpha * mod * (q - gamma * p * logp);
      } else {
        float mod = (gamma == 2.0f) ? p * p : mxdet_expf(gamma * mxdet_logf(p > 1e-30f ? p : 1e-30f));
        acc += -(1.0f - alpha) * mod * log1mp;
        g = (1.0f - alpha) * mod * (p - gamma * (1.0f - p) * log1mp);
      }
    }
    store_from_f32(grad, i, dtype, g * inv_norm * grad_scale);
  }
  float s = block_sum_fixed(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s * inv_norm;
}

// out[j] = sum_i partial[i*ncomp + j], index-ordered, one workgroup
__global__ void __launch_bounds__(256)
finalize_kernel(const float* __restrict__ partial, int count, int ncomp, float* __restrict__ out) {
  __shared__ float red[8];
  for (int j = 0; j < ncomp; ++j) {
    float acc = 0.0f;
    for (int i = threadIdx.x; i < count; i += blockDim.x) acc += partial[(long long)i * ncomp + j];
    float s = block_sum_fixed(acc, red);
    if (threadIdx.x == 0) out[j] = s;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// RPN losses over one level's head output [N,H,W,Cpad]; one thread per cell.
__global__ void __launch_bounds__(256)
rpn_loss_kernel(const uint16_t* __restrict__ head, int N, int H, int W, int A, int Cpad,
                const int32_t* __restrict__ labels, const float4* __restrict__ targets,
                long long A_total, long long level_offset, float sigma2, float norm, float loss_scale,
                uint16_t* __restrict__ grad, float* __restrict__ partial) {
  __shared__ float red[8];
  long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long cells = (long long)N * H * W;
  float lc = 0.0f, lr = 0.0f;
  if (cell < cells) {
    int n = (int)(cell / ((long long)H * W));
    long long local = cell - (long long)n * H * W;
    const uint16_t* h = head + cell * Cpad;
    uint16_t* g = grad + cell * Cpad;
    for (int c = 5 * A; c < Cpad; ++c) g[c] = 0;
    for (int a = 0; a < A; ++a) {
      long long gi = (long long)n * A_total + level_offset + local * A + a;
      int lab = labels[gi];
      float gz = 0.0f;
      if (lab >= 0) {
        float z = bf16_bits_to_f32(h[a]);
        float sp = softplus_neg_abs(z);
        float l = (z > 0.0f ? z : 0.0f) - z * (float)lab + sp;
        lc += l;
        gz = (sigmoidf_det(z) - (float)lab) * norm * loss_scale;
      }
      g[a] = f32_to_bf16_bits(gz);
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lab == 1) t = targets[gi];
      float tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gd = 0.0f;
        if (lab == 1) {
          float d = bf16_bits_to_f32(h[A + 4 * a + k]) - tt[k];
          lr += mxdet_smooth_l1(d, sigma2);
          gd = mxdet_smooth_l1_grad(d, sigma2) * norm * loss_scale;
        }
        g[A + 4 * a + k] = f32_to_bf16_bits(gd);
      }
    }
  }
  float s0 = block_sum_fixed(lc, red);
  float s1 = block_sum_fixed(lr, red);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x + 0] = s0 * norm;
    partial[2 * blockIdx.x + 1] = s1 * norm;
  }
}

// ---------------------------------------------------------------------------------------------
// Box-head losses: one wave per roi.
__global__ void __launch_bounds__(256)
rcnn_loss_kernel(const void* __restrict__ cls, const void* __restrict__ reg, int dtype, int ld_cls,
                 int ld_reg, const int32_t* __restrict__ labels, const float* __restrict__ tgt,
                 const float* __restrict__ wgt, long long R, int num_classes, int reg_dim,
                 float sigma2, float norm, float loss_scale, void* __restrict__ gcls,
                 void* __restrict__ greg, float* __restrict__ partial) {
  long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = lane_id();
  int lab = labels[r];
  // softmax CE
  float mx = -3.0e38f;
  for (int c = lane; c < num_classes; c += 64) {
    float z = load_as_f32(cls, r * ld_cls + c, dtype);
    mx = z > mx ? z : mx;
  }
  for (int off = 32; off > 0; off >>= 1) {
    float o = __shfl_xor(mx, off);
    mx = o > mx ? o : mx;
  }
  float se = 0.0f;
  for (int c = lane; c < num_classes; c += 64)
    se += mxdet_expf(load_as_f32(cls, r * ld_cls + c, dtype) - mx);
  for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off);
  float lse = mxdet_logf(se);
  float lcls = 0.0f;
  for (int c = lane; c < num_classes; c += 64) {
    float z = load_as_f32(cls, r * ld_cls + c, dtype);
    float g = 0.0f;
    if (lab >= 0) {
      float p = mxdet_expf(z - mx) / se;
      g = (p - (c == lab ? 1.0f : 0.0f)) * norm * loss_scale;
      if (c == lab) lcls = -(z - mx - lse);
    }
    store_from_f32(gcls, r * ld_cls + c, dtype, g);
  }
  for (int off = 32; off > 0; off >>= 1) lcls += __shfl_xor(lcls, off);
  // smooth-L1
  float lreg = 0.0f;
  for (int c = lane; c < reg_dim; c += 64) {
    float w = wgt[r * reg_dim + c];
    float g = 0.0f;
    if (w != 0.0f && lab > 0) {
      float d = (load_as_f32(reg, r * ld_reg + c, dtype) - tgt[r * reg_dim + c]) * w;
      lreg += mxdet_smooth_l1(d, sigma2);
      g = mxdet_smooth_l1_grad(d, sigma2) * w * norm * loss_scale;
    }
    store_from_f32(greg, r * ld_reg + c, dtype, g);
  }
  // fixed xor tree: every lane ends with the same value
  for (int off = 32; off > 0; off >>= 1) lreg += __shfl_xor(lreg, off);
  if (lane == 0) {
    partial[2 * r + 0] = lcls * norm;
    partial[2 * r + 1] = lreg * norm;
  }
}

}  // namespace mxdet

using namespace mxdet;

extern "C" int mxdet_smooth_l1_fwd(const float* pred, const float* target, const float* weight,
                                   int64_t n, float sigma, float* out, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n >= 0 && sigma > 0.0f, MXDET_ESHAPE, "smooth_l1_fwd: bad arguments");
  if (n == 0) return MXDET_OK;
  MXDET_REQUIRE(pred && target && out, MXDET_EINVAL, "smooth_l1_fwd: null pointer");
  hipLaunchKernelGGL(smooth_l1_fwd_kernel, dim3((unsigned)ceil_div<int64_t>(n, 256)), dim3(256), 0,
                     as_stream(stream), pred, target, weight, (long long)n, sigma * sigma, out);
  return check_launch("smooth_l1_fwd");
}

extern "C" int mxdet_smooth_l1_bwd(const float* pred, const float* target, const float* weight,
                                   const float* grad_out, int64_t n, float sigma, int32_t accumulate,
                                   float* grad_pred, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n >= 0 && sigma > 0.0f, MXDET_ESHAPE, "smooth_l1_bwd: bad arguments");
  if (n == 0) return MXDET_OK;
  MXDET_REQUIRE(pred && target && grad_pred, MXDET_EINVAL, "smooth_l1_bwd: null pointer");
  hipLaunchKernelGGL(smooth_l1_bwd_kernel, dim3((unsigned)ceil_div<int64_t>(n, 256)), dim3(256), 0,
                     as_stream(stream), pred, target, weight, grad_out, (long long)n, sigma * sigma,
                     accumulate, grad_pred);
  return check_launch("smooth_l1_bwd");
}

extern "C" size_t mxdet_loss_workspace_bytes(int64_t n) {
  size_t a = (size_t)(n > 0 ? n : 0) * 2 * sizeof(float);
  size_t b = 4096 * sizeof(float);
  return (a > b ? a : b) + 256;
}

extern "C" int mxdet_focal_loss(const void* logits, int32_t dtype, const int32_t* labels, int64_t n,
                                int32_t C, float alpha, float gamma, float grad_scale,
                                float* loss_out, void* grad_logits, void* workspace,
                                size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(n > 0 && C > 0, MXDET_ESHAPE, "focal_loss: bad shape");
  MXDET_REQUIRE(dtype == MXDET_DTYPE_F32 || dtype == MXDET_DTYPE_BF16, MXDET_EINVAL, "focal_loss: dtype");
  MXDET_REQUIRE(logits && labels && loss_out && grad_logits, MXDET_EINVAL, "focal_loss: null pointer");
  MXDET_REQUIRE(workspace && workspace_bytes >= mxdet_loss_workspace_bytes(n), MXDET_EWORKSPACE,
                "focal_loss: workspace too small");
  hipStream_t s = as_stream(stream);
  int* cnt = (int*)workspace;
  float* partial = (float*)((char*)workspace + 256);
  hipError_t e = zero_async(cnt, 256, s);
  MXDET_REQUIRE(e == hipSuccess, MXDET_EHIP, "focal_loss: memset failed");
  hipLaunchKernelGGL(count_fg_kernel, dim3((unsigned)ceil_div<int64_t>(n, 256)), dim3(256), 0, s,
                     labels, (long long)n, cnt);
  long long total = (long long)n * C;
  int blocks = (int)(ceil_div<long long>(total, 256) < 2048 ? ceil_div<long long>(total, 256) : 2048);
  hipLaunchKernelGGL(focal_kernel, dim3(blocks), dim3(256), 0, s, logits, dtype, labels, (long long)n,
                     C, alpha, gamma, grad_scale, cnt, grad_logits, partial);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, partial, blocks, 1, loss_out);
  return check_launch("focal_loss");
}

extern "C" int32_t mxdet_rpn_loss_num_partials(int32_t N, int32_t H, int32_t W) {
  long long cells = (long long)N * H * W;
  return (int32_t)((cells + 255) / 256);
}

extern "C" int mxdet_rpn_loss_level(const uint16_t* head, int32_t N, int32_t H, int32_t W, int32_t A,
                                    int32_t Cpad, const int32_t* labels, const float* bbox_targets,
                                    int64_t A_total, int64_t level_offset, float sigma, float norm,
                                    float loss_scale, uint16_t* grad_head, float* partial,
                                    mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && H > 0 && W > 0 && A > 0 && Cpad >= 5 * A, MXDET_ESHAPE,
                "rpn_loss_level: bad shape (Cpad must be >= 5*A)");
  MXDET_REQUIRE(head && labels && bbox_targets && grad_head && partial, MXDET_EINVAL,
                "rpn_loss_level: null pointer");
  MXDET_REQUIRE(level_offset >= 0 && level_offset + (int64_t)H * W * A <= A_total, MXDET_ESHAPE,
                "rpn_loss_level: level outside the anchor range");
  int blocks = mxdet_rpn_loss_num_partials(N, H, W);
  hipLaunchKernelGGL(rpn_loss_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), head, N, H, W, A,
                     Cpad, labels, (const float4*)bbox_targets, (long long)A_total,
                     (long long)level_offset, sigma * sigma, norm, loss_scale, grad_head, partial);
  return check_launch("rpn_loss_level");
}

extern "C" int mxdet_loss_finalize(const float* partial, int32_t count, int32_t ncomp, float* out,
                                   mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(count >= 0 && ncomp > 0 && ncomp <= 8, MXDET_ESHAPE, "loss_finalize: bad arguments");
  MXDET_REQUIRE(partial && out, MXDET_EINVAL, "loss_finalize: null pointer");
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, count, ncomp,
                     out);
  return check_launch("loss_finalize");
}

extern "C" int mxdet_rcnn_loss(const void* cls_logits, const void* bbox_pred, int32_t dtype,
                               int32_t ld_cls, int32_t ld_reg, const int32_t* labels,
                               const float* bbox_targets, const float* bbox_weights, int64_t R,
                               int32_t num_classes, int32_t reg_dim, float sigma, float norm,
                               float loss_scale, float* loss_out, void* grad_cls, void* grad_reg,
                               void* workspace, size_t workspace_bytes, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(R > 0 && num_classes > 0 && reg_dim > 0 && ld_cls >= num_classes && ld_reg >= reg_dim,
                MXDET_ESHAPE, "rcnn_loss: bad shape");
  MXDET_REQUIRE(dtype == MXDET_DTYPE_F32 || dtype == MXDET_DTYPE_BF16, MXDET_EINVAL, "rcnn_loss: dtype");
  MXDET_REQUIRE(cls_logits && bbox_pred && labels && bbox_targets && bbox_weights && loss_out &&
                    grad_cls && grad_reg,
                MXDET_EINVAL, "rcnn_loss: null pointer");
  MXDET_REQUIRE(workspace && workspace_bytes >= mxdet_loss_workspace_bytes(R), MXDET_EWORKSPACE,
                "rcnn_loss: workspace too small");
  float* partial = (float*)((char*)workspace + 256);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(rcnn_loss_kernel, dim3((unsigned)ceil_div<int64_t>(R, 4)), dim3(256), 0, s,
                     cls_logits, bbox_pred, dtype, ld_cls, ld_reg, labels, bbox_targets, bbox_weights,
                     (long long)R, num_classes, reg_dim, sigma * sigma, norm, loss_scale, grad_cls,
                     grad_reg, partial);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, partial, (int)R, 2, loss_out);
  return check_launch("rcnn_loss");
}

// ------------------------------------------------------------------------------------------------
// RetinaNet (BASELINE.json config 5): dense-anchor class labels and the fused focal + box loss of one level.
namespace mxdet {

// cls_label = class of the matched GT for foreground anchors, else the {-1, 0} label itself
__global__ void anchor_class_labels_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ matched,
                                           const float* __restrict__ gt, long long A_total, int G_max, long long total,
                                           int32_t* __restrict__ cls_labels, int* __restrict__ num_fg) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int lab = -1;
  bool fg = false;
  if (i < total) {
    lab = labels[i];
    if (lab == 1) {
      int n = (int)(i / A_total);
      lab = (int)gt[((long long)n * G_max + matched[i]) * 5 + 4];
      fg = true;
    }
    cls_labels[i] = lab;
  }
  unsigned long long m = __ballot(fg);
  if (lane_id() == 0 && m) atomicAdd(num_fg, __popcll(m));
}

// one thread per (cell, anchor): C contiguous class logits at cls[cell*ld_cls + a*C], 4 deltas at reg[cell*ld_reg + a*4]
// Vectorised form (C % 8 == 0): a workgroup still owns 256 consecutive anchors (so the partial-sum layout is unchanged),
// but the class logits are walked 16 bytes per lane with consecutive lanes on consecutive addresses -- the scalar form
// below has every lane striding C*2 bytes apart with 2-byte accesses (920 us per step on RetinaNet-R101 for 129 MB).
__global__ void __launch_bounds__(256)
retina_loss_vec_kernel(const uint16_t* __restrict__ cls, const uint16_t* __restrict__ reg, int N, int HW, int A, int C,
                       int ld_cls, int ld_reg, const int32_t* __restrict__ cls_labels,
                       const float4* __restrict__ targets, long long A_total, long long level_offset, float alpha,
                       float gamma, float sigma2, const int* __restrict__ num_fg, float loss_scale,
                       uint16_t* __restrict__ gcls, uint16_t* __restrict__ greg, float* __restrict__ partial) {
  __shared__ float red[8];
  const long long total = (long long)N * HW * A;
  const long long a_first = (long long)blockIdx.x * 256;
  const int nf = *num_fg;
  const float inv = 1.0f / (float)(nf > 1 ? nf : 1);
  const int CH = C >> 3;
  float lc = 0.0f, lr = 0.0f;
  for (int it = threadIdx.x; it < 256 * CH; it += 256) {
    const int al = it / CH, ck = it - al * CH;
    const long long idx = a_first + al;
    if (idx >= total) continue;
    const int a = (int)(idx % A);
    const long long cell = idx / A;
    const int n = (int)(cell / HW);
    const long long local = cell - (long long)n * HW;
    const int lab = cls_labels[(long long)n * A_total + level_offset + local * A + a];
    const long long off = cell * ld_cls + (long long)a * C + ck * 8;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (lab >= 0) {
      const uint4 zv = *(const uint4*)(cls + off);
      const unsigned zw[4] = {zv.x, zv.y, zv.z, zv.w};
      unsigned ow[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned packed = 0u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = ck * 8 + 2 * k + h;
          const float x = h ? __uint_as_float(zw[k] & 0xffff0000u) : __uint_as_float(zw[k] << 16);
          float p, sp;
          sigmoid_softplus(x, p, sp);
          const float logp = -((x < 0.0f ? -x : 0.0f) + sp);
          const float log1mp = -((x > 0.0f ? x : 0.0f) + sp);
          float g;
          if (lab == c + 1) {
            const float q = 1.0f - p;
            const float mod = (gamma == 2.0f) ? q * q : mxdet_expf(gamma * mxdet_logf(q > 1e-30f ? q : 1e-30f));
            lc += -alpha * mod * logp;
            g = -alpha * mod * (q - gamma * p * logp);
          } else {
            const float mod = (gamma == 2.0f) ? p * p : mxdet_expf(gamma * mxdet_logf(p > 1e-30f ? p : 1e-30f));
            lc += -(1.0f - alpha) * mod * log1mp;
            g = (1.0f - alpha) * mod * (p - gamma * (1.0f - p) * log1mp);
          }
          packed |= (unsigned)f32_to_bf16_bits(g * inv * loss_scale) << (16 * h);
        }
        ow[k] = packed;
      }
      o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
    *(uint4*)(gcls + off) = o;
  }
  {   // box part: one lane per anchor, 8 bytes in, 8 bytes out
    const long long idx = a_first + threadIdx.x;
    if (idx < total) {
      const int a = (int)(idx % A);
      const long long cell = idx / A;
      const int n = (int)(cell / HW);
      const long long local = cell - (long long)n * HW;
      const long long gi = (long long)n * A_total + level_offset + local * A + a;
      const int lab = cls_labels[gi];
      const uint16_t* d = reg + cell * ld_reg + a * 4;
      uint16_t* gd = greg + cell * ld_reg + a * 4;
      uint2 go = make_uint2(0u, 0u);
      if (lab > 0) {
        const float4 t = targets[gi];
        const uint2 dv = *(const uint2*)d;
        const float dd[4] = {__uint_as_float(dv.x << 16), __uint_as_float(dv.x & 0xffff0000u),
                             __uint_as_float(dv.y << 16), __uint_as_float(dv.y & 0xffff0000u)};
        const float tt[4] = {t.x, t.y, t.z, t.w};
        unsigned short gb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float e = dd[k] - tt[k];
          lr += mxdet_smooth_l1(e, sigma2);
          gb[k] = f32_to_bf16_bits(mxdet_smooth_l1_grad(e, sigma2) * inv * loss_scale);
        }
        go = make_uint2((unsigned)gb[0] | ((unsigned)gb[1] << 16), (unsigned)gb[2] | ((unsigned)gb[3] << 16));
      }
      *(uint2*)gd = go;
    }
  }
  float s0 = block_sum_fixed(lc, red);
  float s1 = block_sum_fixed(lr, red);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x + 0] = s0 * inv;
    partial[2 * blockIdx.x + 1] = s1 * inv;
  }
}

__global__ void __launch_bounds__(256)
retina_loss_kernel(const uint16_t* __restrict__ cls, const uint16_t* __restrict__ reg, int N, int HW, int A, int C,
                   int ld_cls, int ld_reg, const int32_t* __restrict__ cls_labels, const float4* __restrict__ targets,
                   long long A_total, long long level_offset, float alpha, float gamma, float sigma2,
                   const int* __restrict__ num_fg, float loss_scale, uint16_t* __restrict__ gcls,
                   uint16_t* __restrict__ greg, float* __restrict__ partial) {
  __shared__ float red[8];
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * HW * A;
  const int nf = *num_fg;
  const float inv = 1.0f / (float)(nf > 1 ? nf : 1);
  float lc = 0.0f, lr = 0.0f;
  if (idx < total) {
    const int a = (int)(idx % A);
    const long long cell = idx / A;                 // n*HW + local cell
    const int n = (int)(cell / HW);
    const long long local = cell - (long long)n * HW;
    const long long gi = (long long)n * A_total + level_offset + local * A + a;
    const int lab = cls_labels[gi];
    const uint16_t* z = cls + cell * ld_cls + (long long)a * C;
    uint16_t* gz = gcls + cell * ld_cls + (long long)a * C;
    for (int c = 0; c < C; ++c) {
      float g = 0.0f;
      if (lab >= 0) {
        float x = bf16_bits_to_f32(z[c]);
        float p, sp;
        sigmoid_softplus(x, p, sp);
        float logp = -((x < 0.0f ? -x : 0.0f) + sp);
        float log1mp = -((x > 0.0f ? x : 0.0f) + sp);
        if (lab == c + 1) {
          float q = 1.0f - p;
          float mod = (gamma == 2.0f) ? q * q : mxdet_expf(gamma * mxdet_logf(q > 1e-30f ? q : 1e-30f));
          lc += -alpha * mod * logp;
          g = -alpha * mod * (q - gamma * p * logp);
        } else {
          float mod = (gamma == 2.0f) ? p * p : mxdet_expf(gamma * mxdet_logf(p > 1e-30f ? p : 1e-30f));
          lc += -(1.0f - alpha) * mod * log1mp;
          g = (1.0f - alpha) * mod * (p - gamma * (1.0f - p) * log1mp);
        }
      }
      gz[c] = f32_to_bf16_bits(g * inv * loss_scale);
    }
    const uint16_t* d = reg + cell * ld_reg + a * 4;
    uint16_t* gd = greg + cell * ld_reg + a * 4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lab > 0) t = targets[gi];
    const float tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float g = 0.0f;
      if (lab > 0) {
        float e = bf16_bits_to_f32(d[k]) - tt[k];
        lr += mxdet_smooth_l1(e, sigma2);
        g = mxdet_smooth_l1_grad(e, sigma2) * inv * loss_scale;
      }
      gd[k] = f32_to_bf16_bits(g);
    }
  }
  float s0 = block_sum_fixed(lc, red);
  float s1 = block_sum_fixed(lr, red);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x + 0] = s0 * inv;
    partial[2 * blockIdx.x + 1] = s1 * inv;
  }
}

}  // namespace mxdet

extern "C" int mxdet_anchor_class_labels(const int32_t* labels, const int32_t* matched_gt, const float* gt_boxes,
                                         int32_t N, int64_t A_total, int32_t G_max, int32_t* cls_labels,
                                         int32_t* num_fg, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && A_total > 0 && G_max > 0, MXDET_ESHAPE, "anchor_class_labels: bad shape");
  MXDET_REQUIRE(labels && matched_gt && gt_boxes && cls_labels && num_fg, MXDET_EINVAL, "anchor_class_labels: null pointer");
  hipStream_t s = as_stream(stream);
  hipError_t e = zero_async(num_fg, sizeof(int32_t), s);
  MXDET_REQUIRE(e == hipSuccess, MXDET_EHIP, "anchor_class_labels: memset failed");
  long long total = (long long)N * A_total;
  hipLaunchKernelGGL(anchor_class_labels_kernel, dim3((unsigned)ceil_div<long long>(total, 256)), dim3(256), 0, s, labels,
                     matched_gt, gt_boxes, (long long)A_total, G_max, total, cls_labels, num_fg);
  return check_launch("anchor_class_labels");
}

extern "C" int32_t mxdet_retina_loss_num_partials(int32_t N, int32_t H, int32_t W, int32_t A) {
  return (int32_t)(((long long)N * H * W * A + 255) / 256);
}

extern "C" int mxdet_retina_loss_level(const uint16_t* cls, const uint16_t* reg, int32_t N, int32_t H, int32_t W,
                                       int32_t A, int32_t C, int32_t ld_cls, int32_t ld_reg,
                                       const int32_t* cls_labels, const float* bbox_targets, int64_t A_total,
                                       int64_t level_offset, float alpha, float gamma, float sigma,
                                       const int32_t* num_fg, float loss_scale, uint16_t* grad_cls,
                                       uint16_t* grad_reg, float* partial, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(N > 0 && H > 0 && W > 0 && A > 0 && C > 0 && ld_cls >= A * C && ld_reg >= 4 * A, MXDET_ESHAPE,
                "retina_loss_level: bad shape");
  MXDET_REQUIRE(cls && reg && cls_labels && bbox_targets && num_fg && grad_cls && grad_reg && partial, MXDET_EINVAL,
                "retina_loss_level: null pointer");
  MXDET_REQUIRE(level_offset >= 0 && level_offset + (int64_t)H * W * A <= A_total, MXDET_ESHAPE,
                "retina_loss_level: level outside the anchor range");
  int blocks = mxdet_retina_loss_num_partials(N, H, W, A);
  // 16-byte path: whole 8-channel chunks per anchor and 16-B / 8-B aligned rows
  const bool vec = (C % 8 == 0) && (ld_cls % 8 == 0) && (ld_reg % 4 == 0) && (((uintptr_t)cls | (uintptr_t)grad_cls) % 16 == 0) &&
                   (((uintptr_t)reg | (uintptr_t)grad_reg) % 8 == 0);
  if (vec)
    hipLaunchKernelGGL(retina_loss_vec_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), cls, reg, N, H * W, A, C,
                       ld_cls, ld_reg, cls_labels, (const float4*)bbox_targets, (long long)A_total, (long long)level_offset,
                       alpha, gamma, sigma * sigma, (const int*)num_fg, loss_scale, grad_cls, grad_reg, partial);
  else
    hipLaunchKernelGGL(retina_loss_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), cls, reg, N, H * W, A, C, ld_cls,
                       ld_reg, cls_labels, (const float4*)bbox_targets, (long long)A_total, (long long)level_offset, alpha,
                       gamma, sigma * sigma, (const int*)num_fg, loss_scale, grad_cls, grad_reg, partial);
  return check_launch("retina_loss_level");
}
