// The optimizer half of the training step on gfx950, over FLAT fp32 buffers (all parameters / gradients / moments of the
// model contiguous in HBM, fcaf3d_amd/flat.py): the global gradient norm with its clip coefficient, and one fused
// AdamW pass that applies the clip while it reads the gradient.  Bandwidth-bound: 4 B/param for the norm, 28 B/param for
// the step (p, g, m, v read; p, m, v written) — 70.4 M parameters -> 0.28 GB + 1.97 GB per step.
//
// Replaces, for the reference's recipe (configs/fcaf3d/fcaf3d.py:30-31: AdamW lr 1e-3 / weight decay 1e-4,
// grad_clip max_norm 10 / norm_type 2), mmcv's OptimizerHook.after_train_iter: torch.nn.utils.clip_grad_norm_ +
// torch.optim.AdamW.step — ~14 multi-tensor launches over 161 tensors and one extra read-modify-write of every gradient.
#include "fc_common.h"

typedef float of4 __attribute__((ext_vector_type(4)));

#define SQ_BLOCKS 1024

// part[b] = sum of g^2 over the block's float4 slices (grid-stride, fixed assignment: deterministic)
__global__ __launch_bounds__(256) void k_sqsum_partial(const of4* __restrict__ g, int64_t n4, double* __restrict__ part) {
  __shared__ double red[4];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    of4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = i + u * stride;
      v[u] = j < n4 ? g[j] : of4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s = v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
      acc += (double)s;
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = ||g||_2, out[1] = min(1, max_norm / (||g|| + 1e-6))  (clip_grad_norm_'s coefficient; max_norm <= 0: 1)
__global__ __launch_bounds__(1024) void k_sqsum_final(const double* __restrict__ part, int nb, float max_norm,
                                                      float* __restrict__ out) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 1024) acc += fc_ld(&part[i]);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    const float norm = (float)sqrt(t);
    out[0] = norm;
    float c = 1.f;
    if (max_norm > 0.f) c = fminf(max_norm / (norm + 1e-6f), 1.f);
    out[1] = c;
  }
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad), the arithmetic of torch's own kernels:
//   p *= 1 - lr * wd ; m += (g - m) * (1 - b1) ; v = b2 * v + (1 - b2) * g * g ;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)          with g already scaled by the clip coefficient
__global__ __launch_bounds__(256) void k_adamw(of4* __restrict__ p, const of4* __restrict__ g, of4* __restrict__ m,
                                               of4* __restrict__ v, int64_t n4, float lr, float b1, float b2, float eps,
                                               float wd, float step_size, float bc2_sqrt, const float* __restrict__ clip) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float c = clip ? fc_ld(&clip[1]) : 1.f;      // written by k_sqsum_final one launch ago, same address every step
  of4 pv = p[i], gv = g[i], mv = m[i], vv = v[i];
  const float decay = 1.f - lr * wd;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gg = gv[j] * c;
    float pp = pv[j] * decay;
    const float mm = mv[j] + (gg - mv[j]) * (1.f - b1);
    const float v2 = vv[j] * b2 + (1.f - b2) * gg * gg;
    const float denom = sqrtf(v2) / bc2_sqrt + eps;
    pp -= step_size * (mm / denom);
    pv[j] = pp; mv[j] = mm; vv[j] = v2;
  }
  p[i] = pv; m[i] = mv; v[i] = vv;
}

extern "C" {

int64_t fc_grad_norm_ws_bytes(int64_t n) { return (int64_t)SQ_BLOCKS * (int64_t)sizeof(double); }

int fc_grad_norm(const float* g, int64_t n, float max_norm, float* out, void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (n < 0 || n % 4 || !out) return FC_EINVAL;
  if (ws_bytes < fc_grad_norm_ws_bytes(n)) return FC_EWS;
  const int64_t n4 = n / 4;
  int nb = (int)fc_cdiv(n4 > 0 ? n4 : 1, 256 * 4);
  if (nb > SQ_BLOCKS) nb = SQ_BLOCKS;
  k_sqsum_partial<<<nb, 256, 0, stream>>>((const of4*)g, n4, (double*)ws);
  FC_CHECK_LAUNCH();
  k_sqsum_final<<<1, 1024, 0, stream>>>((const double*)ws, nb, max_norm, out);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float bias_correction1, float bias_correction2, const float* norm_and_clip,
                  hipStream_t stream) {
  if (n < 0 || n % 4 || bias_correction1 <= 0.f || bias_correction2 <= 0.f) return FC_EINVAL;
  if (n == 0) return FC_OK;
  const int64_t n4 = n / 4;
  k_adamw<<<(unsigned)fc_cdiv(n4, 256), 256, 0, stream>>>((of4*)p, (const of4*)g, (of4*)m, (of4*)v, n4, lr, beta1, beta2, eps,
                                                         weight_decay, lr / bias_correction1, sqrtf(bias_correction2),
                                                         norm_and_clip);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"
