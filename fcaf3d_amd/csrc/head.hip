// Head epilogue of Fcaf3DNeckWithHead.forward_single (fcaf3d_neck_with_head.py:256-279), fused.
// The three 1x1 convolutions run as ONE GEMM into y (N, ld) = [centerness | reg (n_reg) | cls (n_cls) | pad]; this file
// turns y into the reference's three outputs (+ the per-row max class logit the pruning step interpolates) in one
// pass, and their gradients back into gy in one pass (instead of 3 zero-filled slice gradients + 2 adds per level).
// One wave64 per row: lane = column.
#include "fc_common.h"

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void k_head_split_fwd(const float* __restrict__ y, int ld, const float* __restrict__ bias,
                                                        const float* __restrict__ scale, int64_t n, int n_reg, int n_cls,
                                                        float* __restrict__ centerness, float* __restrict__ bbox,
                                                        float* __restrict__ cls, float* __restrict__ cls_max) {
  const int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float v = lane < ld ? y[row * ld + lane] : 0.f;
  const float sc = scale[0];
  float m = -INFINITY;
  if (lane == 0) {
    centerness[row] = v;
  } else if (lane <= n_reg) {
    int j = lane - 1;
    bbox[row * n_reg + j] = j < 6 ? expf(v * sc) : v;
  } else if (lane <= n_reg + n_cls) {
    int c = lane - 1 - n_reg;
    m = v + bias[c];
    cls[row * n_cls + c] = m;
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) cls_max[row] = m;
}

// gy[row] = [g_cent | g_bbox[:6] * bbox[:6] * scale , g_bbox[6:] | g_cls | 0...];  gscale_row[row] = sum_j<6 g_bbox*bbox*reg
__global__ __launch_bounds__(256) void k_head_split_bwd(const float* __restrict__ y, int ld, const float* __restrict__ scale,
                                                        const float* __restrict__ bbox, const float* __restrict__ g_cent,
                                                        const float* __restrict__ g_bbox, const float* __restrict__ g_cls,
                                                        int64_t n, int n_reg, int n_cls, float* __restrict__ gy,
                                                        float* __restrict__ gscale_row) {
  const int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float sc = scale[0];
  float g = 0.f, s = 0.f;
  if (lane == 0) {
    g = g_cent ? g_cent[row] : 0.f;
  } else if (lane <= n_reg) {
    int j = lane - 1;
    float gb = g_bbox ? g_bbox[row * n_reg + j] : 0.f;
    if (j < 6) {
      float e = gb * bbox[row * n_reg + j];          // d exp(reg*scale) = exp(.) * (scale dreg + reg dscale)
      g = e * sc;
      s = e * y[row * ld + lane];
    } else {
      g = gb;
    }
  } else if (lane <= n_reg + n_cls) {
    g = g_cls ? g_cls[row * n_cls + (lane - 1 - n_reg)] : 0.f;
  }
  if (lane < ld) gy[row * ld + lane] = g;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) gscale_row[row] = s;
}

extern "C" {

int fc_head_split_fwd(const float* y, int ld, const float* bias, const float* scale_dev, int64_t n, int n_reg, int n_cls,
                      float* centerness, float* bbox_pred, float* cls_score, float* cls_max, hipStream_t stream) {
  if (n < 0 || ld < 1 || ld > 64 || n_reg < 6 || n_cls < 1 || 1 + n_reg + n_cls > ld) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_head_split_fwd<<<(unsigned)fc_cdiv(n, 4), 256, 0, stream>>>(y, ld, bias, scale_dev, n, n_reg, n_cls, centerness,
                                                                bbox_pred, cls_score, cls_max);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_head_split_bwd(const float* y, int ld, const float* scale_dev, const float* bbox_pred, const float* g_centerness,
                      const float* g_bbox, const float* g_cls, int64_t n, int n_reg, int n_cls, float* gy,
                      float* gscale_row, hipStream_t stream) {
  if (n < 0 || ld < 1 || ld > 64 || n_reg < 6 || n_cls < 1 || 1 + n_reg + n_cls > ld) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_head_split_bwd<<<(unsigned)fc_cdiv(n, 4), 256, 0, stream>>>(y, ld, scale_dev, bbox_pred, g_centerness, g_bbox, g_cls,
                                                                n, n_reg, n_cls, gy, gscale_row);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"
