// Loss kernels of the FCAF3D head on gfx950 (elementwise / per-box, latency- and HBM-bound).
//   * sigmoid focal loss  — mmcv `sigmoid_focal_loss` CUDA op reached through mmdet FocalLoss
//     (fcaf3d_neck_with_head.py:29-34, :180); label -1 (background) => every class is a negative.
//   * axis-aligned 3D IoU + its gradient — iou3d_loss.py:21-35 (axis_aligned_iou_loss) over
//     iou3d_calculator.py:201-330 (axis_aligned_bbox_overlaps_3d, is_aligned=True, eps=1e-6).
//   * rotated 3D IoU + its gradient — rotated_iou/oriented_iou_loss.py:86-109 (cal_iou_3d) with
//     box_intersection_2d.py:13-184 and the un-vendored `sort_v`, as ONE
//     fused kernel: the gradient is propagated in forward mode (7 dual parts per value) through the
//     same arithmetic graph torch autograd differentiates in the reference.
#include "fc_common.h"
// exact products as in the reference's torch / host code (e.g. num == 0 for parallel edges): no FMA contraction
#pragma clang fp contract(off)
#include <float.h>

// Angular sort key of sort_v ("sign of y, then |x| x / (x^2 + y^2)", eps 1e-8): monotone in
// atan2(y, x) over (-pi, pi], built from +, *, / only — with contraction off it is bit-identical to the numpy restatement
// (oracle/loss_oracle.py::sort_key), so the index order is EXACT across machines (r1/r2 used atan2f: two libms ordered
// ~0.5 % of near-tied vertices differently).
__device__ static inline float sort_key(float y, float x) {
  const float r = x * fabsf(x) / (x * x + y * y + 1e-8f);
  return y < 0.f ? r - 3.f : 1.f - r;
}

// ---------------------------------------------------------------------------------------------------
// t^gamma; gamma == 2 (every FCAF3D config, fcaf3d_neck_with_head.py:29-34) is the plain product — what torch's pow
// does for an exponent of 2 as well — instead of the ~60-instruction general powf
__device__ static inline float pow_gamma(float t, float gamma) { return gamma == 2.f ? t * t : powf(t, gamma); }

__device__ static inline float focal_elem(float x, bool is_pos, float gamma, float alpha) {
  float p = 1.f / (1.f + expf(-x));
  if (is_pos) return -alpha * pow_gamma(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN));
  return -(1.f - alpha) * pow_gamma(p, gamma) * logf(fmaxf(1.f - p, FLT_MIN));
}

// one thread per row: loss_rows[r] = w[r] * sum_c focal(x[r,c])   (class order fixed -> deterministic)
__global__ void k_focal_fwd(const float* __restrict__ x, const long long* __restrict__ label, const float* __restrict__ w,
                            int64_t n, int C, float gamma, float alpha, float* __restrict__ loss_rows) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  long long y = label[r];
  float acc = 0.f;
  for (int c = 0; c < C; ++c) acc += focal_elem(x[r * C + c], y == c, gamma, alpha);
  loss_rows[r] = w ? acc * w[r] : acc;
}

__global__ void k_focal_bwd(const float* __restrict__ x, const long long* __restrict__ label, const float* __restrict__ w,
                            int64_t n, int C, float gamma, float alpha, const float* __restrict__ gscale,
                            float* __restrict__ gx) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * C) return;
  int64_t r = t / C;
  int c = (int)(t % C);
  long long y = label[r];
  float p = 1.f / (1.f + expf(-x[t]));
  float g;
  if (y == c) g = -alpha * pow_gamma(1.f - p, gamma) * (1.f - p - gamma * p * logf(fmaxf(p, FLT_MIN)));
  else g = -(1.f - alpha) * pow_gamma(p, gamma) * (gamma * (1.f - p) * logf(fmaxf(1.f - p, FLT_MIN)) - p);
  gx[t] = g * gscale[0] * (w ? w[r] : 1.f);
}

// ---------------------------------------------------------------------------------------------------
// torch.max / torch.min split the gradient evenly on ties; clamp(min=0) passes it at x >= 0.
__device__ static inline void tie_max(float a, float b, float* v, float* wa) {
  *v = a > b ? a : b;
  *wa = a > b ? 1.f : (a == b ? 0.5f : 0.f);
}
__device__ static inline void tie_min(float a, float b, float* v, float* wa) {
  *v = a < b ? a : b;
  *wa = a < b ? 1.f : (a == b ? 0.5f : 0.f);
}

// p, t: [cx,cy,cz,w,l,h]; returns the IoU, writes d IoU / d p into dp[6] when dp != NULL (the gradient torch autograd
// gives through max / min / clamp of iou3d_calculator.py:201-330, ties split evenly)
__device__ static inline float aiou3d_eval(const float* __restrict__ p, const float* __restrict__ t, float eps, float* dp) {
  float wh[3], dwh_dc[3], dwh_ds[3], sp[3];
  float a1 = 1.f, a2 = 1.f, ov = 1.f;
  for (int a = 0; a < 3; ++a) {
    float p1 = p[a] - p[3 + a] / 2, p2 = p[a] + p[3 + a] / 2;
    float t1 = t[a] - t[3 + a] / 2, t2 = t[a] + t[3 + a] / 2;
    float lt, rb, wl, wr;
    tie_max(p1, t1, &lt, &wl);       // d lt / d p1
    tie_min(p2, t2, &rb, &wr);       // d rb / d p2
    float d = rb - lt;
    float pass = d >= 0.f ? 1.f : 0.f;
    wh[a] = d > 0.f ? d : 0.f;
    // wh = clamp(rb - lt): d/dc = wr - wl ; d/ds = wr/2 + wl/2
    dwh_dc[a] = pass * (wr - wl);
    dwh_ds[a] = pass * 0.5f * (wr + wl);
    sp[a] = p2 - p1;
    a1 *= sp[a];
    a2 *= (t2 - t1);
    ov *= wh[a];
  }
  float un = a1 + a2 - ov;
  float upass = un > eps ? 1.f : 0.f;     // torch.max(union, eps)
  float U = un > eps ? un : eps;
  if (dp) {
    for (int a = 0; a < 3; ++a) {
      int b = (a + 1) % 3, c = (a + 2) % 3;
      float dov_dwh = wh[b] * wh[c];
      float dov_dc = dov_dwh * dwh_dc[a];
      float dov_ds = dov_dwh * dwh_ds[a];
      float da1_ds = sp[b] * sp[c];          // area1 = prod (p2-p1): d/ds_a = prod of the others
      float dU_dc = upass * (-dov_dc);
      float dU_ds = upass * (da1_ds - dov_ds);
      dp[a] = (dov_dc * U - ov * dU_dc) / (U * U);
      dp[3 + a] = (dov_ds * U - ov * dU_ds) / (U * U);
    }
  }
  return ov / U;
}

// pred (n,6) [cx,cy,cz,w,l,h]; target rows of `tstride` floats whose first 6 are [cx,cy,cz,w,l,h]
__global__ void k_aiou3d(const float* __restrict__ pred, const float* __restrict__ target, int tstride, int64_t n,
                         float eps, float* __restrict__ iou, float* __restrict__ dpred) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float dp[6];
  iou[i] = aiou3d_eval(pred + i * 6, target + i * tstride, eps, dpred ? dp : nullptr);
  if (dpred)
    for (int a = 0; a < 6; ++a) dpred[i * 6 + a] = dp[a];
}

// ---------------------------------------------------------------------------------------------------
// The three losses of Fcaf3DNeckWithHead._loss_single (fcaf3d_neck_with_head.py:160-203) for ALL locations of the batch in
// one pass (r3; yaw-less heads: ScanNet / S3DIS).  Row r of scene s carries the weights w_pos = inv_pos[s] =
// 1 / (B max(n_pos_s, 1)) and w_den = inv_den[s] = 1 / (B max(sum of centerness targets_s, 1e-6)), so the sums below are
// the means over scenes of the per-scene losses:
//   loss_cls        = sum_r w_pos sum_c focal(cls[r,c], labels[r] == c)                          (:180, mmcv sigmoid_focal_loss)
//   loss_centerness = sum_{r positive} w_pos BCEWithLogits(centerness[r], ct[r])                   (:191-193)
//   loss_bbox       = sum_{r: ct w_den > 0} ct[r] w_den (1 - IoU(decode(points[r], bbox_pred[r]), bt[r]))   (:194-199, :281-300,
//                      iou3d_loss.py:21-35)       decode: centre = p + (d1-d0, d3-d2, d5-d4)/2, size = (d0+d1, d2+d3, d4+d5)
// Forward: per-block partial sums [nb][4] in a fixed order -> k_fcaf3d_loss_final (deterministic).  Backward: the same
// per-row arithmetic with the incoming scalar gradients; rows without weight write exact zeros.
__device__ static inline void decode6(const float* __restrict__ pt, const float* __restrict__ d, float* box) {
  box[0] = pt[0] + (d[1] - d[0]) / 2;
  box[1] = pt[1] + (d[3] - d[2]) / 2;
  box[2] = pt[2] + (d[5] - d[4]) / 2;
  box[3] = d[0] + d[1];
  box[4] = d[2] + d[3];
  box[5] = d[4] + d[5];
}

__device__ static inline float bce_logits(float x, float t) {          // torch BCEWithLogits: max(x,0) - x t + log1p(exp(-|x|))
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void k_fcaf3d_loss_fwd(const float* __restrict__ points, const float* __restrict__ bbox_pred,
                                                         const float* __restrict__ centerness, const float* __restrict__ cls,
                                                         const float* __restrict__ ct, const float* __restrict__ bt,
                                                         const long long* __restrict__ labels, const int* __restrict__ scene,
                                                         const float* __restrict__ inv_pos, const float* __restrict__ inv_den,
                                                         int64_t N, int C, float gamma, float alpha, float* __restrict__ part) {
  __shared__ float red[4][3];
  constexpr int CMAX = 32;
  __shared__ float sc[256 * CMAX];          // the block's 256 x C class scores, loaded as they lie (r6: a row per thread walked 72-byte strides)
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool staged = C <= CMAX;
  if (staged) {
    const int64_t base = (int64_t)blockIdx.x * 256 * C, lim = N * C;
    for (int i = threadIdx.x; i < 256 * C; i += 256) sc[i] = base + i < lim ? cls[base + i] : 0.f;
    __syncthreads();
  }
  float v[3] = {0.f, 0.f, 0.f};
  if (r < N) {
    const int s = scene[r];
    const float wp = inv_pos[s], wd = inv_den[s];
    const long long y = labels[r];
    float acc = 0.f;
    if (staged) for (int c = 0; c < C; ++c) acc += focal_elem(sc[threadIdx.x * C + c], y == c, gamma, alpha);
    else for (int c = 0; c < C; ++c) acc += focal_elem(cls[r * C + c], y == c, gamma, alpha);
    v[0] = acc * wp;
    if (y >= 0) v[1] = bce_logits(centerness[r], ct[r]) * wp;
    const float w = ct[r] * wd;
    if (w > 0.f) {
      float box[6];
      decode6(points + r * 3, bbox_pred + r * 6, box);
      v[2] = (1.f - aiou3d_eval(box, bt + r * 7, 1e-6f, nullptr)) * w;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
    for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_xor(v[j], off, 64);
  if ((threadIdx.x & 63) == 0)
    for (int j = 0; j < 3; ++j) red[threadIdx.x >> 6][j] = v[j];
  __syncthreads();
  if (threadIdx.x < 3) part[(int64_t)blockIdx.x * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[j] = lw[j] * sum_b part[b][j]  (one block, fixed order; double accumulation)
__global__ __launch_bounds__(256) void k_fcaf3d_loss_final(const float* __restrict__ part, int64_t nb, float lw0, float lw1,
                                                           float lw2, float* __restrict__ o0, float* __restrict__ o1,
                                                           float* __restrict__ o2) {
  __shared__ double red[4][3];
  double v[3] = {0.0, 0.0, 0.0};
  for (int64_t b = threadIdx.x; b < nb; b += 256)
    for (int j = 0; j < 3; ++j) v[j] += (double)fc_ld(&part[b * 4 + j]);
  for (int j = 0; j < 3; ++j)
    for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_xor(v[j], off, 64);
  if ((threadIdx.x & 63) == 0)
    for (int j = 0; j < 3; ++j) red[threadIdx.x >> 6][j] = v[j];
  __syncthreads();
  if (threadIdx.x == 0) {
    *o0 = lw0 * (float)((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
    *o1 = lw1 * (float)((red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
    *o2 = lw2 * (float)((red[0][2] + red[1][2]) + (red[2][2] + red[3][2]));
  }
}

// gradients of (lw0 g0 loss_cls + lw1 g1 loss_centerness + lw2 g2 loss_bbox) w.r.t. cls (N,C), centerness (N), bbox_pred (N,6);
// g0..g2: device scalars (NULL = that loss is not part of the objective)
__global__ __launch_bounds__(256) void k_fcaf3d_loss_bwd(const float* __restrict__ points, const float* __restrict__ bbox_pred,
                                                         const float* __restrict__ centerness, const float* __restrict__ cls,
                                                         const float* __restrict__ ct, const float* __restrict__ bt,
                                                         const long long* __restrict__ labels, const int* __restrict__ scene,
                                                         const float* __restrict__ inv_pos, const float* __restrict__ inv_den,
                                                         int64_t N, int C, float gamma, float alpha, float lw0, float lw1,
                                                         float lw2, const float* __restrict__ g0, const float* __restrict__ g1,
                                                         const float* __restrict__ g2, float* __restrict__ gcls,
                                                         float* __restrict__ gcent, float* __restrict__ gbbox) {
  // the class-score gradient one ELEMENT per thread (coalesced over the (N, C) array; r6: one row per thread walked 72-byte strides,
  // 144 us for 590k locations), then centerness and box one ROW per thread; the grid covers N C elements >= N rows
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < N * C) {
    const int64_t re = e / C;
    const int c = (int)(e - re * C);
    const long long ye = labels[re];
    const float s0 = g0 ? g0[0] * lw0 * inv_pos[scene[re]] : 0.f;
    const float x = cls[e];
    const float p = 1.f / (1.f + expf(-x));
    float g;
    if (ye == c) g = -alpha * pow_gamma(1.f - p, gamma) * (1.f - p - gamma * p * logf(fmaxf(p, FLT_MIN)));
    else g = -(1.f - alpha) * pow_gamma(p, gamma) * (gamma * (1.f - p) * logf(fmaxf(1.f - p, FLT_MIN)) - p);
    gcls[e] = g * s0;
  }
  const int64_t r = e;
  if (r >= N) return;
  const int s = scene[r];
  const float wp = inv_pos[s], wd = inv_den[s];
  const long long y = labels[r];
  float gc = 0.f;
  if (y >= 0 && g1) {
    const float x = centerness[r];
    gc = (1.f / (1.f + expf(-x)) - ct[r]) * (g1[0] * lw1 * wp);
  }
  gcent[r] = gc;
  float gb[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float w = ct[r] * wd;
  if (w > 0.f && g2) {
    float box[6], dp[6];
    const float* d = bbox_pred + r * 6;
    decode6(points + r * 3, d, box);
    aiou3d_eval(box, bt + r * 7, 1e-6f, dp);
    const float sc = -(g2[0] * lw2 * w);                    // loss = w (1 - IoU)
    for (int a = 0; a < 3; ++a) {                           // centre_a = p + (d[2a+1] - d[2a]) / 2, size_a = d[2a] + d[2a+1]
      gb[2 * a] = sc * (dp[3 + a] - dp[a] / 2);
      gb[2 * a + 1] = sc * (dp[3 + a] + dp[a] / 2);
    }
  }
  for (int j = 0; j < 6; ++j) gbbox[r * 6 + j] = gb[j];
}

extern "C" {

int fc_focal_loss_fwd(const float* logits, const long long* labels, const float* row_weight, int64_t n, int C, float gamma,
                      float alpha, float* loss_rows, hipStream_t stream) {
  if (n < 0 || C < 1) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_focal_fwd<<<(unsigned)fc_cdiv(n, 256), 256, 0, stream>>>(logits, labels, row_weight, n, C, gamma, alpha, loss_rows);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_focal_loss_bwd(const float* logits, const long long* labels, const float* row_weight, int64_t n, int C, float gamma,
                      float alpha, const float* gscale_dev, float* glogits, hipStream_t stream) {
  if (n < 0 || C < 1) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_focal_bwd<<<(unsigned)fc_cdiv(n * C, 256), 256, 0, stream>>>(logits, labels, row_weight, n, C, gamma, alpha, gscale_dev,
                                                                glogits);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_aiou3d_fwd_bwd(const float* pred, const float* target, int target_stride, int64_t n, float eps, float* iou,
                      float* dpred, hipStream_t stream) {
  if (n < 0 || target_stride < 6) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_aiou3d<<<(unsigned)fc_cdiv(n, 128), 128, 0, stream>>>(pred, target, target_stride, n, eps, iou, dpred);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int64_t fc_fcaf3d_loss_ws_bytes(int64_t n) { return fc_cdiv(n > 0 ? n : 1, 256) * 4 * (int64_t)sizeof(float); }

int fc_fcaf3d_loss_fwd(const float* points, const float* bbox_pred, const float* centerness, const float* cls_score,
                       const float* centerness_t, const float* bbox_t, const long long* labels, const int* scene,
                       const float* inv_pos, const float* inv_den, int64_t n, int n_classes, float gamma, float alpha,
                       float lw_cls, float lw_centerness, float lw_bbox, float* loss_cls, float* loss_centerness,
                       float* loss_bbox, void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (n < 0 || n_classes < 1 || !loss_cls || !loss_centerness || !loss_bbox) return FC_EINVAL;
  if (ws_bytes < fc_fcaf3d_loss_ws_bytes(n)) return FC_EWS;
  const int64_t nb = n > 0 ? fc_cdiv(n, 256) : 0;
  if (nb) {
    k_fcaf3d_loss_fwd<<<(unsigned)nb, 256, 0, stream>>>(points, bbox_pred, centerness, cls_score, centerness_t, bbox_t, labels,
                                                       scene, inv_pos, inv_den, n, n_classes, gamma, alpha, (float*)ws);
    FC_CHECK_LAUNCH();
  }
  k_fcaf3d_loss_final<<<1, 256, 0, stream>>>((const float*)ws, nb, lw_cls, lw_centerness, lw_bbox, loss_cls, loss_centerness,
                                             loss_bbox);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_fcaf3d_loss_bwd(const float* points, const float* bbox_pred, const float* centerness, const float* cls_score,
                       const float* centerness_t, const float* bbox_t, const long long* labels, const int* scene,
                       const float* inv_pos, const float* inv_den, int64_t n, int n_classes, float gamma, float alpha,
                       float lw_cls, float lw_centerness, float lw_bbox, const float* g_cls, const float* g_centerness,
                       const float* g_bbox, float* grad_cls_score, float* grad_centerness, float* grad_bbox_pred,
                       hipStream_t stream) {
  if (n < 0 || n_classes < 1) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_fcaf3d_loss_bwd<<<(unsigned)fc_cdiv(n * n_classes, 256), 256, 0, stream>>>(points, bbox_pred, centerness, cls_score, centerness_t, bbox_t,
                                                                  labels, scene, inv_pos, inv_den, n, n_classes, gamma, alpha,
                                                                  lw_cls, lw_centerness, lw_bbox, g_cls, g_centerness, g_bbox,
                                                                  grad_cls_score, grad_centerness, grad_bbox_pred);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// rotated 3D IoU with forward-mode derivatives w.r.t. the 7 parameters of `pred`
struct D7 {
  float v;
  float d[7];
};
__device__ static inline D7 dconst(float c) { D7 r; r.v = c; for (int i = 0; i < 7; ++i) r.d[i] = 0.f; return r; }
__device__ static inline D7 dvar(float c, int i) { D7 r = dconst(c); r.d[i] = 1.f; return r; }
__device__ static inline D7 operator+(const D7& a, const D7& b) { D7 r; r.v = a.v + b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ static inline D7 operator-(const D7& a, const D7& b) { D7 r; r.v = a.v - b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ static inline D7 operator*(const D7& a, const D7& b) { D7 r; r.v = a.v * b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ static inline D7 operator/(const D7& a, const D7& b) {
  D7 r; r.v = a.v / b.v; float inv = 1.f / b.v;
  for (int i = 0; i < 7; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ static inline D7 dscale(const D7& a, float s) { D7 r; r.v = a.v * s; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ static inline D7 dmin_tie(const D7& a, const D7& b) {     // torch.min(a,b): ties split the gradient
  if (a.v < b.v) return a;
  if (b.v < a.v) return b;
  return dscale(a + b, 0.5f);
}
__device__ static inline D7 dmax_tie(const D7& a, const D7& b) {
  if (a.v > b.v) return a;
  if (b.v > a.v) return b;
  return dscale(a + b, 0.5f);
}

struct V2 { D7 x, y; };

__device__ static void corners_of(const D7& cx, const D7& cy, const D7& w, const D7& h, const D7& alpha, V2* out) {
  const float sx[4] = {0.5f, -0.5f, -0.5f, 0.5f}, sy[4] = {0.5f, 0.5f, -0.5f, -0.5f};
  D7 cs, sn;
  cs.v = cosf(alpha.v); sn.v = sinf(alpha.v);
  for (int i = 0; i < 7; ++i) { cs.d[i] = -sn.v * alpha.d[i]; sn.d[i] = cs.v * alpha.d[i]; }
  for (int k = 0; k < 4; ++k) {
    D7 x4 = dscale(w, sx[k]), y4 = dscale(h, sy[k]);
    out[k].x = x4 * cs - y4 * sn + cx;
    out[k].y = x4 * sn + y4 * cs + cy;
  }
}

// value-only: corner m lies inside the rectangle with corners q[0..3] (box_intersection_2d.py:57-82)
__device__ static inline bool corner_in_rect(const V2& m, const V2* q) {
  float abx = q[1].x.v - q[0].x.v, aby = q[1].y.v - q[0].y.v;
  float adx = q[3].x.v - q[0].x.v, ady = q[3].y.v - q[0].y.v;
  float amx = m.x.v - q[0].x.v, amy = m.y.v - q[0].y.v;
  float pab = (abx * amx + aby * amy) / (abx * abx + aby * aby);
  float pad = (adx * amx + ady * amy) / (adx * adx + ady * ady);
  return pab > -1e-6f && pab < 1.f + 1e-6f && pad > -1e-6f && pad < 1.f + 1e-6f;
}

__global__ void k_riou3d(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ weight,
                         int64_t n, float* __restrict__ iou_out, float* __restrict__ dpred) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (weight && !(weight[i] > 0.f)) {
    iou_out[i] = 0.f;
    for (int e = 0; e < 7; ++e) dpred[i * 7 + e] = 0.f;
    return;
  }
  const float* p = pred + i * 7;
  const float* t = target + i * 7;
  D7 P[7];
  for (int e = 0; e < 7; ++e) P[e] = dvar(p[e], e);
  V2 vert[24];
  bool valid[24];
  corners_of(P[0], P[1], P[3], P[4], P[6], vert);                                     // 0..3  box1
  corners_of(dconst(t[0]), dconst(t[1]), dconst(t[3]), dconst(t[4]), dconst(t[6]), vert + 4);   // 4..7 box2
  // 16 edge-edge intersections (box_intersection_2d.py:13-54)
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) {
      const V2& p1 = vert[a]; const V2& p2 = vert[(a + 1) & 3];
      const V2& p3 = vert[4 + b]; const V2& p4 = vert[4 + ((b + 1) & 3)];
      D7 num = (p1.x - p2.x) * (p3.y - p4.y) - (p1.y - p2.y) * (p3.x - p4.x);
      D7 den_t = (p1.x - p3.x) * (p3.y - p4.y) - (p1.y - p3.y) * (p3.x - p4.x);
      float den_u = (p1.x.v - p2.x.v) * (p1.y.v - p3.y.v) - (p1.y.v - p2.y.v) * (p1.x.v - p3.x.v);
      float tt = num.v == 0.f ? -1.f : den_t.v / num.v;
      float uu = num.v == 0.f ? -1.f : -den_u / num.v;
      bool m = tt > 0.f && tt < 1.f && uu > 0.f && uu < 1.f;
      int k = 8 + a * 4 + b;
      valid[k] = m;
      if (m) {
        D7 ts = den_t / (num + dconst(1e-8f));
        vert[k].x = p1.x + ts * (p2.x - p1.x);
        vert[k].y = p1.y + ts * (p2.y - p1.y);
      } else {
        vert[k].x = dconst(0.f);
        vert[k].y = dconst(0.f);
      }
    }
  for (int k = 0; k < 4; ++k) {
    valid[k] = corner_in_rect(vert[k], vert + 4);
    valid[4 + k] = corner_in_rect(vert[4 + k], vert);
  }
  // order the valid vertices by polar angle about their mean (sort_v)
  int nv = 0;
  float mx = 0.f, my = 0.f;
  for (int k = 0; k < 24; ++k)
    if (valid[k]) { mx += vert[k].x.v; my += vert[k].y.v; ++nv; }
  int order[24];
  int cnt = 0;
  if (nv >= 3) {
    mx /= nv; my /= nv;
    float ang[24];
    for (int k = 0; k < 24; ++k) {
      if (!valid[k]) continue;
      float g = sort_key(vert[k].y.v - my, vert[k].x.v - mx);
      int m = cnt - 1;
      while (m >= 0 && ang[m] > g) { ang[m + 1] = ang[m]; order[m + 1] = order[m]; --m; }   // stable insertion
      ang[m + 1] = g; order[m + 1] = k;
      ++cnt;
    }
    // drop coincident neighbours (identical boxes list every corner twice)
    int kept = 0;
    for (int k = 0; k < cnt; ++k) {
      if (kept > 0) {
        int q = order[kept - 1];
        if (fmaxf(fabsf(vert[order[k]].x.v - vert[q].x.v), fabsf(vert[order[k]].y.v - vert[q].y.v)) <= 1e-6f) continue;
      }
      order[kept++] = order[k];
    }
    if (kept > 1) {
      int a0 = order[0], q = order[kept - 1];
      if (fmaxf(fabsf(vert[a0].x.v - vert[q].x.v), fabsf(vert[a0].y.v - vert[q].y.v)) <= 1e-6f) --kept;
    }
    cnt = kept > 8 ? 8 : kept;
  }
  D7 total = dconst(0.f);
  if (cnt >= 3) {
    for (int k = 0; k < cnt; ++k) {
      const V2& a = vert[order[k]];
      const V2& b = vert[order[(k + 1) % cnt]];
      total = total + (a.x * b.y - a.y * b.x);
    }
  }
  D7 inter = dscale(total, total.v > 0.f ? 0.5f : (total.v < 0.f ? -0.5f : 0.f));
  D7 area1 = P[3] * P[4];
  D7 u2d = area1 + dconst(t[3] * t[4]) - inter;
  D7 iou2d = inter / u2d;
  D7 zmax1 = P[2] + dscale(P[5], 0.5f), zmin1 = P[2] - dscale(P[5], 0.5f);
  D7 zmax2 = dconst(t[2] + t[5] * 0.5f), zmin2 = dconst(t[2] - t[5] * 0.5f);
  D7 zo = dmin_tie(zmax1, zmax2) - dmax_tie(zmin1, zmin2);
  if (!(zo.v >= 0.f)) zo = dconst(0.f);                       // clamp_min(0): gradient passes at x >= 0
  else if (zo.v == 0.f) zo.v = 0.f;
  D7 inter3d = iou2d * u2d * zo;
  D7 v1 = P[3] * P[4] * P[5];
  D7 u3d = v1 + dconst(t[3] * t[4] * t[5]) - inter3d;
  D7 r = inter3d / u3d;
  iou_out[i] = r.v;
  for (int e = 0; e < 7; ++e) dpred[i * 7 + e] = r.d[e];
}

// ----------------------------------------------------------------------------------------------
// Standalone `sort_v` (Rotated_IoU cuda_op, bound by the reference at box_intersection_2d.py:147; SURVEY.md
// Appendix D): for every box pair, the indices of the valid intersection-polygon vertices in angular order about the
// origin (the caller has centred them on their mean, box_intersection_2d.py:144-146), coincident neighbours
// dropped, polygon closed, padded with a masked intersection slot.  One thread per pair.
__global__ void k_sort_v(const float* __restrict__ vertices, const unsigned char* __restrict__ mask,
                         const int* __restrict__ num_valid, int64_t n, int* __restrict__ idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* v = vertices + i * 48;
  const unsigned char* m = mask + i * 24;
  int pad = 8;
  for (int k = 8; k < 24; ++k)
    if (!m[k]) { pad = k; break; }
  int* out = idx + i * 9;
  int order[24];
  float ang[24];
  int cnt = 0;
  if (num_valid[i] >= 3) {
    for (int k = 0; k < 24; ++k) {
      if (!m[k]) continue;
      float g = sort_key(v[2 * k + 1], v[2 * k]);
      int q = cnt - 1;
      while (q >= 0 && ang[q] > g) { ang[q + 1] = ang[q]; order[q + 1] = order[q]; --q; }     // stable insertion
      ang[q + 1] = g; order[q + 1] = k;
      ++cnt;
    }
    int kept = 0;
    for (int k = 0; k < cnt; ++k) {
      if (kept > 0) {
        int q = order[kept - 1], c = order[k];
        if (fmaxf(fabsf(v[2 * c] - v[2 * q]), fabsf(v[2 * c + 1] - v[2 * q + 1])) <= 1e-6f) continue;
      }
      order[kept++] = order[k];
    }
    if (kept > 1) {
      int a0 = order[0], q = order[kept - 1];
      if (fmaxf(fabsf(v[2 * a0] - v[2 * q]), fabsf(v[2 * a0 + 1] - v[2 * q + 1])) <= 1e-6f) --kept;
    }
    cnt = kept > 8 ? 8 : kept;
  }
  if (cnt < 3) {
    for (int k = 0; k < 9; ++k) out[k] = pad;
    return;
  }
  for (int k = 0; k < cnt; ++k) out[k] = order[k];
  out[cnt] = order[0];
  for (int k = cnt + 1; k < 9; ++k) out[k] = pad;
}

extern "C" {

int fc_sort_v(const float* vertices, const unsigned char* mask, const int* num_valid, int64_t n_pairs, int* idx,
              hipStream_t stream) {
  if (n_pairs < 0) return FC_EINVAL;
  if (n_pairs == 0) return FC_OK;
  k_sort_v<<<(unsigned)fc_cdiv(n_pairs, 64), 64, 0, stream>>>(vertices, mask, num_valid, n_pairs, idx);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_riou3d_fwd_bwd(const float* pred, const float* target, const float* weight, int64_t n, float* iou, float* dpred,
                      hipStream_t stream) {
  if (n < 0) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_riou3d<<<(unsigned)fc_cdiv(n, 64), 64, 0, stream>>>(pred, target, weight, n, iou, dpred);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"
