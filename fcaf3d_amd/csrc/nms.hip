// Per-class BEV NMS on gfx950 — all classes of a scene in two launches, nothing copied to the host.
//
// Replaces mmdet3d/ops/pcdet_nms: nms_gpu / nms_normal_gpu (src/iou3d_nms.cpp:90-186) and their
// kernels nms_kernel / nms_normal_kernel (src/iou3d_nms_kernel.cu:267-372) with the device
// functions they call (:35-234, :314-325), as driven per class by
// Fcaf3DNeckWithHead._nms (fcaf3d_neck_with_head.py:332-353).
//
// Kernel 1: one wavefront per (row block, column block, class): thread = row box, 64 column boxes in
//           LDS, 64-bit suppression word per (row, column block) — wave64 makes the word one lane each.
// Kernel 2: one wavefront per class runs the greedy scan on the device (the reference copies the whole
//           mask to the host and scans there, iou3d_nms.cpp:111-132): lane j owns removed-word j.
#include "fc_common.h"
// exact products as in the reference's torch / host code (e.g. num == 0 for parallel edges): no FMA contraction
#pragma clang fp contract(off)

#define NMS_EPS 1e-8f

struct P2 { float x, y; };

__device__ static inline float cross3(P2 p1, P2 p2, P2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

// proper crossing of segments p0p1 and q0q1 (touching / collinear do not count)
__device__ static inline bool seg_cross(P2 p1, P2 p0, P2 q1, P2 q0, P2* out) {
  bool overlap = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                 fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!overlap) return false;
  float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > NMS_EPS) {
    out->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    out->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    out->x = (b0 * c1 - b1 * c0) / D;
    out->y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ static inline bool in_box_bev(const float* box, P2 p) {
  const float MARGIN = 1e-2f;
  float c = cosf(-box[6]), s = sinf(-box[6]);
  float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
  float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
  return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

__device__ static inline void box_corners(const float* box, P2* c /*[5]*/) {
  float hx = box[3] / 2, hy = box[4] / 2;
  float cs = cosf(box[6]), sn = sinf(box[6]);
  const float sx[4] = {-1.f, 1.f, 1.f, -1.f}, sy[4] = {-1.f, -1.f, 1.f, 1.f};
  for (int k = 0; k < 4; ++k) {
    // axis-aligned corner, then rotation about the centre
    float px = box[0] + sx[k] * hx, py = box[1] + sy[k] * hy;
    c[k].x = (px - box[0]) * cs + (py - box[1]) * (-sn) + box[0];
    c[k].y = (px - box[0]) * sn + (py - box[1]) * cs + box[1];
  }
  c[4] = c[0];
}

// area of the intersection polygon of two rotated BEV rectangles (x,y,_,dx,dy,_,heading)
__device__ static float bev_overlap_rotated(const float* a, const float* b) {
  P2 ca[5], cb[5], pts[24];
  box_corners(a, ca);
  box_corners(b, cb);
  int cnt = 0;
  P2 ctr = {0.f, 0.f};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 x;
      if (seg_cross(ca[i + 1], ca[i], cb[j + 1], cb[j], &x)) {
        pts[cnt++] = x;
        ctr.x += x.x; ctr.y += x.y;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box_bev(a, cb[k])) { pts[cnt++] = cb[k]; ctr.x += cb[k].x; ctr.y += cb[k].y; }
    if (in_box_bev(b, ca[k])) { pts[cnt++] = ca[k]; ctr.x += ca[k].x; ctr.y += ca[k].y; }
  }
  if (cnt < 3) return 0.f;
  ctr.x /= cnt; ctr.y /= cnt;
  float ang[24];
  for (int k = 0; k < cnt; ++k) ang[k] = atan2f(pts[k].y - ctr.y, pts[k].x - ctr.x);
  for (int k = 1; k < cnt; ++k) {            // stable insertion sort by polar angle
    P2 p = pts[k]; float g = ang[k];
    int m = k - 1;
    while (m >= 0 && ang[m] > g) { pts[m + 1] = pts[m]; ang[m + 1] = ang[m]; --m; }
    pts[m + 1] = p; ang[m + 1] = g;
  }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    float ux = pts[k].x - pts[0].x, uy = pts[k].y - pts[0].y;
    float vx = pts[k + 1].x - pts[0].x, vy = pts[k + 1].y - pts[0].y;
    area += ux * vy - uy * vx;
  }
  return fabsf(area) / 2.f;
}

__device__ static inline float iou_bev_rotated(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4];
  float ov = bev_overlap_rotated(a, b);
  return ov / fmaxf(sa + sb - ov, NMS_EPS);
}

__device__ static inline float iou_bev_aligned(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  float inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, NMS_EPS);
}

// grid (block pairs of the UPPER triangle, 1, classes), 64 threads.  The greedy scan only reads the words of column blocks at or
// after a row's own block (a box can only suppress boxes that come later), so the lower triangle is never computed (r4: half
// the IoU work and half the workgroups of the r1-r3 (col blocks, row blocks) grid; the reference computes the full matrix,
// iou3d_nms_kernel.cu:267-313, and its host scan likewise only acts on later boxes).  Pair t = cb (cb + 1) / 2 + rb, rb <= cb.
__global__ __launch_bounds__(64) void k_nms_mask(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                 int stride, float thresh, int rotated,
                                                 unsigned long long* __restrict__ mask) {
  const int seg = blockIdx.z;
  const int n = counts[seg];
  const int t = blockIdx.x;
  int cb = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
  while ((cb + 1) * (cb + 2) / 2 <= t) ++cb;           // float rounding: settle on cb (cb + 1) / 2 <= t < (cb + 1)(cb + 2) / 2
  while (cb * (cb + 1) / 2 > t) --cb;
  const int rblk = t - cb * (cb + 1) / 2;
  const int row0 = rblk * 64, col0 = cb * 64;
  if (row0 >= n || col0 >= n) return;
  const int words = (stride + 63) / 64;
  const float* sb = boxes + (int64_t)seg * stride * 7;
  __shared__ float cbox[64 * 7];
  int col_size = min(n - col0, 64);
  if ((int)threadIdx.x < col_size)
    for (int e = 0; e < 7; ++e) cbox[threadIdx.x * 7 + e] = sb[(int64_t)(col0 + threadIdx.x) * 7 + e];
  __syncthreads();
  int row = row0 + threadIdx.x;
  if (row >= n) return;
  float rb[7];
  for (int e = 0; e < 7; ++e) rb[e] = sb[(int64_t)row * 7 + e];
  unsigned long long bits = 0;
  int start = (row0 == col0) ? (int)threadIdx.x + 1 : 0;
  for (int j = start; j < col_size; ++j) {
    float v = rotated ? iou_bev_rotated(rb, cbox + j * 7) : iou_bev_aligned(rb, cbox + j * 7);
    if (v > thresh) bits |= 1ull << j;
  }
  mask[((int64_t)seg * stride + row) * words + cb] = bits;
}

// grid (classes), 64 threads: lane j holds removed-words j, j+64, ...
__global__ __launch_bounds__(64) void k_nms_greedy(const unsigned long long* __restrict__ mask, const int* __restrict__ counts,
                                                   int stride, int* __restrict__ keep, int* __restrict__ keep_count) {
  const int seg = blockIdx.x;
  const int n = counts[seg];
  const int words = (stride + 63) / 64;
  const int lane = threadIdx.x;
  constexpr int MAXW = 16;                    // up to 64*16 words = 65536 boxes per class
  unsigned long long remv[MAXW];
#pragma unroll
  for (int w = 0; w < MAXW; ++w) remv[w] = 0ull;
  const unsigned long long* sm = mask + (int64_t)seg * stride * words;
  int* sk = keep + (int64_t)seg * stride;
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    int nblock = i >> 6, inblock = i & 63;
    // the word that holds box i lives in lane nblock%64, slot nblock/64
    unsigned long long word = 0ull;
#pragma unroll
    for (int w = 0; w < MAXW; ++w)
      if (w == (nblock >> 6)) word = remv[w];
    word = __shfl(word, nblock & 63, 64);
    if (word & (1ull << inblock)) continue;     // wave-uniform
    if (lane == 0) sk[kept] = i;
    ++kept;
    const unsigned long long* row = sm + (int64_t)i * words;
    const int nwords_used = (n + 63) >> 6;
#pragma unroll
    for (int w = 0; w < MAXW; ++w) {
      int j = w * 64 + lane;
      if (j >= nblock && j < nwords_used) remv[w] |= row[j];
    }
  }
  if (lane == 0) keep_count[seg] = kept;
}

extern "C" {

int64_t fc_nms_bev_ws_bytes(int nseg, int stride) {
  return (int64_t)nseg * stride * ((stride + 63) / 64) * (int64_t)sizeof(unsigned long long);
}

// boxes (nseg, stride, 7) — segment s holds counts[s] boxes sorted by descending score.
// keep (nseg, stride): ascending positions of the surviving boxes; keep_count (nseg).
int fc_nms_bev(const float* boxes, const int* counts_dev, int nseg, int stride, float thresh, int rotated,
               unsigned long long* mask_ws, int64_t ws_bytes, int* keep, int* keep_count, hipStream_t stream) {
  if (nseg < 1 || stride < 1 || stride > 65536 || nseg > 65535) return FC_EINVAL;
  if (ws_bytes < fc_nms_bev_ws_bytes(nseg, stride)) return FC_EWS;
  int nb = (stride + 63) / 64;
  dim3 grid(nb * (nb + 1) / 2, 1, nseg);
  k_nms_mask<<<grid, 64, 0, stream>>>(boxes, counts_dev, stride, thresh, rotated, mask_ws);
  FC_CHECK_LAUNCH();
  k_nms_greedy<<<nseg, 64, 0, stream>>>(mask_ws, counts_dev, stride, keep, keep_count);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// pairwise BEV IoU matrix (N,M) — pcdet `boxes_iou_bev_gpu` (iou3d_nms_kernel.cu:251-265); used by the parity tests
__global__ void k_iou_bev_matrix(const float* __restrict__ a, int n, const float* __restrict__ b, int m, int rotated,
                                 float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n * m) return;
  int i = (int)(t / m), j = (int)(t % m);
  out[t] = rotated ? iou_bev_rotated(a + i * 7, b + j * 7) : iou_bev_aligned(a + i * 7, b + j * 7);
}

int fc_boxes_iou_bev(const float* boxes_a, int n, const float* boxes_b, int m, int rotated, float* out,
                     hipStream_t stream) {
  if (n < 0 || m < 0) return FC_EINVAL;
  if (n == 0 || m == 0) return FC_OK;
  k_iou_bev_matrix<<<(unsigned)fc_cdiv((int64_t)n * m, 128), 128, 0, stream>>>(boxes_a, n, boxes_b, m, rotated, out);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"
