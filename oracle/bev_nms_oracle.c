/* ORACLE — test infrastructure only (built by oracle/Makefile into libfcaf3d_oracle.so, called via
 * ctypes from tests/ and bench.py's cpu_baseline leg; never linked into the product).
 *
 * Plain-C restatement of the BEV IoU the reference's NMS is built on:
 *   rotated   iou_bev    mmdet3d/ops/pcdet_nms/src/iou3d_nms_kernel.cu:35-234 (== src/iou3d_cpu.cpp:41-229)
 *   aligned   iou_normal mmdet3d/ops/pcdet_nms/src/iou3d_nms_kernel.cu:314-325
 * Pinned against tests/golden/bev_iou.npz, produced by the reference's own iou3d_cpu.cpp compiled
 * here (oracle/_ref). */
#include <math.h>

#define EPSF 1e-8f
typedef struct { float x, y; } pt;

static float crs(pt p1, pt p2, pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
static float mn(float a, float b) { return a > b ? b : a; }
static float mx(float a, float b) { return a > b ? a : b; }

static int seg_x(pt p1, pt p0, pt q1, pt q0, pt* ans) {
  if (!(mn(p0.x, p1.x) <= mx(q0.x, q1.x) && mn(q0.x, q1.x) <= mx(p0.x, p1.x) &&
        mn(p0.y, p1.y) <= mx(q0.y, q1.y) && mn(q0.y, q1.y) <= mx(p0.y, p1.y))) return 0;
  float s1 = crs(q0, p1, p0), s2 = crs(p1, q1, p0), s3 = crs(p0, q1, q0), s4 = crs(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = crs(q1, p1, p0);
  if (fabs(s5 - s1) > EPSF) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static int in_box(const float* b, pt p) {
  float c = cos(-b[6]), s = sin(-b[6]);
  float rx = (p.x - b[0]) * c + (p.y - b[1]) * (-s), ry = (p.x - b[0]) * s + (p.y - b[1]) * c;
  return fabs(rx) < b[3] / 2 + 1e-2 && fabs(ry) < b[4] / 2 + 1e-2;
}

static void corners(const float* b, pt* c) {
  float x1 = b[0] - b[3] / 2, x2 = b[0] + b[3] / 2, y1 = b[1] - b[4] / 2, y2 = b[1] + b[4] / 2;
  float cs = cos(b[6]), sn = sin(b[6]);
  float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
  for (int k = 0; k < 4; k++) {
    c[k].x = (px[k] - b[0]) * cs + (py[k] - b[1]) * (-sn) + b[0];
    c[k].y = (px[k] - b[0]) * sn + (py[k] - b[1]) * cs + b[1];
  }
  c[4] = c[0];
}

static float overlap(const float* a, const float* b) {
  pt ca[5], cb[5], p[24], ctr = {0, 0};
  int n = 0;
  corners(a, ca); corners(b, cb);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
    if (seg_x(ca[i + 1], ca[i], cb[j + 1], cb[j], &p[n])) { ctr.x += p[n].x; ctr.y += p[n].y; n++; }
  for (int k = 0; k < 4; k++) {
    if (in_box(a, cb[k])) { ctr.x += cb[k].x; ctr.y += cb[k].y; p[n++] = cb[k]; }
    if (in_box(b, ca[k])) { ctr.x += ca[k].x; ctr.y += ca[k].y; p[n++] = ca[k]; }
  }
  ctr.x /= n; ctr.y /= n;
  for (int j = 0; j < n - 1; j++) for (int i = 0; i < n - j - 1; i++)
    if (atan2(p[i].y - ctr.y, p[i].x - ctr.x) > atan2(p[i + 1].y - ctr.y, p[i + 1].x - ctr.x)) {
      pt t = p[i]; p[i] = p[i + 1]; p[i + 1] = t;
    }
  float area = 0;
  for (int k = 0; k < n - 1; k++) {
    pt u = {p[k].x - p[0].x, p[k].y - p[0].y}, v = {p[k + 1].x - p[0].x, p[k + 1].y - p[0].y};
    area += u.x * v.y - u.y * v.x;
  }
  return fabs(area) / 2.0;
}

float oracle_iou_bev(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4], so = overlap(a, b);
  return so / fmaxf(sa + sb - so, EPSF);
}

float oracle_iou_normal(const float* a, const float* b) {
  float l = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), r = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float t = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), d = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float w = fmaxf(r - l, 0.f), h = fmaxf(d - t, 0.f), s = w * h;
  return s / fmaxf(a[3] * a[4] + b[3] * b[4] - s, EPSF);
}

/* (n,m) IoU matrix */
void oracle_iou_matrix(const float* a, int n, const float* b, int m, int rotated, float* out) {
  for (int i = 0; i < n; i++) for (int j = 0; j < m; j++)
    out[i * m + j] = rotated ? oracle_iou_bev(a + 7 * i, b + 7 * j) : oracle_iou_normal(a + 7 * i, b + 7 * j);
}

/* greedy NMS over boxes already sorted by descending score (iou3d_nms.cpp:119-132); returns #kept */
int oracle_nms(const float* boxes, int n, float thresh, int rotated, long long* keep) {
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    int dead = 0;
    for (int k = 0; k < cnt && !dead; k++) {
      const float* kb = boxes + 7 * keep[k];
      float v = rotated ? oracle_iou_bev(kb, boxes + 7 * i) : oracle_iou_normal(kb, boxes + 7 * i);
      if (v > thresh) dead = 1;
    }
    if (!dead) keep[cnt++] = i;
  }
  return cnt;
}
