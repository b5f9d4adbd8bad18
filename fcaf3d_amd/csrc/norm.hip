// Bandwidth-bound feature-matrix ops on gfx950: segmented column statistics (BatchNorm over all
// voxels / InstanceNorm per scene), fused normalise + affine + residual + ReLU/ELU forward and
// backward, max pooling over the k2s2 kernel map, row gather / scatter.
// All reductions are two-level with a fixed summation order (deterministic, no float atomics).
//
// Replaces MinkowskiBatchNorm / MinkowskiInstanceNorm / MinkowskiReLU / MinkowskiELU /
// MinkowskiMaxPooling / MinkowskiPruning feature paths used at me_resnet.py:22-24,63,
// BasicBlock (MinkowskiEngine.modules.resnet_block), fcaf3d_neck_with_head.py:53-54,67-71,76,125.
#include "fc_common.h"
#include "../../include/fcaf3d_hip.h"

#define MAXSEG 64
#define MAXBLOCKS 1024       // partial-sum blocks of the two-level reductions

__device__ static inline int seg_of(const int* seg, int seg_stride, int64_t row) {
  return seg ? seg[row * seg_stride] : 0;
}

// segment range touched by rows [r0,r1) — block-parallel min/max (rows are nearly segment-contiguous)
__device__ static inline void seg_range(const int* seg, int seg_stride, int64_t r0, int64_t r1, int* lo, int* hi,
                                        int* sh /*[2] shared*/) {
  if (!seg) { *lo = 0; *hi = 0; return; }
  if (threadIdx.x == 0) { sh[0] = 0x7fffffff; sh[1] = -1; }
  __syncthreads();
  int l = 0x7fffffff, h = -1;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
    int s = seg[r * seg_stride];
    l = s < l ? s : l;
    h = s > h ? s : h;
  }
  if (h >= 0) { atomicMin(&sh[0], l); atomicMax(&sh[1], h); }
  __syncthreads();
  *lo = sh[0]; *hi = sh[1];
  __syncthreads();
}

// ---- pass 1: partial sums of f(x) per (block, segment, channel) --------------------------------
// mode 0: sum x ; mode 1: sum (x-mean[seg])^2 ; mode 2 (r5): sum x AND sum x^2 in one pass, part layout [block][seg][2][C]
// block = (C/4 lanes of float4) x (row lanes); C % 4 == 0, C <= 1024; rows [b*rpb, (b+1)*rpb)
__global__ void k_stats_partial(const float* __restrict__ x, const int* __restrict__ seg, int seg_stride, int64_t n, int C,
                                int nseg, const float* __restrict__ mean, int mode, int64_t rpb, float* __restrict__ part,
                                float* __restrict__ part_cnt) {
  extern __shared__ float sm[];               // [rl][C] staging for the cross-row-lane reduction
  __shared__ int srange[2];
  const int c4n = C / 4;
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const int nrl = blockDim.x / c4n;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb;
  if (r1 > n) r1 = n;
  int s_lo, s_hi;
  seg_range(seg, seg_stride, r0, r1, &s_lo, &s_hi, srange);
  // a row range inside ONE segment (almost every block: rows are nearly segment-contiguous) needs no per-row segment test —
  // r2: the test is a dependent 4-byte load in front of every row load (InstanceNorm of the 580k-row stem: 1.4 TB/s)
  const bool chk = seg && s_lo != s_hi;
  for (int s = 0; s < nseg; ++s) {
    float4 acc[4];
    float4 sq = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= s_lo && s <= s_hi) {
      float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mode == 1) mu = *reinterpret_cast<const float4*>(mean + (int64_t)s * C + cl * 4);
      for (int64_t rb = r0 + rl; rb < r1; rb += 4 * nrl) {
        float4 vv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                     // the four rows are requested together (see k_norm_bwd_partial)
          const int64_t r = rb + (int64_t)u * nrl;
          ok[u] = r < r1;
          const int64_t rc = ok[u] ? r : r0;
          if (chk) ok[u] = ok[u] && seg[rc * seg_stride] == s;
          vv[u] = *reinterpret_cast<const float4*>(x + rc * C + cl * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
          float4 v = vv[u];
          if (mode == 1) {
            v.x -= mu.x; v.y -= mu.y; v.z -= mu.z; v.w -= mu.w;
            v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w;
          }
          if (mode == 2) { sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w; }
          acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
          cnt += 1.f;
        }
      }
    }
    float4 a;
    a.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x);
    a.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
    a.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z);
    a.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
    __syncthreads();
    *reinterpret_cast<float4*>(&sm[(rl * c4n + cl) * 4]) = a;
    float* smc = sm + nrl * C;
    if (cl == 0) smc[rl] = cnt;
    __syncthreads();
    const int64_t slot = (int64_t)blockIdx.x * nseg + s;
    float* dst = mode == 2 ? part + slot * 2 * C : part + slot * C;
    if (rl == 0) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < nrl; ++j) {
        float4 u = *reinterpret_cast<const float4*>(&sm[(j * c4n + cl) * 4]);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      *reinterpret_cast<float4*>(dst + cl * 4) = t;
      if (cl == 0 && part_cnt) {
        float c = 0.f;
        for (int j = 0; j < nrl; ++j) c += smc[j];
        part_cnt[slot] = c;
      }
    }
    if (mode == 2) {                             // the squares through the same staging buffer
      __syncthreads();
      *reinterpret_cast<float4*>(&sm[(rl * c4n + cl) * 4]) = sq;
      __syncthreads();
      if (rl == 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < nrl; ++j) {
          float4 u = *reinterpret_cast<const float4*>(&sm[(j * c4n + cl) * 4]);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4*>(dst + C + cl * 4) = t;
      }
    }
  }
}

// r5: per-segment mean / biased variance / count from the one-pass partials of k_stats_partial mode 2 ([block][seg][2][C] +
// counts [block][seg]): 4 channels x 64 slices per 256-thread block, fp64 accumulation in a fixed order (the instance norm of the
// stem read its 148 MB input twice — mean, then centred squares — in four launches; now once, in two)
__global__ __launch_bounds__(256) void k_seg_meanvar_final(const float* __restrict__ part, const float* __restrict__ part_cnt,
                                                           int64_t nblocks, int nseg, int C, float* __restrict__ mean,
                                                           float* __restrict__ var, float* __restrict__ cnt_out) {
  __shared__ double r1[64][5], r2[64][5], rc[64];
  const int cgroups = (C + 3) / 4;
  const int s = blockIdx.x / cgroups, cg = blockIdx.x % cgroups;
  const int cl = threadIdx.x & 3, j = threadIdx.x >> 2;
  const int c = cg * 4 + cl;
  double a1 = 0., a2 = 0., ac = 0.;
  if (c < C) {
    for (int64_t b = j; b < nblocks; b += 64) {
      const float* src = part + ((b * nseg + s) * 2) * C;
      a1 += fc_ld(&src[c]);
      a2 += fc_ld(&src[C + c]);
      if (cl == 0) ac += fc_ld(&part_cnt[b * nseg + s]);
    }
  }
  r1[j][cl] = a1; r2[j][cl] = a2;
  if (cl == 0) rc[j] = ac;
  __syncthreads();
  if (j == 0 && c < C) {
    double s1 = 0., s2 = 0., n = 0.;
    for (int q = 0; q < 64; ++q) { s1 += r1[q][cl]; s2 += r2[q][cl]; n += rc[q]; }
    double m = 0., v = 0.;
    if (n > 0.) {
      m = s1 / n;
      v = s2 / n - m * m;
      if (v < 0.) v = 0.;
    }
    mean[(int64_t)s * C + c] = (float)m;
    var[(int64_t)s * C + c] = (float)v;
    if (c == 0) cnt_out[s] = (float)n;
  }
}

// ---- pass 2: sum partials over blocks (fixed order); block = 16 channels x 64 slices -----------------
// mode 0: out = sum / cnt (mean), also writes cnt[seg];  mode 1: out = sum / cnt (biased variance); mode 2: out = sum
// r5: 4 channels x 64 slices = 256 threads per block (r1-r4: 64 channels x 16 slices = 1 024 threads).  These launches are pure
// latency, and a 1 024-thread workgroup needs four free wave slots on every SIMD of ONE compute unit: beside the weight-gradient
// stream (two resident 250-register workgroups per CU) it waited for a whole CU to drain — up to 1.8 ms for a 10 us kernel
// (rocprofv3, r5).  256-thread workgroups fit next to anything; 64 slices keep the chain per thread at nb / 256 loads.
#define FIN_CB 4
#define FIN_SL 64
__global__ __launch_bounds__(256) void k_stats_final(const float* __restrict__ part, const float* __restrict__ part_cnt,
                                                      int64_t nblocks, int nseg, int C, int mode, float* __restrict__ out,
                                                      float* __restrict__ cnt_io) {
  __shared__ float red[FIN_SL][FIN_CB + 1];
  __shared__ float redc[FIN_SL];
  const int cgroups = (C + FIN_CB - 1) / FIN_CB;
  const int s = blockIdx.x / cgroups, cg = blockIdx.x % cgroups;
  const int cl = threadIdx.x & (FIN_CB - 1), j = threadIdx.x / FIN_CB;
  const int c = cg * FIN_CB + cl;
  // four partials in flight per thread (r3: one dependent load per iteration made these tiny launches 13 us each); the
  // summation order is fixed by the code, hence still deterministic
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C) {
    int64_t b = j;
    for (; b + 3 * FIN_SL < nblocks; b += 4 * FIN_SL) {
      const float v0 = fc_ld(&part[(b * nseg + s) * C + c]), v1 = fc_ld(&part[((b + FIN_SL) * nseg + s) * C + c]);      // (one block reads the
      const float v2 = fc_ld(&part[((b + 2 * FIN_SL) * nseg + s) * C + c]), v3 = fc_ld(&part[((b + 3 * FIN_SL) * nseg + s) * C + c]);   //  table once: past the L1)
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; b < nblocks; b += FIN_SL) a0 += fc_ld(&part[(b * nseg + s) * C + c]);
  }
  float acc = (a0 + a1) + (a2 + a3);
  red[j][cl] = acc;
  if (mode == 0 && cl == 0) {
    float cc = 0.f;
    for (int64_t b = j; b < nblocks; b += FIN_SL) cc += fc_ld(&part_cnt[b * nseg + s]);
    redc[j] = cc;
  }
  __syncthreads();
  if (j == 0 && c < C) {
    float t = 0.f;
    for (int q = 0; q < FIN_SL; ++q) t += red[q][cl];
    float cnt = 1.f;
    if (mode == 0) {
      float cc = 0.f;
      for (int q = 0; q < FIN_SL; ++q) cc += redc[q];
      if (c == 0) cnt_io[s] = cc;
      cnt = cc;
    } else if (mode == 1) {
      cnt = cnt_io[s];
    }
    out[(int64_t)s * C + c] = (mode == 2) ? t : (cnt > 0.f ? t / cnt : 0.f);
  }
}

__device__ static inline float act_fwd(float v, int act) {
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return v > 0.f ? v : expm1f(v);
  return v;
}
// derivative expressed through the OUTPUT y (ReLU: y>0 ; ELU(alpha=1): y>0 ? 1 : y+1)
__device__ static inline float act_bwd_from_y(float y, int act) {
  if (act == 1) return y > 0.f ? 1.f : 0.f;
  if (act == 2) return y > 0.f ? 1.f : y + 1.f;
  return 1.f;
}

// The pre-activation of a normalised element, ONE instruction sequence for the forward kernels and for the backward
// kernels that recompute it (r3: without a residual the backward pass derives act'(.) from x instead of reading y —
// 2 of 7 passes over the matrix gone; the explicit fmaf keeps the recomputed sign bit-identical to the forward's)
__device__ static inline float bn_pre(float v, float mu, float is, float g, float b) { return fmaf((v - mu) * is, g, b); }
// derivative of the activation from its PRE-activation p (ReLU: p > 0; ELU(alpha=1): p > 0 ? 1 : exp(p) = y + 1)
__device__ static inline float act_bwd_from_pre(float p, int act) {
  if (act == 1) return p > 0.f ? 1.f : 0.f;
  if (act == 2) return p > 0.f ? 1.f : expf(p);
  return 1.f;
}

// ---- r6: max |.| of what a kernel writes, for the convolution that will gather it (csrc/conv_x6.h h3: the operand's scale) ----------
// A producer that is told where (fc_amax_out_hint -> amax_out, a ZEROED slot of FC_AMAX_SUB sub-words) folds the finite elements it
// stores into one integer atomicMax per block on its block's sub-word — skipped when that already holds a larger value — instead of
// the consumer reading the tensor once more (fc_amax).  Integer max: order-independent, bit-reproducible.  EVERY lane of the wave must reach amax_commit.
thread_local unsigned* t_fc_amax_out = nullptr;          // fc_amax_out_hint: consumed by the next producer entry point of this thread (fc_common.h)

// y = act( (x-mean[seg])*invstd[seg]*gamma + beta (+ residual) ) ; invstd = 1/sqrt(var+eps)
__global__ void k_norm_act_fwd(const float* __restrict__ x, const int* __restrict__ seg, int seg_stride, int64_t n, int C,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ residual, int act, float* __restrict__ y, unsigned* __restrict__ amax_out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = C / 4;
  unsigned am = 0u;
  if (t < n * c4n) {
  int64_t r = t / c4n;
  int c = (int)(t % c4n) * 4;
  int s = seg_of(seg, seg_stride, r);
  float4 v = *reinterpret_cast<const float4*>(x + r * C + c);
  float4 mu = *reinterpret_cast<const float4*>(mean + (int64_t)s * C + c);
  float4 va = *reinterpret_cast<const float4*>(var + (int64_t)s * C + c);
  float4 g = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  float4 b = beta ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 o;
  o.x = bn_pre(v.x, mu.x, 1.f / sqrtf(va.x + eps), g.x, b.x);
  o.y = bn_pre(v.y, mu.y, 1.f / sqrtf(va.y + eps), g.y, b.y);
  o.z = bn_pre(v.z, mu.z, 1.f / sqrtf(va.z + eps), g.z, b.z);
  o.w = bn_pre(v.w, mu.w, 1.f / sqrtf(va.w + eps), g.w, b.w);
  if (residual) {
    float4 rs = *reinterpret_cast<const float4*>(residual + r * C + c);
    o.x += rs.x; o.y += rs.y; o.z += rs.z; o.w += rs.w;
  }
  o.x = act_fwd(o.x, act); o.y = act_fwd(o.y, act); o.z = act_fwd(o.z, act); o.w = act_fwd(o.w, act);
  *reinterpret_cast<float4*>(y + r * C + c) = o;
  amax_fold(am, o.x); amax_fold(am, o.y); amax_fold(am, o.z); amax_fold(am, o.w);
  }
  if (amax_out) amax_commit(am, amax_out);
}

// backward pass 1: per (block, seg, channel) partial sums of g' and g'*xhat, g' = gy * act'(y)
// part layout: [block][seg][2][C]
// y == NULL (no residual in the forward): act'(.) is recomputed from x with gamma / beta (bn_pre) instead of read from y
// gy2 (nullable, r5): a second contribution to the incoming gradient — g = gy + gy2 is formed on the fly here and, with the same
// operands in the same order, in the apply kernels: the executor no longer materialises the sum where a normalised tensor has
// two consumers (k_add_inplace: 22 launches per step in r4)
__global__ void k_norm_bwd_partial(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy,
                                   const float* __restrict__ gy2, const int* __restrict__ seg, int seg_stride, int64_t n, int C, int nseg,
                                   const float* __restrict__ mean, const float* __restrict__ var, float eps, int act,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   int64_t rpb, float* __restrict__ part) {
  extern __shared__ float sm[];               // [rl][2][C]
  __shared__ int srange[2];
  const int c4n = C / 4;
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const int nrl = blockDim.x / c4n;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb;
  if (r1 > n) r1 = n;
  int s_lo, s_hi;
  seg_range(seg, seg_stride, r0, r1, &s_lo, &s_hi, srange);
  // a row range inside ONE segment (almost every block: rows are nearly segment-contiguous) needs no per-row segment test —
  // r2: the test is a dependent 4-byte load in front of every row load (InstanceNorm of the 580k-row stem: 1.4 TB/s)
  const bool chk = seg && s_lo != s_hi;
  for (int s = 0; s < nseg; ++s) {
    float4 a1[4], a2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a1[u] = make_float4(0.f, 0.f, 0.f, 0.f); a2[u] = a1[u]; }
    if (s >= s_lo && s <= s_hi) {
      float4 mu = *reinterpret_cast<const float4*>(mean + (int64_t)s * C + cl * 4);
      float4 va = *reinterpret_cast<const float4*>(var + (int64_t)s * C + cl * 4);
      float4 is = make_float4(1.f / sqrtf(va.x + eps), 1.f / sqrtf(va.y + eps), 1.f / sqrtf(va.z + eps),
                              1.f / sqrtf(va.w + eps));
      const float4 gm = gamma ? *reinterpret_cast<const float4*>(gamma + cl * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 bt = beta ? *reinterpret_cast<const float4*>(beta + cl * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const bool from_y = act && y;
      for (int64_t rb = r0 + rl; rb < r1; rb += 4 * nrl) {
        // branch-free batch: the 4 x (x, gy, y) rows are requested together (rows past the range re-read row r0 and are
        // dropped by `ok`), r3 — a `continue` in front of each load serialised them
        float4 xv[4], gv[4], yv[4], g2[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t r = rb + (int64_t)u * nrl;
          ok[u] = r < r1;
          const int64_t rc = ok[u] ? r : r0;
          if (chk) ok[u] = ok[u] && seg[rc * seg_stride] == s;
          xv[u] = *reinterpret_cast<const float4*>(x + rc * C + cl * 4);
          gv[u] = *reinterpret_cast<const float4*>(gy + rc * C + cl * 4);
          if (gy2) g2[u] = *reinterpret_cast<const float4*>(gy2 + rc * C + cl * 4);
          if (from_y) yv[u] = *reinterpret_cast<const float4*>(y + rc * C + cl * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
          float4 g = gv[u];
          if (gy2) { g.x += g2[u].x; g.y += g2[u].y; g.z += g2[u].z; g.w += g2[u].w; }
          if (from_y) {
            g.x *= act_bwd_from_y(yv[u].x, act); g.y *= act_bwd_from_y(yv[u].y, act);
            g.z *= act_bwd_from_y(yv[u].z, act); g.w *= act_bwd_from_y(yv[u].w, act);
          } else if (act) {
            g.x *= act_bwd_from_pre(bn_pre(xv[u].x, mu.x, is.x, gm.x, bt.x), act);
            g.y *= act_bwd_from_pre(bn_pre(xv[u].y, mu.y, is.y, gm.y, bt.y), act);
            g.z *= act_bwd_from_pre(bn_pre(xv[u].z, mu.z, is.z, gm.z, bt.z), act);
            g.w *= act_bwd_from_pre(bn_pre(xv[u].w, mu.w, is.w, gm.w, bt.w), act);
          }
          a1[u].x += g.x; a1[u].y += g.y; a1[u].z += g.z; a1[u].w += g.w;
          a2[u].x += g.x * (xv[u].x - mu.x) * is.x; a2[u].y += g.y * (xv[u].y - mu.y) * is.y;
          a2[u].z += g.z * (xv[u].z - mu.z) * is.z; a2[u].w += g.w * (xv[u].w - mu.w) * is.w;
        }
      }
    }
    float4 b1 = make_float4((a1[0].x + a1[1].x) + (a1[2].x + a1[3].x), (a1[0].y + a1[1].y) + (a1[2].y + a1[3].y),
                            (a1[0].z + a1[1].z) + (a1[2].z + a1[3].z), (a1[0].w + a1[1].w) + (a1[2].w + a1[3].w));
    float4 b2 = make_float4((a2[0].x + a2[1].x) + (a2[2].x + a2[3].x), (a2[0].y + a2[1].y) + (a2[2].y + a2[3].y),
                            (a2[0].z + a2[1].z) + (a2[2].z + a2[3].z), (a2[0].w + a2[1].w) + (a2[2].w + a2[3].w));
    __syncthreads();
    *reinterpret_cast<float4*>(&sm[(rl * 2 + 0) * C + cl * 4]) = b1;
    *reinterpret_cast<float4*>(&sm[(rl * 2 + 1) * C + cl * 4]) = b2;
    __syncthreads();
    if (rl == 0) {
      float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
      for (int j = 0; j < nrl; ++j) {
        float4 u = *reinterpret_cast<const float4*>(&sm[(j * 2 + 0) * C + cl * 4]);
        float4 w = *reinterpret_cast<const float4*>(&sm[(j * 2 + 1) * C + cl * 4]);
        t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
        t2.x += w.x; t2.y += w.y; t2.z += w.z; t2.w += w.w;
      }
      float* dst = part + (((int64_t)blockIdx.x * nseg + s) * 2) * C;
      *reinterpret_cast<float4*>(dst + cl * 4) = t1;
      *reinterpret_cast<float4*>(dst + C + cl * 4) = t2;
    }
  }
}

// backward pass 3:  gx = gamma*invstd*( g' - sum_g/cnt - xhat * sum_gx/cnt ) ; gres = g'
__global__ void k_norm_bwd_apply(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy,
                                 const float* __restrict__ gy2, const int* __restrict__ seg, int seg_stride, int64_t n, int C,
                                 const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ sums /*[seg][2][C]*/,
                                 const float* __restrict__ cnt, int act, float* __restrict__ gx, float* __restrict__ gres,
                                 unsigned* __restrict__ amax_out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = C / 4;
  unsigned am = 0u;
  if (t < n * c4n) {
  int64_t r = t / c4n;
  int c = (int)(t % c4n) * 4;
  int s = seg_of(seg, seg_stride, r);
  float inv_n = 1.f / cnt[s];
  float xv[4], gv[4], muv[4], vav[4], gam[4], bet[4] = {0.f, 0.f, 0.f, 0.f}, s1[4], s2[4], yv[4], o[4], gr[4];
  *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(x + r * C + c);
  *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(gy + r * C + c);
  if (gy2) {
    const float4 t2 = *reinterpret_cast<const float4*>(gy2 + r * C + c);
    gv[0] += t2.x; gv[1] += t2.y; gv[2] += t2.z; gv[3] += t2.w;
  }
  *reinterpret_cast<float4*>(muv) = *reinterpret_cast<const float4*>(mean + (int64_t)s * C + c);
  *reinterpret_cast<float4*>(vav) = *reinterpret_cast<const float4*>(var + (int64_t)s * C + c);
  *reinterpret_cast<float4*>(s1) = *reinterpret_cast<const float4*>(sums + ((int64_t)s * 2) * C + c);
  *reinterpret_cast<float4*>(s2) = *reinterpret_cast<const float4*>(sums + ((int64_t)s * 2 + 1) * C + c);
  if (gamma) *reinterpret_cast<float4*>(gam) = *reinterpret_cast<const float4*>(gamma + c);
  else gam[0] = gam[1] = gam[2] = gam[3] = 1.f;
  const bool from_y = act && y;
  if (from_y) *reinterpret_cast<float4*>(yv) = *reinterpret_cast<const float4*>(y + r * C + c);
  if (beta) *reinterpret_cast<float4*>(bet) = *reinterpret_cast<const float4*>(beta + c);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float g = gv[j];
    float is = 1.f / sqrtf(vav[j] + eps);
    if (from_y) g *= act_bwd_from_y(yv[j], act);
    else if (act) g *= act_bwd_from_pre(bn_pre(xv[j], muv[j], is, gam[j], bet[j]), act);
    float xh = (xv[j] - muv[j]) * is;
    gr[j] = g;
    o[j] = gam[j] * is * (g - s1[j] * inv_n - xh * s2[j] * inv_n);
  }
  *reinterpret_cast<float4*>(gx + r * C + c) = *reinterpret_cast<float4*>(o);
  if (gres) *reinterpret_cast<float4*>(gres + r * C + c) = *reinterpret_cast<float4*>(gr);
#pragma unroll
  for (int j = 0; j < 4; ++j) amax_fold(am, o[j]);
  }
  if (amax_out) amax_commit(am, amax_out);
}

// ---- max pooling over a (K, n_out) neighbour table ---------------------------------------------
__global__ void k_maxpool_fwd(const float* __restrict__ in, const int* __restrict__ nbr, int64_t n_out, int K, int C,
                              float* __restrict__ out, int* __restrict__ argrow) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * C) return;
  int64_t o = t / C;
  int c = (int)(t % C);
  float best = -INFINITY;
  int arg = -1;
  for (int k = 0; k < K; ++k) {
    int i = nbr[(int64_t)k * n_out + o];
    if (i < 0) continue;
    float v = in[(int64_t)i * C + c];
    if (arg < 0 || v > best) { best = v; arg = i; }   // first max in offset order wins ties (A.5)
  }
  out[t] = arg < 0 ? 0.f : best;
  argrow[t] = arg;
}

// k2s2 (K == 8), C % 4 == 0: one thread per (output row, 4 channels); the 8 child rows are looked up first, then their
// 8 x 16 B are in flight together (the scalar kernel above keeps one dependent 4-byte load per lane in flight: 1.6 TB/s)
__global__ void k_maxpool8_fwd(const float* __restrict__ in, const int* __restrict__ nbr, int64_t n_out, int C,
                               float* __restrict__ out, int* __restrict__ argrow, unsigned* __restrict__ amax_out) {
  const int c4n = C / 4;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned am = 0u;
  if (t < n_out * c4n) {
  const int64_t o = t / c4n;
  const int c = (int)(t % c4n) * 4;
  int idx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) idx[k] = nbr[(int64_t)k * n_out + o];
  float4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = idx[k] < 0 ? 0 : idx[k];
    v[k] = *reinterpret_cast<const float4*>(in + (int64_t)i * C + c);
  }
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int arg[4] = {-1, -1, -1, -1};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (idx[k] < 0) continue;
    const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (arg[j] < 0 || e[j] > best[j]) { best[j] = e[j]; arg[j] = idx[k]; }   // first max in offset order wins ties (A.5)
  }
  float4 ob = make_float4(arg[0] < 0 ? 0.f : best[0], arg[1] < 0 ? 0.f : best[1], arg[2] < 0 ? 0.f : best[2], arg[3] < 0 ? 0.f : best[3]);
  *reinterpret_cast<float4*>(out + o * C + c) = ob;
  *reinterpret_cast<int4*>(argrow + o * C + c) = make_int4(arg[0], arg[1], arg[2], arg[3]);
  amax_fold(am, ob.x); amax_fold(am, ob.y); amax_fold(am, ob.z); amax_fold(am, ob.w);
  }
  if (amax_out) amax_commit(am, amax_out);
}

__global__ void k_maxpool_bwd(const float* __restrict__ gout, const int* __restrict__ argrow, int64_t n_out, int C,
                              float* __restrict__ gin) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * C) return;
  int i = argrow[t];
  // k2s2 children are disjoint between output cells -> each (row, channel) receives at most one write
  if (i >= 0) gin[(int64_t)i * C + (t % C)] = gout[t];
}

// ---- row gather / scatter ----------------------------------------------------------------------
__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ idx, int64_t n, int C,
                              float* __restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * C) return;
  int64_t r = t / C;
  int i = idx[r];
  dst[t] = i >= 0 ? src[(int64_t)i * C + (t % C)] : 0.f;
}

// dst[idx[r]] += src[r]   (idx unique -> race free)
__global__ void k_scatter_rows_add(const float* __restrict__ src, const int* __restrict__ idx, int64_t n, int C,
                                   float* __restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * C) return;
  int64_t r = t / C;
  int i = idx[r];
  if (i >= 0) dst[(int64_t)i * C + (t % C)] += src[t];
}

// ================================================================================================
// Small-tensor BatchNorm (n*C <= 4 M elements: backbone levels 2-4, coarse neck levels): TWO launches per
// direction instead of six / three.  Launch 1 writes per-block partial sums; launch 2 lets every block
// re-reduce the (<= 64) partials in its prologue (a few hundred KB out of L2) and apply straight away.
// Forward statistics in one pass: sums of (x - s) and (x - s)^2 with the shift s = x[0][c] (a sample of the
// column, so |mean - s| ~ sigma and E[(x-s)^2] - E[x-s]^2 loses no digits).
#define BN1_MAXB 64

// part [nb][2][C]
__global__ void k_bn1_partial(const float* __restrict__ x, int64_t n, int C, int64_t rpb, float* __restrict__ part) {
  extern __shared__ float sm[];               // [rl][2][C]
  const int c4n = C / 4;
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const int nrl = blockDim.x / c4n;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb;
  if (r1 > n) r1 = n;
  const float4 sh = *reinterpret_cast<const float4*>(x + cl * 4);
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  for (int64_t rb = r0 + rl; rb < r1; rb += 4 * (int64_t)nrl) {      // four rows in flight per thread (padding rows read the shift: 0)
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = rb + (int64_t)u * nrl;
      v[u] = r < r1 ? *reinterpret_cast<const float4*>(x + r * C + cl * 4) : sh;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u].x -= sh.x; v[u].y -= sh.y; v[u].z -= sh.z; v[u].w -= sh.w;
      a1.x += v[u].x; a1.y += v[u].y; a1.z += v[u].z; a1.w += v[u].w;
      a2.x += v[u].x * v[u].x; a2.y += v[u].y * v[u].y; a2.z += v[u].z * v[u].z; a2.w += v[u].w * v[u].w;
    }
  }
  *reinterpret_cast<float4*>(&sm[(rl * 2 + 0) * C + cl * 4]) = a1;
  *reinterpret_cast<float4*>(&sm[(rl * 2 + 1) * C + cl * 4]) = a2;
  __syncthreads();
  if (rl == 0) {
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
    for (int j = 0; j < nrl; ++j) {
      float4 u = *reinterpret_cast<const float4*>(&sm[(j * 2 + 0) * C + cl * 4]);
      float4 w = *reinterpret_cast<const float4*>(&sm[(j * 2 + 1) * C + cl * 4]);
      t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
      t2.x += w.x; t2.y += w.y; t2.z += w.z; t2.w += w.w;
    }
    float* dst = part + ((int64_t)blockIdx.x * 2) * C;
    *reinterpret_cast<float4*>(dst + cl * 4) = t1;
    *reinterpret_cast<float4*>(dst + C + cl * 4) = t2;
  }
}

// every block: reduce part[nb][2][C] -> (S1,S2) per channel (fixed order), then apply to its rows
// (CG channels from c0 on: a block may own a channel window of the table, k_bn1_bwd_apply's grid.y; CG == C, c0 == 0: all of them)
__device__ static inline void bn1_reduce_parts(const float* __restrict__ part, int nb, int C, int cl, int rl, int nrl,
                                               float* sm /*[nrl][2][CG]*/, float4* s1, float4* s2, int CG, int c0) {
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  // a SMALL table (the deep levels: a few KB at the same workspace address layer after layer) can survive in a CU's L1
  // from the previous layer: read it past the L1 (fc_common.h: fc_ld4); a big one streams through the L1 anyway
  const bool past_l1 = (int64_t)nb * C * 8 <= 32768;
  for (int b = rl; b < nb; b += nrl) {
    const float* src = part + ((int64_t)b * 2) * C + c0;
    float4 u = past_l1 ? fc_ld4(src + cl * 4) : *reinterpret_cast<const float4*>(src + cl * 4);
    float4 w = past_l1 ? fc_ld4(src + C + cl * 4) : *reinterpret_cast<const float4*>(src + C + cl * 4);
    a1.x += u.x; a1.y += u.y; a1.z += u.z; a1.w += u.w;
    a2.x += w.x; a2.y += w.y; a2.z += w.z; a2.w += w.w;
  }
  *reinterpret_cast<float4*>(&sm[(rl * 2 + 0) * CG + cl * 4]) = a1;
  *reinterpret_cast<float4*>(&sm[(rl * 2 + 1) * CG + cl * 4]) = a2;
  __syncthreads();
  float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
  for (int j = 0; j < nrl; ++j) {
    float4 u = *reinterpret_cast<const float4*>(&sm[(j * 2 + 0) * CG + cl * 4]);
    float4 w = *reinterpret_cast<const float4*>(&sm[(j * 2 + 1) * CG + cl * 4]);
    t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
    t2.x += w.x; t2.y += w.y; t2.z += w.z; t2.w += w.w;
  }
  *s1 = t1; *s2 = t2;
}

__global__ void k_bn1_apply(const float* __restrict__ x, int64_t n, int C, int64_t rpb, const float* __restrict__ part, int nb,
                            float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                            const float* __restrict__ residual, int act, float momentum, float* __restrict__ y,
                            float* __restrict__ mean_out, float* __restrict__ var_out, float* __restrict__ cnt_out,
                            float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt) {
  extern __shared__ float sm[];
  const int c4n = C / 4;
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const int nrl = blockDim.x / c4n;
  float4 s1, s2;
  bn1_reduce_parts(part, nb, C, cl, rl, nrl, sm, &s1, &s2, C, 0);
  const float4 sh = *reinterpret_cast<const float4*>(x + cl * 4);
  const float inv_n = 1.f / (float)n;
  float d[4] = {s1.x * inv_n, s1.y * inv_n, s1.z * inv_n, s1.w * inv_n};
  float q[4] = {s2.x * inv_n, s2.y * inv_n, s2.z * inv_n, s2.w * inv_n};
  float shv[4] = {sh.x, sh.y, sh.z, sh.w};
  float mu[4], va[4], is[4], g[4], bt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = shv[j] + d[j];
    va[j] = fmaxf(q[j] - d[j] * d[j], 0.f);
    is[j] = 1.f / sqrtf(va[j] + eps);
    g[j] = gamma ? gamma[cl * 4 + j] : 1.f;
    bt[j] = beta ? beta[cl * 4 + j] : 0.f;
  }
  if (blockIdx.x == 0 && rl == 0) {
    *reinterpret_cast<float4*>(mean_out + cl * 4) = make_float4(mu[0], mu[1], mu[2], mu[3]);
    *reinterpret_cast<float4*>(var_out + cl * 4) = make_float4(va[0], va[1], va[2], va[3]);
    if (cl == 0) { cnt_out[0] = (float)n; if (nbt) nbt[0] += 1; }
    if (rmean) {
      float unbias = (float)n / fmaxf((float)n - 1.f, 1.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rmean[cl * 4 + j] = (1.f - momentum) * rmean[cl * 4 + j] + momentum * mu[j];
        rvar[cl * 4 + j] = (1.f - momentum) * rvar[cl * 4 + j] + momentum * va[j] * unbias;
      }
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb;
  if (r1 > n) r1 = n;
  for (int64_t rb = r0 + rl; rb < r1; rb += 4 * (int64_t)nrl) {      // four rows in flight per thread
    float v[4][4], rs[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = rb + (int64_t)u * nrl;
      const int64_t rc = r < r1 ? r : r0;
      *reinterpret_cast<float4*>(v[u]) = *reinterpret_cast<const float4*>(x + rc * C + cl * 4);
      if (residual) *reinterpret_cast<float4*>(rs[u]) = *reinterpret_cast<const float4*>(residual + rc * C + cl * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = rb + (int64_t)u * nrl;
      if (r >= r1) continue;
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = bn_pre(v[u][j], mu[j], is[j], g[j], bt[j]);
        if (residual) o[j] += rs[u][j];
        o[j] = act_fwd(o[j], act);
      }
      *reinterpret_cast<float4*>(y + r * C + cl * 4) = *reinterpret_cast<float4*>(o);
    }
  }
}

// general-size BatchNorm statistics, finalisation: one 1024-thread block per 64 channels reduces the
// [nb][2][C] shifted partial sums of k_bn1_partial (fixed order), writes mean / biased var / count and applies the
// nn.BatchNorm1d running-buffer update — 3 launches per training-mode BatchNorm forward instead of 6.
__global__ __launch_bounds__(1024) void k_bn_finalize(const float* __restrict__ part, int nb, int C, int64_t n,
                                                      const float* __restrict__ x, float momentum, float* __restrict__ mean,
                                                      float* __restrict__ var, float* __restrict__ cnt,
                                                      float* __restrict__ rmean, float* __restrict__ rvar,
                                                      long long* __restrict__ nbt) {
  __shared__ float r1[16][65], r2[16][65];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), j = threadIdx.x >> 6;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    float p1[4] = {0.f, 0.f, 0.f, 0.f}, p2[4] = {0.f, 0.f, 0.f, 0.f};      // four blocks' partials in flight (see k_stats_final)
    int b = j;
    for (; b + 48 < nb; b += 64) {
      float u[4], w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u[q] = fc_ld(&part[((int64_t)(b + 16 * q) * 2) * C + c]);
        w[q] = fc_ld(&part[((int64_t)(b + 16 * q) * 2 + 1) * C + c]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { p1[q] += u[q]; p2[q] += w[q]; }
    }
    for (; b < nb; b += 16) {
      p1[0] += fc_ld(&part[((int64_t)b * 2) * C + c]);
      p2[0] += fc_ld(&part[((int64_t)b * 2 + 1) * C + c]);
    }
    a1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
    a2 = (p2[0] + p2[1]) + (p2[2] + p2[3]);
  }
  r1[j][threadIdx.x & 63] = a1;
  r2[j][threadIdx.x & 63] = a2;
  __syncthreads();
  if (j == 0 && c < C) {
    float s1 = 0.f, s2 = 0.f;
    for (int q = 0; q < 16; ++q) { s1 += r1[q][threadIdx.x & 63]; s2 += r2[q][threadIdx.x & 63]; }
    const float inv_n = 1.f / (float)n;
    float d = s1 * inv_n;
    float mu = x[c] + d;
    float va = fmaxf(s2 * inv_n - d * d, 0.f);
    mean[c] = mu;
    var[c] = va;
    if (c == 0) { cnt[0] = (float)n; if (nbt) nbt[0] += 1; }
    if (rmean) {
      float unbias = (float)n / fmaxf((float)n - 1.f, 1.f);
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * va * unbias;
    }
  }
}

// backward launch 2: reduce the partials of k_norm_bwd_partial (nseg = 1, layout [nb][1][2][C]) and apply
__global__ void k_bn1_bwd_apply(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy,
                                const float* __restrict__ gy2, int64_t n, int C, int64_t rpb, const float* __restrict__ part, int nb,
                                const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                float* __restrict__ gx, float* __restrict__ gres, float* __restrict__ sums /*[2][C]*/, int CG,
                                unsigned* __restrict__ amax_out) {
  // r6: grid.y channel windows of CG channels (few rows x many channels — 872 x 512 — used to be 14 blocks of 2 row lanes, 49 us)
  extern __shared__ float sm[];
  const int c0 = blockIdx.y * CG;
  x += c0; gy += c0; gx += c0; mean += c0; var += c0;
  if (y) y += c0;
  if (gy2) gy2 += c0;
  if (gres) gres += c0;
  if (gamma) gamma += c0;
  if (beta) beta += c0;
  const int c4n = CG / 4;
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const int nrl = blockDim.x / c4n;
  float4 t1, t2;
  bn1_reduce_parts(part, nb, C, cl, rl, nrl, sm, &t1, &t2, CG, c0);
  if (blockIdx.x == 0 && rl == 0) {
    *reinterpret_cast<float4*>(sums + c0 + cl * 4) = t1;
    *reinterpret_cast<float4*>(sums + C + c0 + cl * 4) = t2;
  }
  const float inv_n = 1.f / (float)n;
  float s1[4] = {t1.x * inv_n, t1.y * inv_n, t1.z * inv_n, t1.w * inv_n};
  float s2[4] = {t2.x * inv_n, t2.y * inv_n, t2.z * inv_n, t2.w * inv_n};
  float mu[4], is[4], g[4], bt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = mean[cl * 4 + j];
    is[j] = 1.f / sqrtf(var[cl * 4 + j] + eps);
    g[j] = gamma ? gamma[cl * 4 + j] : 1.f;
    bt[j] = beta ? beta[cl * 4 + j] : 0.f;
  }
  const bool from_y = act && y;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb;
  if (r1 > n) r1 = n;
  unsigned am = 0u;
  for (int64_t rb = r0 + rl; rb < r1; rb += 4 * (int64_t)nrl) {      // four rows (x, gy, y of each) in flight per thread
    float xv[4][4], gv[4][4], yv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = rb + (int64_t)u * nrl;
      const int64_t rc = r < r1 ? r : r0;
      *reinterpret_cast<float4*>(xv[u]) = *reinterpret_cast<const float4*>(x + rc * C + cl * 4);
      *reinterpret_cast<float4*>(gv[u]) = *reinterpret_cast<const float4*>(gy + rc * C + cl * 4);
      if (gy2) {
        const float4 t2 = *reinterpret_cast<const float4*>(gy2 + rc * C + cl * 4);
        gv[u][0] += t2.x; gv[u][1] += t2.y; gv[u][2] += t2.z; gv[u][3] += t2.w;
      }
      if (from_y) *reinterpret_cast<float4*>(yv[u]) = *reinterpret_cast<const float4*>(y + rc * C + cl * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = rb + (int64_t)u * nrl;
      if (r >= r1) continue;
      float o[4], gr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gg = gv[u][j];
        if (from_y) gg *= act_bwd_from_y(yv[u][j], act);
        else if (act) gg *= act_bwd_from_pre(bn_pre(xv[u][j], mu[j], is[j], g[j], bt[j]), act);
        float xh = (xv[u][j] - mu[j]) * is[j];
        gr[j] = gg;
        o[j] = g[j] * is[j] * (gg - s1[j] - xh * s2[j]);
      }
      *reinterpret_cast<float4*>(gx + r * C + cl * 4) = *reinterpret_cast<float4*>(o);
      if (gres) *reinterpret_cast<float4*>(gres + r * C + cl * 4) = *reinterpret_cast<float4*>(gr);
#pragma unroll
      for (int j = 0; j < 4; ++j) amax_fold(am, o[j]);
    }
  }
  if (amax_out) amax_commit(am, amax_out);
}

// ================================================================================================
// r5: BatchNorm statistics out of the PRODUCER's epilogue.  The kernel that writes a convolution's result (the MFMA tile epilogue
// of k_conv_x6, or the fixed-order sums k_sum_parts_stats / k_sum_pairs_stats of an offset-split / pair-list launch) also leaves,
// per row block, the column sums of x and x^2 of the rows it wrote: part[nb][2][G * C] (plain fp32 sums over <= a few hundred
// rows; G > 1: the (n, G C) matrix of a generative transposed convolution's GEMM viewed as (G n, C): channel c collects the
// columns g C + c).  They are combined in fp64 in a fixed order — mean = S1 / n, var = S2 / n - mean^2 — either in the prologue
// of the apply kernel (<= BN1_MAXB partial blocks: ONE launch per BatchNorm) or by k_bn2_finalize (two launches).  The
// read pass over x of k_bn1_partial is gone (r4: 46 launches, 0.32 ms alone / 0.56 ms beside the weight-gradient stream).
__device__ static inline void bn2_stats(double s1, double s2, int64_t n, float* mu, float* va) {
  const double m = s1 / (double)n;
  double v = s2 / (double)n - m * m;
  if (v < 0.0) v = 0.0;
  *mu = (float)m;
  *va = (float)v;
}

__global__ void k_bn2_apply(const float* __restrict__ x, int64_t n, int C, int64_t rpb, const float* __restrict__ part, int nb, int G,
                            float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                            const float* __restrict__ residual, int act, float momentum, float* __restrict__ y,
                            float* __restrict__ mean_out, float* __restrict__ var_out, float* __restrict__ cnt_out,
                            float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt, int CG,
                            unsigned* __restrict__ amax_out) {
  // r6: grid.y channel windows of CG channels (see k_bn1_bwd_apply)
  extern __shared__ double smd[];             // [nrl][2][CG]
  const int c0 = blockIdx.y * CG;
  x += c0; y += c0; part += c0; mean_out += c0; var_out += c0;
  if (residual) residual += c0;
  if (gamma) gamma += c0;
  if (beta) beta += c0;
  if (rmean) { rmean += c0; rvar += c0; }
  const int c4n = CG / 4;
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const int nrl = blockDim.x / c4n;
  double a1[4] = {0., 0., 0., 0.}, a2[4] = {0., 0., 0., 0.};
  const int GC = G * C;
  for (int t = rl; t < nb * G; t += nrl) {            // (block, group) pairs, fixed order per row lane
    const int b = t / G, g = t % G;
    const float* src = part + ((int64_t)b * 2) * GC + g * C + cl * 4;
    const float4 u = fc_ld4(src), w = fc_ld4(src + GC);
    a1[0] += u.x; a1[1] += u.y; a1[2] += u.z; a1[3] += u.w;
    a2[0] += w.x; a2[1] += w.y; a2[2] += w.z; a2[3] += w.w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    smd[(rl * 2 + 0) * CG + cl * 4 + j] = a1[j];
    smd[(rl * 2 + 1) * CG + cl * 4 + j] = a2[j];
  }
  __syncthreads();
  float mu[4], va[4], is[4], g[4], bt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double t1 = 0., t2 = 0.;
    for (int q = 0; q < nrl; ++q) { t1 += smd[(q * 2 + 0) * CG + cl * 4 + j]; t2 += smd[(q * 2 + 1) * CG + cl * 4 + j]; }
    bn2_stats(t1, t2, n, &mu[j], &va[j]);
    is[j] = 1.f / sqrtf(va[j] + eps);
    g[j] = gamma ? gamma[cl * 4 + j] : 1.f;
    bt[j] = beta ? beta[cl * 4 + j] : 0.f;
  }
  if (blockIdx.x == 0 && rl == 0) {
    *reinterpret_cast<float4*>(mean_out + cl * 4) = make_float4(mu[0], mu[1], mu[2], mu[3]);
    *reinterpret_cast<float4*>(var_out + cl * 4) = make_float4(va[0], va[1], va[2], va[3]);
    if (cl == 0 && blockIdx.y == 0) { cnt_out[0] = (float)n; if (nbt) nbt[0] += 1; }
    if (rmean) {
      float unbias = (float)n / fmaxf((float)n - 1.f, 1.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rmean[cl * 4 + j] = (1.f - momentum) * rmean[cl * 4 + j] + momentum * mu[j];
        rvar[cl * 4 + j] = (1.f - momentum) * rvar[cl * 4 + j] + momentum * va[j] * unbias;
      }
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb;
  if (r1 > n) r1 = n;
  unsigned am = 0u;
  for (int64_t rb = r0 + rl; rb < r1; rb += 4 * (int64_t)nrl) {      // four rows in flight per thread
    float v[4][4], rs[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = rb + (int64_t)u * nrl;
      const int64_t rc = r < r1 ? r : r0;
      *reinterpret_cast<float4*>(v[u]) = *reinterpret_cast<const float4*>(x + rc * C + cl * 4);
      if (residual) *reinterpret_cast<float4*>(rs[u]) = *reinterpret_cast<const float4*>(residual + rc * C + cl * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = rb + (int64_t)u * nrl;
      if (r >= r1) continue;
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = bn_pre(v[u][j], mu[j], is[j], g[j], bt[j]);
        if (residual) o[j] += rs[u][j];
        o[j] = act_fwd(o[j], act);
      }
      *reinterpret_cast<float4*>(y + r * C + cl * 4) = *reinterpret_cast<float4*>(o);
#pragma unroll
      for (int j = 0; j < 4; ++j) amax_fold(am, o[j]);
    }
  }
  if (amax_out) amax_commit(am, amax_out);
}

// many partial blocks: one 256-thread block per 4 channels (64 slices of the table each) adds them (fp64, fixed order) and writes
// mean / biased var / count + the nn.BatchNorm1d running-buffer update; follow with k_norm_act_fwd
__global__ __launch_bounds__(256) void k_bn2_finalize(const float* __restrict__ part, int nb, int C, int G, int64_t n, float momentum,
                                                       float* __restrict__ mean, float* __restrict__ var, float* __restrict__ cnt,
                                                       float* __restrict__ rmean, float* __restrict__ rvar,
                                                       long long* __restrict__ nbt) {
  __shared__ double r1[FIN_SL][FIN_CB + 1], r2[FIN_SL][FIN_CB + 1];
  const int cl = threadIdx.x & (FIN_CB - 1), j = threadIdx.x / FIN_CB;
  const int c = blockIdx.x * FIN_CB + cl;
  const int GC = G * C;
  double a1 = 0., a2 = 0.;
  if (c < C) {
    double p1[4] = {0., 0., 0., 0.}, p2[4] = {0., 0., 0., 0.};      // four partial blocks in flight
    const int total = nb * G;
    int t = j;
    for (; t + 3 * FIN_SL < total; t += 4 * FIN_SL) {
      float u[4], w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int tt = t + FIN_SL * q, b = tt / G, g = tt % G;
        u[q] = fc_ld(&part[((int64_t)b * 2) * GC + g * C + c]);
        w[q] = fc_ld(&part[((int64_t)b * 2 + 1) * GC + g * C + c]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { p1[q] += u[q]; p2[q] += w[q]; }
    }
    for (; t < total; t += FIN_SL) {
      const int b = t / G, g = t % G;
      p1[0] += fc_ld(&part[((int64_t)b * 2) * GC + g * C + c]);
      p2[0] += fc_ld(&part[((int64_t)b * 2 + 1) * GC + g * C + c]);
    }
    a1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
    a2 = (p2[0] + p2[1]) + (p2[2] + p2[3]);
  }
  r1[j][cl] = a1;
  r2[j][cl] = a2;
  __syncthreads();
  if (j == 0 && c < C) {
    double s1 = 0., s2 = 0.;
    for (int q = 0; q < FIN_SL; ++q) { s1 += r1[q][cl]; s2 += r2[q][cl]; }
    float mu, va;
    bn2_stats(s1, s2, n, &mu, &va);
    mean[c] = mu;
    var[c] = va;
    if (c == 0) { cnt[0] = (float)n; if (nbt) nbt[0] += 1; }
    if (rmean) {
      float unbias = (float)n / fmaxf((float)n - 1.f, 1.f);
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * va * unbias;
    }
  }
}

extern "C" {

static int stats_geometry(int C, int* threads, size_t* smem_fwd, size_t* smem_bwd) {
  if (C < 4 || C % 4 || C > 1024) return FC_EINVAL;
  int c4n = C / 4;
  int nrl = 256 / c4n;
  if (nrl < 1) nrl = 1;
  if (nrl > 16) nrl = 16;
  *threads = nrl * c4n;
  *smem_fwd = (size_t)(nrl * C + nrl) * sizeof(float);
  *smem_bwd = (size_t)(nrl * 2 * C) * sizeof(float);
  return FC_OK;
}

static void red_plan(int64_t n, int64_t* nb, int64_t* rpb) {
  int64_t m = n > 0 ? n : 1;
  int64_t g = fc_cdiv(m, 64);
  if (g > MAXBLOCKS) g = MAXBLOCKS;
  *rpb = fc_cdiv(m, g);
  *nb = fc_cdiv(m, *rpb);
}

int64_t fc_col_stats_ws_bytes(int64_t n, int C, int nseg) {
  int64_t nb, rpb;
  red_plan(n, &nb, &rpb);
  return nb * nseg * (2 * (int64_t)C + 1) * (int64_t)sizeof(float);
}

// mean (nseg,C), var (nseg,C) biased, cnt (nseg) ; seg = per-row segment id pointer (NULL -> one segment)
int fc_col_stats(const float* x, const int* seg, int seg_stride, int64_t n, int C, int nseg, float* mean, float* var,
                 float* cnt, void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (n < 0 || nseg < 1 || nseg > MAXSEG) return FC_EINVAL;
  int threads; size_t sf, sb;
  if (stats_geometry(C, &threads, &sf, &sb)) return FC_EINVAL;
  if (ws_bytes < fc_col_stats_ws_bytes(n, C, nseg)) return FC_EWS;
  int64_t nb, rpb;
  red_plan(n, &nb, &rpb);
  float* part = (float*)ws;
  float* part_cnt = part + nb * nseg * 2 * C;
  if (n == 0) {
    FC_HIP(hipMemsetAsync(mean, 0, sizeof(float) * nseg * C, stream));
    FC_HIP(hipMemsetAsync(var, 0, sizeof(float) * nseg * C, stream));
    FC_HIP(hipMemsetAsync(cnt, 0, sizeof(float) * nseg, stream));
    return FC_OK;
  }
  // r5: ONE pass over x (sums of x and of x^2 per block and segment), combined in fp64 — r1-r4 read x twice (mean, then the
  // centred squares), four launches
  k_stats_partial<<<(unsigned)nb, threads, sf, stream>>>(x, seg, seg_stride, n, C, nseg, nullptr, 2, rpb, part, part_cnt);
  FC_CHECK_LAUNCH();
  k_seg_meanvar_final<<<(unsigned)(nseg * ((C + 3) / 4)), 256, 0, stream>>>(part, part_cnt, nb, nseg, C, mean, var, cnt);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// out (nseg,C) = per-segment column sums of x (N,C) — deterministic two-level reduction
int fc_seg_col_sums(const float* x, const int* seg, int seg_stride, int64_t n, int C, int nseg, float* out, void* ws,
                    int64_t ws_bytes, hipStream_t stream) {
  if (n < 0 || nseg < 1 || nseg > MAXSEG) return FC_EINVAL;
  int threads; size_t sf, sb;
  if (stats_geometry(C, &threads, &sf, &sb)) return FC_EINVAL;
  if (ws_bytes < fc_col_stats_ws_bytes(n, C, nseg)) return FC_EWS;
  if (n == 0) {
    FC_HIP(hipMemsetAsync(out, 0, sizeof(float) * nseg * C, stream));
    return FC_OK;
  }
  int64_t nb, rpb;
  red_plan(n, &nb, &rpb);
  float* part = (float*)ws;
  k_stats_partial<<<(unsigned)nb, threads, sf, stream>>>(x, seg, seg_stride, n, C, nseg, nullptr, 0, rpb, part, nullptr);
  FC_CHECK_LAUNCH();
  k_stats_final<<<(unsigned)(nseg * ((C + FIN_CB - 1) / FIN_CB)), FIN_CB * FIN_SL, 0, stream>>>(part, nullptr, nb, nseg, C, 2, out, nullptr);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// BatchNorm1d running statistics (momentum update with the unbiased variance) + num_batches_tracked, one launch
__global__ void k_bn_running(const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ cnt,
                             float momentum, int C, float* __restrict__ rmean, float* __restrict__ rvar,
                             long long* __restrict__ nbt) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) nbt[0] += 1;
  if (c >= C) return;
  float n = cnt[0];
  float unbias = n / fmaxf(n - 1.f, 1.f);
  rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
  rvar[c] = (1.f - momentum) * rvar[c] + momentum * var[c] * unbias;
}

int fc_bn_running_update(const float* mean, const float* var, const float* cnt, float momentum, int C, float* running_mean,
                         float* running_var, long long* num_batches_tracked, hipStream_t stream) {
  if (C < 1) return FC_EINVAL;
  k_bn_running<<<(unsigned)fc_cdiv(C, 256), 256, 0, stream>>>(mean, var, cnt, momentum, C, running_mean, running_var,
                                                             num_batches_tracked);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ---- small-tensor BatchNorm, two launches per direction --------------------------------------------------
static void bn1_plan(int64_t n, int64_t* nb, int64_t* rpb) {
  int64_t m = n > 0 ? n : 1;
  int64_t g = fc_cdiv(m, 64);
  if (g > BN1_MAXB) g = BN1_MAXB;
  *rpb = fc_cdiv(m, g);
  *nb = fc_cdiv(m, *rpb);
}

int64_t fc_bn_small_ws_bytes(int C) { return (int64_t)BN1_MAXB * 2 * C * (int64_t)sizeof(float); }

int64_t fc_bn_stats_ws_bytes(int64_t n, int C) {
  int64_t nb, rpb;
  red_plan(n, &nb, &rpb);
  return nb * 2 * (int64_t)C * (int64_t)sizeof(float);
}

// training-mode BatchNorm statistics of a feature matrix of any size in one pass over x (+ buffer update):
// mean (C), biased var (C), cnt (1).  Follow with fc_norm_act_fwd.
int fc_bn_stats_train(const float* x, int64_t n, int C, float momentum, float* mean, float* var, float* cnt,
                      float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                      int64_t ws_bytes, hipStream_t stream) {
  if (n < 1) return FC_EINVAL;
  int threads; size_t sf, sb;
  if (stats_geometry(C, &threads, &sf, &sb)) return FC_EINVAL;
  if (ws_bytes < fc_bn_stats_ws_bytes(n, C)) return FC_EWS;
  int64_t nb, rpb;
  red_plan(n, &nb, &rpb);
  float* part = (float*)ws;
  k_bn1_partial<<<(unsigned)nb, threads, sb, stream>>>(x, n, C, rpb, part);
  FC_CHECK_LAUNCH();
  k_bn_finalize<<<(unsigned)((C + 63) / 64), 1024, 0, stream>>>(part, (int)nb, C, n, x, momentum, mean, var, cnt, running_mean,
                                                                running_var, num_batches_tracked);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// training-mode BatchNorm forward for one segment: statistics + running-buffer update + normalise/affine/residual/act.
// mean/var (C) and cnt (1) are written for the backward pass; running_* / num_batches_tracked may be NULL.
int fc_bn_act_train_fwd(const float* x, int64_t n, int C, float eps, const float* gamma, const float* beta,
                        const float* residual, int act, float momentum, float* y, float* mean, float* var, float* cnt,
                        float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                        int64_t ws_bytes, hipStream_t stream) {
  if (n < 1 || act < 0 || act > 2) return FC_EINVAL;
  int threads; size_t sf, sb;
  if (stats_geometry(C, &threads, &sf, &sb)) return FC_EINVAL;
  if (ws_bytes < fc_bn_small_ws_bytes(C)) return FC_EWS;
  int64_t nb, rpb;
  bn1_plan(n, &nb, &rpb);
  float* part = (float*)ws;
  k_bn1_partial<<<(unsigned)nb, threads, sb, stream>>>(x, n, C, rpb, part);
  FC_CHECK_LAUNCH();
  k_bn1_apply<<<(unsigned)nb, threads, sb, stream>>>(x, n, C, rpb, part, (int)nb, eps, gamma, beta, residual, act, momentum, y,
                                                    mean, var, cnt, running_mean, running_var, num_batches_tracked);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// its backward: sums (2,C) = [d beta, d gamma]
int fc_bn_act_train_bwd(const float* x, const float* y, const float* gy, int64_t n, int C, const float* mean,
                        const float* var, float eps, const float* gamma, const float* beta, int act, float* gx, float* gres,
                        float* sums, void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (n < 1 || act < 0 || act > 2) return FC_EINVAL;
  int threads; size_t sf, sb;
  if (stats_geometry(C, &threads, &sf, &sb)) return FC_EINVAL;
  if (ws_bytes < fc_bn_small_ws_bytes(C)) return FC_EWS;
  int64_t nb, rpb;
  bn1_plan(n, &nb, &rpb);
  float* part = (float*)ws;
  k_norm_bwd_partial<<<(unsigned)nb, threads, sb, stream>>>(x, y, gy, nullptr, nullptr, 0, n, C, 1, mean, var, eps, act, gamma, beta, rpb, part);
  FC_CHECK_LAUNCH();
  unsigned* ao = take_amax_out();
  k_bn1_bwd_apply<<<(unsigned)nb, threads, sb, stream>>>(x, y, gy, nullptr, n, C, rpb, part, (int)nb, mean, var, eps, gamma, beta, act, gx,
                                                        gres, sums, C, ao);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_norm_act_fwd(const float* x, const int* seg, int seg_stride, int64_t n, int C, const float* mean, const float* var,
                    float eps, const float* gamma, const float* beta, const float* residual, int act, float* y,
                    hipStream_t stream) {
  unsigned* ao = take_amax_out();                // fc_amax_out_hint: max |y| into the caller's (zeroed) word
  if (n < 0 || C < 4 || C % 4 || act < 0 || act > 2) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_norm_act_fwd<<<(unsigned)fc_cdiv(n * (C / 4), 256), 256, 0, stream>>>(x, seg, seg_stride, n, C, mean, var, eps, gamma,
                                                                         beta, residual, act, y, ao);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int64_t fc_norm_act_bwd_ws_bytes(int64_t n, int C, int nseg) {
  int64_t nb, rpb;
  red_plan(n, &nb, &rpb);
  return nb * nseg * 2 * (int64_t)C * (int64_t)sizeof(float);
}

// sums (nseg,2,C): [.,0,.] = sum g' (= d beta per segment), [.,1,.] = sum g'*xhat (= d gamma per segment)
int fc_norm_act_bwd(const float* x, const float* y, const float* gy, const int* seg, int seg_stride, int64_t n, int C,
                    int nseg, const float* mean, const float* var, const float* cnt, float eps, const float* gamma,
                    const float* beta, int act, float* gx, float* gres, float* sums, void* ws, int64_t ws_bytes,
                    hipStream_t stream) {
  unsigned* ao = take_amax_out();                // fc_amax_out_hint: max |gx| into the caller's (zeroed) word
  if (n < 0 || nseg < 1 || nseg > MAXSEG || act < 0 || act > 2) return FC_EINVAL;
  int threads; size_t sf, sb;
  if (stats_geometry(C, &threads, &sf, &sb)) return FC_EINVAL;
  if (ws_bytes < fc_norm_act_bwd_ws_bytes(n, C, nseg)) return FC_EWS;
  if (n == 0) {
    FC_HIP(hipMemsetAsync(sums, 0, sizeof(float) * nseg * 2 * C, stream));
    return FC_OK;
  }
  int64_t nb, rpb;
  red_plan(n, &nb, &rpb);
  float* part = (float*)ws;
  k_norm_bwd_partial<<<(unsigned)nb, threads, sb, stream>>>(x, y, gy, nullptr, seg, seg_stride, n, C, nseg, mean, var, eps, act, gamma,
                                                          beta, rpb, part);
  FC_CHECK_LAUNCH();
  k_stats_final<<<(unsigned)(nseg * ((2 * C + FIN_CB - 1) / FIN_CB)), FIN_CB * FIN_SL, 0, stream>>>(part, nullptr, nb, nseg, 2 * C, 2, sums, nullptr);
  FC_CHECK_LAUNCH();
  k_norm_bwd_apply<<<(unsigned)fc_cdiv(n * (C / 4), 256), 256, 0, stream>>>(x, y, gy, nullptr, seg, seg_stride, n, C, mean, var, eps,
                                                                           gamma, beta, sums, cnt, act, gx, gres, ao);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ---- r5: training-mode BatchNorm, one entry point per direction for every size ------------------------------------------------
// channel window of the prologue-reducing apply kernels: with few row blocks (the deep levels: 872 x 512, 3.5k x 256) a block owns 64
// channels and the grid's y dimension walks the windows — 8x the blocks, 16 row lanes each, the SAME total table traffic; else all of C
static int ap_window(int64_t n, int C) { return (C >= 128 && C % 64 == 0 && fc_cdiv(n > 0 ? n : 1, 64) * (C / 64) <= 1024 && fc_cdiv(n > 0 ? n : 1, 64) < 128) ? 64 : C; }
static void ap_plan(int64_t n, int64_t* nb, int64_t* rpb) {     // apply grid of the prologue-reducing kernels: up to 256 blocks
  int64_t m = n > 0 ? n : 1;
  int64_t g = fc_cdiv(m, 64);
  if (g > 256) g = 256;
  *rpb = fc_cdiv(m, g);
  *nb = fc_cdiv(m, *rpb);
}

int64_t fc_bn_train_ws_bytes(int64_t n, int C) {
  const int64_t a = fc_bn_stats_ws_bytes(n, C), b = fc_norm_act_bwd_ws_bytes(n, C, 1), c = fc_bn_small_ws_bytes(C);
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

// Forward.  part == NULL: the statistics are computed here from x (n * C <= small_elems: fc_bn_act_train_fwd, else fc_bn_stats_train +
// fc_norm_act_fwd — the r1-r4 routes).  part != NULL: nb_part row blocks of producer-written column sums [nb_part][2][groups * C]
// (see k_bn2_apply); x is then (groups * n_rows_of_the_producer, C) = n rows.
int fc_bn_train_fwd(const float* x, int64_t n, int C, float eps, const float* gamma, const float* beta, const float* residual,
                    int act, float momentum, float* y, float* mean, float* var, float* cnt, float* running_mean,
                    float* running_var, long long* num_batches_tracked, const float* part, int64_t nb_part, int groups,
                    int64_t small_elems, void* ws, int64_t ws_bytes, hipStream_t stream) {
  unsigned* ao = take_amax_out();                // fc_amax_out_hint: max |y| into the caller's (zeroed) word
  if (n < 1 || act < 0 || act > 2) return FC_EINVAL;
  if (!part) {
    if (n * C <= small_elems) {
      int rc = fc_bn_act_train_fwd(x, n, C, eps, gamma, beta, residual, act, momentum, y, mean, var, cnt, running_mean, running_var,
                                   num_batches_tracked, ws, ws_bytes, stream);
      if (rc == FC_OK && ao) rc = fc_amax(y, n * (int64_t)C, ao, stream);          // (this route's apply kernel does not fold: a pass of its own)
      return rc;
    }
    int rc = fc_bn_stats_train(x, n, C, momentum, mean, var, cnt, running_mean, running_var, num_batches_tracked, ws, ws_bytes, stream);
    if (rc) return rc;
    t_fc_amax_out = ao;
    return fc_norm_act_fwd(x, nullptr, 0, n, C, mean, var, eps, gamma, beta, residual, act, y, stream);
  }
  int threads; size_t sf, sb;
  if (stats_geometry(C, &threads, &sf, &sb)) return FC_EINVAL;
  if (nb_part < 1 || nb_part > 0x7fffffff / 64 || groups < 1 || groups > 64) return FC_EINVAL;
  if (nb_part <= BN1_MAXB) {
    int64_t nb, rpb;
    ap_plan(n, &nb, &rpb);
    const int CG = ap_window(n, C);
    if (CG != C && stats_geometry(CG, &threads, &sf, &sb)) return FC_EINVAL;
    const size_t smem = (size_t)(threads / (CG / 4)) * 2 * CG * sizeof(double);
    k_bn2_apply<<<dim3((unsigned)nb, C / CG), threads, smem, stream>>>(x, n, C, rpb, part, (int)nb_part, groups, eps, gamma, beta, residual, act,
                                                        momentum, y, mean, var, cnt, running_mean, running_var, num_batches_tracked, CG, ao);
    FC_CHECK_LAUNCH();
    return FC_OK;
  }
  k_bn2_finalize<<<(unsigned)((C + FIN_CB - 1) / FIN_CB), FIN_CB * FIN_SL, 0, stream>>>(part, (int)nb_part, C, groups, n, momentum, mean, var, cnt,
                                                                 running_mean, running_var, num_batches_tracked);
  FC_CHECK_LAUNCH();
  t_fc_amax_out = ao;
  return fc_norm_act_fwd(x, nullptr, 0, n, C, mean, var, eps, gamma, beta, residual, act, y, stream);
}

// Backward.  gy2 (nullable): a second contribution to the incoming gradient, added on the fly.  part == NULL: the sums of g' and
// g' xhat are reduced here (two or three launches by size, as fc_bn_act_train_bwd / fc_norm_act_bwd); part != NULL: nb_part blocks
// [nb_part][2][C] written by the producer of gy.
int fc_bn_train_bwd(const float* x, const float* y, const float* gy, const float* gy2, int64_t n, int C, const float* mean,
                    const float* var, const float* cnt, float eps, const float* gamma, const float* beta, int act, float* gx,
                    float* gres, float* sums, const float* part, int64_t nb_part, int64_t small_elems, void* ws, int64_t ws_bytes,
                    hipStream_t stream) {
  unsigned* ao = take_amax_out();                // fc_amax_out_hint: max |gx| into the caller's (zeroed) word
  if (n < 1 || act < 0 || act > 2) return FC_EINVAL;
  int threads; size_t sf, sb;
  if (stats_geometry(C, &threads, &sf, &sb)) return FC_EINVAL;
  const float* p = part;
  int64_t np = nb_part;
  if (!p) {
    const bool small = n * C <= small_elems;
    if (ws_bytes < (small ? fc_bn_small_ws_bytes(C) : fc_norm_act_bwd_ws_bytes(n, C, 1))) return FC_EWS;
    int64_t rpb;
    if (small) bn1_plan(n, &np, &rpb); else red_plan(n, &np, &rpb);
    k_norm_bwd_partial<<<(unsigned)np, threads, sb, stream>>>(x, y, gy, gy2, nullptr, 0, n, C, 1, mean, var, eps, act, gamma, beta, rpb,
                                                            (float*)ws);
    FC_CHECK_LAUNCH();
    p = (const float*)ws;
  }
  if (np < 1) return FC_EINVAL;
  if (np <= BN1_MAXB) {
    int64_t nb, rpb;
    if (!part) bn1_plan(n, &nb, &rpb); else ap_plan(n, &nb, &rpb);
    const int CG = ap_window(n, C);
    if (CG != C && stats_geometry(CG, &threads, &sf, &sb)) return FC_EINVAL;
    k_bn1_bwd_apply<<<dim3((unsigned)nb, C / CG), threads, sb, stream>>>(x, y, gy, gy2, n, C, rpb, p, (int)np, mean, var, eps, gamma, beta, act, gx,
                                                          gres, sums, CG, ao);
    FC_CHECK_LAUNCH();
    return FC_OK;
  }
  k_stats_final<<<(unsigned)((2 * C + FIN_CB - 1) / FIN_CB), FIN_CB * FIN_SL, 0, stream>>>(p, nullptr, np, 1, 2 * C, 2, sums, nullptr);
  FC_CHECK_LAUNCH();
  k_norm_bwd_apply<<<(unsigned)fc_cdiv(n * (C / 4), 256), 256, 0, stream>>>(x, y, gy, gy2, nullptr, 0, n, C, mean, var, eps, gamma, beta,
                                                                           sums, cnt, act, gx, gres, ao);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_amax_out_hint(unsigned* amax_word) {
  t_fc_amax_out = amax_word;
  return FC_OK;
}

int fc_maxpool_fwd(const float* in, const int* nbr, int64_t n_out, int K, int C, float* out, int* argrow,
                   hipStream_t stream) {
  unsigned* ao = take_amax_out();                // fc_amax_out_hint: max |out| into the caller's (zeroed) word
  if (n_out < 0 || K < 1 || C < 1) return FC_EINVAL;
  if (n_out == 0) return FC_OK;
  if (K == 8 && C % 4 == 0) k_maxpool8_fwd<<<(unsigned)fc_cdiv(n_out * (C / 4), 256), 256, 0, stream>>>(in, nbr, n_out, C, out, argrow, ao);
  else k_maxpool_fwd<<<(unsigned)fc_cdiv(n_out * C, 256), 256, 0, stream>>>(in, nbr, n_out, K, C, out, argrow);
  FC_CHECK_LAUNCH();
  if (ao && !(K == 8 && C % 4 == 0)) return fc_amax(out, n_out * (int64_t)C, ao, stream);
  return FC_OK;
}

// gin must be zero-filled by the caller (n_in rows)
int fc_maxpool_bwd(const float* gout, const int* argrow, int64_t n_out, int C, float* gin, hipStream_t stream) {
  if (n_out < 0 || C < 1) return FC_EINVAL;
  if (n_out == 0) return FC_OK;
  k_maxpool_bwd<<<(unsigned)fc_cdiv(n_out * C, 256), 256, 0, stream>>>(gout, argrow, n_out, C, gin);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_gather_rows(const float* src, const int* idx, int64_t n, int C, float* dst, hipStream_t stream) {
  if (n < 0 || C < 1) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_gather_rows<<<(unsigned)fc_cdiv(n * C, 256), 256, 0, stream>>>(src, idx, n, C, dst);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_scatter_rows_add(const float* src, const int* idx, int64_t n, int C, float* dst, hipStream_t stream) {
  if (n < 0 || C < 1) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_scatter_rows_add<<<(unsigned)fc_cdiv(n * C, 256), 256, 0, stream>>>(src, idx, n, C, dst);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"
