// Head epilogue of Fcaf3DNeckWithHead.forward_single (fcaf3d_neck_with_head.py:256-279), fused.
// The three 1x1 convolutions run as ONE GEMM into y (N, ld) = [centerness | reg (n_reg) | cls (n_cls) | pad]; this file
// turns y into the reference's three outputs (+ the per-row max class logit the pruning step interpolates) in one
// pass, and their gradients back into gy in one pass (instead of 3 zero-filled slice gradients + 2 adds per level).
// One wave64 per row: lane = column.
#include "fc_common.h"

#pragma clang fp contract(off)

// r6 (last take): 64 rows per workgroup — the wave that owns a row still reads its 256 bytes in one load, but the three outputs leave
// through an LDS tile as CONTIGUOUS runs (64 x n_cls, 64 x n_reg, 64 floats) instead of a 72-, a 24- and two 4-byte store per row
// (one wave per row: 104 us for the 441k rows of the finest level; the same values, bit for bit).
#define HEAD_FWD_RPB 64
__global__ __launch_bounds__(256) void k_head_split_fwd(const float* __restrict__ y, int ld, const float* __restrict__ bias,
                                                        const float* __restrict__ scale, int64_t n, int n_reg, int n_cls,
                                                        float* __restrict__ centerness, float* __restrict__ bbox,
                                                        float* __restrict__ cls, float* __restrict__ cls_max) {
  __shared__ float tile[HEAD_FWD_RPB][65];
  __shared__ float cmax[HEAD_FWD_RPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * HEAD_FWD_RPB;
  const float sc = scale[0];
  const float bs = (lane > n_reg && lane <= n_reg + n_cls) ? bias[lane - 1 - n_reg] : 0.f;
  for (int i0 = 0; i0 < HEAD_FWD_RPB / 4; i0 += 4) {                 // four rows of this wave in flight
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t row = base + wave + 4 * (i0 + u);
      v[u] = (row < n && lane < ld) ? y[row * ld + lane] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rl = wave + 4 * (i0 + u);
      float m = -INFINITY, o = v[u];
      if (lane >= 1 && lane <= n_reg) {
        o = lane - 1 < 6 ? expf(v[u] * sc) : v[u];
      } else if (lane > n_reg && lane <= n_reg + n_cls) {
        m = v[u] + bs;
        o = m;
      }
      tile[rl][lane] = o;
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if (lane == 0) cmax[rl] = m;
    }
  }
  __syncthreads();
  const int nr = (int)(n - base < HEAD_FWD_RPB ? n - base : HEAD_FWD_RPB);
  for (int i = threadIdx.x; i < nr; i += 256) { centerness[base + i] = tile[i][0]; cls_max[base + i] = cmax[i]; }
  for (int i = threadIdx.x; i < nr * n_reg; i += 256) { const int r = i / n_reg; bbox[base * n_reg + i] = tile[r][1 + i - r * n_reg]; }
  for (int i = threadIdx.x; i < nr * n_cls; i += 256) { const int r = i / n_cls; cls[base * n_cls + i] = tile[r][1 + n_reg + i - r * n_cls]; }
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

// r5: the same pass over 64 rows per workgroup (16 per wave), which ALSO leaves the column sums of gy and the sum of the per-row
// d/dscale terms of its rows: part[block][65] = {sum over the block's rows of gy[.][lane] (lane 0..63), sum of gscale_row}.
// The class-bias gradient (column sums of d loss / d cls_score) and the scale gradient were two single-block reductions over
// all locations on the critical path (k_col_sum: 5 launches, 0.35 ms per step at 8 scenes, profiles/r4_kernel_stats.md).
#define HEAD_RPB 64
__global__ __launch_bounds__(256) void k_head_split_bwd_sums(const float* __restrict__ y, int ld, const float* __restrict__ scale,
                                                             const float* __restrict__ bbox, const float* __restrict__ g_cent,
                                                             const float* __restrict__ g_bbox, const float* __restrict__ g_cls,
                                                             int64_t n, int n_reg, int n_cls, float* __restrict__ gy,
                                                             float* __restrict__ part, unsigned* __restrict__ amax_out) {
  __shared__ float red[4][65];
  // r6 (last take): the incoming gradients (and the box exponentials) of the workgroup's 64 rows come in as contiguous runs through an
  // LDS tile — one wave per row read a 4-, two 24- and a 72-byte piece per row —; the arithmetic and every sum keep their order
  __shared__ float gt[HEAD_RPB][65];
  __shared__ float bt6[HEAD_RPB][7];
  unsigned am = 0u;                              // r6: max |gy| for the backward-data GEMM that gathers it (fc_amax_out_hint)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * HEAD_RPB;
  const float sc = scale[0];
  {
    const int nr = (int)(n - base < HEAD_RPB ? n - base : HEAD_RPB);
    for (int i = threadIdx.x; i < HEAD_RPB * 64; i += 256) gt[i >> 6][i & 63] = 0.f;
    __syncthreads();
    if (g_cent) for (int i = threadIdx.x; i < nr; i += 256) gt[i][0] = g_cent[base + i];
    for (int i = threadIdx.x; i < nr * n_reg; i += 256) {
      const int r = i / n_reg, j = i - r * n_reg;
      if (g_bbox) gt[r][1 + j] = g_bbox[base * n_reg + i];
      if (j < 6) bt6[r][j] = bbox[base * n_reg + i];
    }
    if (g_cls) for (int i = threadIdx.x; i < nr * n_cls; i += 256) { const int r = i / n_cls; gt[r][1 + n_reg + i - r * n_cls] = g_cls[base * n_cls + i]; }
    __syncthreads();
  }
  float ag = 0.f, as = 0.f;
  // four rows of this wave in flight per pass (their loads first: one row per pass left the 16 passes of a wave as a chain of
  // dependent latencies — 72 us per launch against 37 us of the one-row-per-wave kernel it replaces)
  const int kind = lane == 0 ? 0 : (lane <= n_reg ? (lane - 1 < 6 ? 1 : 2) : (lane <= n_reg + n_cls ? 3 : 4));
  for (int i0 = 0; i0 < HEAD_RPB / 4; i0 += 4) {
    float gin[4], bb[4], yy[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t row = base + wave + 4 * (i0 + u);
      ok[u] = row < n;
      const int64_t rc = ok[u] ? row : 0;
      const int rl = wave + 4 * (i0 + u);
      gin[u] = ok[u] ? gt[rl][lane] : 0.f; bb[u] = 0.f; yy[u] = 0.f;
      if (kind == 1) { bb[u] = ok[u] ? bt6[rl][lane - 1] : 0.f; yy[u] = y[rc * ld + lane]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const int64_t row = base + wave + 4 * (i0 + u);
      float g = gin[u], s = 0.f;
      if (kind == 1) {
        const float e = gin[u] * bb[u];              // d exp(reg*scale) = exp(.) * (scale dreg + reg dscale)
        g = e * sc;
        s = e * yy[u];
      }
      if (lane < ld) { gy[row * ld + lane] = g; amax_fold(am, g); }
      ag += g;
      as += s;
    }
  }
  for (int off = 32; off > 0; off >>= 1) as += __shfl_xor(as, off, 64);
  red[wave][lane] = ag;
  if (lane == 0) red[wave][64] = as;
  __syncthreads();
  if (threadIdx.x < 65)
    part[(int64_t)blockIdx.x * 65 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (amax_out) amax_commit(am, amax_out);
}

// block b < n_cls: gbias[b] = sum over the partial blocks of column 1 + n_reg + b; block n_cls: gscale[0] = sum of entry 64
// (fixed tree order: deterministic)
// (256 threads: a 1 024-thread workgroup waits for a whole compute unit to drain beside the weight-gradient stream, see norm.hip)
__global__ __launch_bounds__(256) void k_head_sums_final(const float* __restrict__ part, int64_t nb, int n_reg, int n_cls,
                                                         float* __restrict__ gbias, float* __restrict__ gscale) {
  __shared__ float red[256];
  const int col = (int)blockIdx.x < n_cls ? 1 + n_reg + (int)blockIdx.x : 64;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int64_t b = threadIdx.x;
  for (; b + 3 * 256 < nb; b += 4 * 256) {
    a0 += part[b * 65 + col]; a1 += part[(b + 256) * 65 + col]; a2 += part[(b + 512) * 65 + col]; a3 += part[(b + 768) * 65 + col];
  }
  for (; b < nb; b += 256) a0 += part[b * 65 + col];
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if ((int)blockIdx.x < n_cls) { if (gbias) gbias[blockIdx.x] = red[0]; }
    else if (gscale) gscale[0] = red[0];
  }
}

extern "C" {

int64_t fc_head_split_bwd_sums_ws_bytes(int64_t n) { return fc_cdiv(n > 0 ? n : 1, HEAD_RPB) * 65 * (int64_t)sizeof(float); }

int fc_head_split_bwd_sums(const float* y, int ld, const float* scale_dev, const float* bbox_pred, const float* g_centerness,
                           const float* g_bbox, const float* g_cls, int64_t n, int n_reg, int n_cls, float* gy, float* gbias,
                           float* gscale, void* ws, int64_t ws_bytes, hipStream_t stream) {
  unsigned* ao = take_amax_out();                // fc_amax_out_hint: max |gy| into the caller's (zeroed) slot
  if (n < 0 || ld < 1 || ld > 64 || n_reg < 6 || n_cls < 1 || 1 + n_reg + n_cls > ld) return FC_EINVAL;
  if (ws_bytes < fc_head_split_bwd_sums_ws_bytes(n)) return FC_EWS;
  if (n == 0) {
    if (gbias) FC_HIP(hipMemsetAsync(gbias, 0, sizeof(float) * n_cls, stream));
    if (gscale) FC_HIP(hipMemsetAsync(gscale, 0, sizeof(float), stream));
    return FC_OK;
  }
  const int64_t nb = fc_cdiv(n, HEAD_RPB);
  float* part = (float*)ws;
  k_head_split_bwd_sums<<<(unsigned)nb, 256, 0, stream>>>(y, ld, scale_dev, bbox_pred, g_centerness, g_bbox, g_cls, n, n_reg, n_cls,
                                                         gy, part, ao);
  FC_CHECK_LAUNCH();
  k_head_sums_final<<<(unsigned)(n_cls + 1), 256, 0, stream>>>(part, nb, n_reg, n_cls, gbias, gscale);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_head_split_fwd(const float* y, int ld, const float* bias, const float* scale_dev, int64_t n, int n_reg, int n_cls,
                      float* centerness, float* bbox_pred, float* cls_score, float* cls_max, hipStream_t stream) {
  if (n < 0 || ld < 1 || ld > 64 || n_reg < 6 || n_cls < 1 || 1 + n_reg + n_cls > ld) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_head_split_fwd<<<(unsigned)fc_cdiv(n, HEAD_FWD_RPB), 256, 0, stream>>>(y, ld, bias, scale_dev, n, n_reg, n_cls, centerness,
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
