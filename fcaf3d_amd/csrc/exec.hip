// Native launch-list executor (r4): the forward / backward pass of the FCAF3D network body as ONE C-ABI call per direction.
//
// r4 host profile (tools/hostprof.py, profiles/r4_hostprof.txt): the training step was HOST-bound at every batch size —
// 16.7 ms of Python per step at 2 scenes (drained 16.75 ms), 20.3 ms at 8 (drained 22.8): ~1 000 launches per step, each
// reached through nn.Module.__call__ -> autograd.Function.apply -> torch.empty -> ctypes (~16 us per launch), where the HIP
// launch itself costs ~3.5 us.  The network body is a STATIC sequence of operators (me_resnet.py:43-50, BasicBlock,
// fcaf3d_neck_with_head.py:94-108; only row counts and addresses change from step to step), so the host side builds the
// operator list once per model (fcaf3d_amd/executor.py) and this interpreter walks it: every operator is one of the
// library's own entry points (the kernels, tiles, routes and results are those of the per-operator path — tests compare the
// two bit for bit in the forward pass), every operand an index into a table of device addresses / row counts that the host
// refreshes per step.  Streams: 0 = the caller's stream (the step's dependent chain), 1 = the head branch of the neck
// (out_block_i + forward_single, i > 0), 2 = weight gradients; cross-stream order through hipEvents owned by the library
// (created once per process, never destroyed: a handful).
//
// Reference: what this replaces is mmcv's per-module Python dispatch of the same graph (mmdet3d/models/detectors/
// single_stage_sparse.py:43-50 `extract_feat`, torch.autograd's backward over it).
#include "fc_common.h"
#include "../../include/fcaf3d_hip.h"
#include <vector>

namespace {

enum : int64_t {
  OP_STEM_FWD = 1, OP_COL_STATS, OP_NORM_FWD, OP_MAXPOOL_FWD, OP_CONV, OP_BN_FWD, OP_UNION_FWD, OP_HEAD_FWD, OP_RECORD, OP_WAIT,
  OP_HEAD_BWD, OP_WGRAD, OP_BN_BWD, OP_NORM_BWD, OP_MAXPOOL_BWD, OP_STEM_WGRAD, OP_GATHER, OP_ADD, OP_SMALL_GRADS,
  OP_PERMUTE_GENT, OP_HEAD_WFIN, OP_COPY, OP_COL_SUM, OP_ROW_SUM, OP_AMAX, OP_CLEAR
};
constexpr int OPW = 24;            // int64 words per operator
constexpr int MAPW = 20;           // int64 words per kernel-map descriptor
constexpr int NSTREAM = 3;
constexpr int MAX_EVENTS = 64;
constexpr int CONV_X6 = (1 << 24) | (1 << 26), WGRAD_X6 = 1 << 24;

// one event set per DEVICE (an event belongs to the device that was current when it was created; a process that drives
// detectors on two devices must not share them — ADVICE r4).  fc_exec is NOT re-entrant per device: one thread per device at a
// time (the reference's loop is one Python thread per process, SURVEY.md 8(b) "Threading").
constexpr int MAX_DEVICES = 16;
hipEvent_t g_events_dev[MAX_DEVICES][MAX_EVENTS];
bool g_events_ready[MAX_DEVICES] = {};
thread_local hipEvent_t* g_events = nullptr;      // the current call's set

int ensure_events() {
  int dev = 0;
  FC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return FC_EINVAL;
  if (!g_events_ready[dev]) {
    for (int i = 0; i < MAX_EVENTS; ++i) FC_HIP(hipEventCreateWithFlags(&g_events_dev[dev][i], hipEventDisableTiming));
    g_events_ready[dev] = true;
  }
  g_events = g_events_dev[dev];
  return 0;
}

// ---- probe: HIP-event brackets around the convolution launches of a call (bench.py's roofline measurement) ----------------
struct ProbeRec { hipEvent_t a, b; int64_t meta[8]; };
std::vector<ProbeRec> g_probe;          // records since the last fc_exec_probe_read
std::vector<hipEvent_t> g_probe_pool;   // timing events, reused

hipEvent_t probe_event() {
  if (!g_probe_pool.empty()) { hipEvent_t e = g_probe_pool.back(); g_probe_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

inline double as_double(int64_t v) { double d; __builtin_memcpy(&d, &v, 8); return d; }

__global__ void k_add_inplace(float* __restrict__ dst, const float* __restrict__ src, int64_t n4) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n4) return;
  float4 a = reinterpret_cast<float4*>(dst)[t];
  const float4 b = reinterpret_cast<const float4*>(src)[t];
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  reinterpret_cast<float4*>(dst)[t] = a;
}

// desc[e] = {src, dst, C, nseg, seg_stride (floats), -, -, -}: dst[c] = sum_s src[s * seg_stride + c]  (the d gamma / d beta sums of every
// normalisation layer, written by the backward kernels into one persistent buffer, land in their gradient slices in ONE launch)
__global__ void k_small_grads(const long long* __restrict__ desc, int n) {
  const int e = blockIdx.x;
  if (e >= n) return;
  const long long* d = desc + 8 * e;
  const float* src = reinterpret_cast<const float*>(d[0]);
  float* dst = reinterpret_cast<float*>(d[1]);
  const int C = (int)d[2], nseg = (int)d[3];
  const long long stride = d[4];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nseg; ++k) s += fc_ld(src + k * stride + c);
    dst[c] = s;
  }
}

// gW' (Cin, 8 Cout) of the generative transposed convolution's dense GEMM -> the layer's kernel gradient (8, Cin, Cout)
__global__ void k_permute_gent(const float* __restrict__ src, float* __restrict__ dst, int Cin, int Cout) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // index into dst
  const int64_t total = (int64_t)8 * Cin * Cout;
  if (t >= total) return;
  const int d = (int)(t % Cout), c = (int)((t / Cout) % Cin), k = (int)(t / ((int64_t)Cout * Cin));
  dst[t] = src[(int64_t)c * 8 * Cout + (int64_t)k * Cout + d];
}

// the packed head kernel's gradient: sum of the per-level partials (nl, R, ld) -> centerness (R, 1), reg (R, n_reg), cls (R, n_cls)
// ... and (r5) the class-bias gradient: sum over the levels of bias_part (nl, n_cls), the column sums fc_head_split_bwd_sums left
__global__ void k_head_wfin(const float* __restrict__ part, int nl, int R, int ld, int n_reg, int n_cls, float* __restrict__ g_cent,
                            float* __restrict__ g_reg, float* __restrict__ g_cls, const float* __restrict__ bias_part,
                            float* __restrict__ g_bias) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ncol = 1 + n_reg + n_cls;
  if (g_bias && t < n_cls) {
    float b = 0.f;
    for (int l = 0; l < nl; ++l) b += fc_ld(&bias_part[l * n_cls + t]);
    g_bias[t] = b;
  }
  if (t >= R * ncol) return;
  const int r = t / ncol, c = t % ncol;
  float s = 0.f;
  for (int l = 0; l < nl; ++l) s += part[((int64_t)l * R + r) * ld + c];
  if (c == 0) g_cent[r] = s;
  else if (c <= n_reg) g_reg[r * n_reg + c - 1] = s;
  else g_cls[r * n_cls + c - 1 - n_reg] = s;
}

// dst[c] = sum over n rows of x[r][c] (C columns, any C): one 1024-thread block per column, rows strided over the threads, fixed
// tree order — the head's class-bias gradient (the column sums of d loss / d cls_score over every location of the batch)
__global__ __launch_bounds__(1024) void k_col_sum(const float* __restrict__ x, int64_t n, int C, float* __restrict__ dst) {
  __shared__ float red[1024];
  const int c = blockIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int64_t r = threadIdx.x;
  for (; r + 3 * 1024 < n; r += 4 * 1024) {
    a0 += x[r * C + c]; a1 += x[(r + 1024) * C + c]; a2 += x[(r + 2048) * C + c]; a3 += x[(r + 3072) * C + c];
  }
  for (; r < n; r += 1024) a0 += x[r * C + c];
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) dst[c] = red[0];
}

struct Ctx {
  const int64_t* ops;            // the WHOLE operator list (an operator may refer to its producer by index)
  const int64_t* addr;
  const int64_t* dims;
  const int64_t* maps;
  hipStream_t streams[NSTREAM];
  void* ws[NSTREAM];
  int64_t ws_bytes[NSTREAM];
  int64_t need[NSTREAM];
  bool dry;
  bool probe;
  int64_t bn_small_elems;
  int flags;
};

int run_op_impl(Ctx& c, const int64_t* op);

int run_op(Ctx& c, const int64_t* op) {
  if (c.dry || !c.probe || (op[0] != OP_CONV && op[0] != OP_AMAX)) return run_op_impl(c, op);
  ProbeRec r;
  r.a = probe_event(); r.b = probe_event();
  if (!r.a || !r.b) return run_op_impl(c, op);
  hipStream_t st = c.streams[op[1]];
  const int64_t Cin = op[8], Cout = op[9];
  if (op[0] == OP_AMAX) {          // the amax pass of a convolution operand (h3): meta[0] = -2, n_in = rows, Cin = columns, no FLOPs
    const int64_t m[8] = {-2, 0, c.dims[op[3]], 0, 0, op[4], 0, 0};
    __builtin_memcpy(r.meta, m, sizeof m);
  } else if (op[4] < 0) {
    const int64_t n = c.dims[op[7]];
    const int64_t m[8] = {-1, op[5], n, n, 1, Cin, Cout, 0};
    __builtin_memcpy(r.meta, m, sizeof m);
  } else {
    const int64_t* mp = c.maps + op[4] * MAPW;
    const bool bwd = op[5] != 0;
    const int64_t m[8] = {op[4], op[5], bwd ? mp[1] : mp[0], bwd ? mp[0] : mp[1], mp[2], Cin, Cout, (mp[19] & (bwd ? 2 : 1)) ? 1 : 0};
    __builtin_memcpy(r.meta, m, sizeof m);
  }
  FC_HIP(hipEventRecord(r.a, st));
  const int rc = run_op_impl(c, op);
  FC_HIP(hipEventRecord(r.b, st));
  g_probe.push_back(r);
  return rc;
}

template <class T>
inline T* P(const Ctx& c, int64_t idx) { return idx < 0 ? nullptr : reinterpret_cast<T*>(c.addr[idx]); }

// Row blocks of the statistics table the convolution operator `pop` leaves (its word 10 = address index + 1 of the table): the
// route logic of OP_CONV below, as a pure function of the step's tables — the BatchNorm operator behind it may sit in another
// segment of a segmented run, so nothing is remembered from the launch.
int64_t stats_blocks_of(const Ctx& c, const int64_t* pop) {
  const int Cin = (int)pop[8], Cout = (int)pop[9];
  const int fl = c.flags | CONV_X6;
  if (pop[4] < 0) return fc_conv_stats_blocks(c.dims[pop[7]], 1, Cin, Cout, fl, 0);
  const int64_t* m = c.maps + pop[4] * MAPW;
  const bool bwd = pop[5] != 0;
  return fc_conv_stats_blocks(bwd ? m[0] : m[1], (int)m[2], Cin, Cout, fl, (m[19] & (bwd ? 2 : 1)) ? 1 : 0);
}

inline bool want_ws(Ctx& c, int s, int64_t bytes) {     // true: the launch may go ahead
  if (bytes > c.need[s]) c.need[s] = bytes;
  return !c.dry;
}

int run_op_impl(Ctx& c, const int64_t* op) {
  const int s = (int)op[1];
  hipStream_t st = c.streams[s];
#ifdef FC_KO_EXEC
  // knock-out build (tools/knockout.sh FC_KO_EXEC; never the product library): FC_KO_OPS = bit mask of op families NOT launched
  // (1 weight gradients, 2 convolutions, 4 normalisation) — what the step costs without them (results are garbage)
  {
    static const int ko = getenv("FC_KO_OPS") ? atoi(getenv("FC_KO_OPS")) : 0;
    const int64_t o = op[0];
    if (!c.dry && (((ko & 1) && (o == OP_WGRAD || o == OP_STEM_WGRAD)) || ((ko & 2) && o == OP_CONV) ||
                   ((ko & 4) && (o == OP_BN_FWD || o == OP_BN_BWD || o == OP_NORM_FWD || o == OP_NORM_BWD || o == OP_COL_STATS))))
      return 0;
  }
#endif
  switch (op[0]) {
    case OP_STEM_FWD: {   // in, W, map, out, col
      const int64_t* m = c.maps + op[4] * MAPW;
      if (c.dry) return 0;
      return fc_stem_conv_fwd(P<const float>(c, op[2]), P<const float>(c, op[3]), reinterpret_cast<const int*>(m[3]), P<float>(c, op[5]),
                              P<float>(c, op[6]), m[0], m[1], (int)m[2], st);
    }
    case OP_COL_STATS: {  // x, seg, n(dim), C, nseg(dim), mean, var, cnt
      const int64_t n = c.dims[op[4]];
      const int C = (int)op[5], nseg = (int)c.dims[op[6]];
      if (!want_ws(c, s, fc_col_stats_ws_bytes(n, C, nseg))) return 0;
      return fc_col_stats(P<const float>(c, op[2]), P<const int>(c, op[3]), op[3] >= 0 ? 4 : 0, n, C, nseg, P<float>(c, op[7]),
                          P<float>(c, op[8]), P<float>(c, op[9]), c.ws[s], c.ws_bytes[s], st);
    }
    case OP_NORM_FWD: {   // x, seg, n(dim), C, mean, var, eps, gamma, beta, res, act, y
      if (c.dry) return 0;
      return fc_norm_act_fwd(P<const float>(c, op[2]), P<const int>(c, op[3]), op[3] >= 0 ? 4 : 0, c.dims[op[4]], (int)op[5],
                             P<const float>(c, op[6]), P<const float>(c, op[7]), (float)as_double(op[8]), P<const float>(c, op[9]),
                             P<const float>(c, op[10]), P<const float>(c, op[11]), (int)op[12], P<float>(c, op[13]), st);
    }
    case OP_MAXPOOL_FWD: {  // in, map, C, out, arg, amax word of out + 1 | 0
      const int64_t* m = c.maps + op[3] * MAPW;
      if (c.dry) return 0;
      if (op[7] > 0) fc_amax_out_hint(P<unsigned>(c, op[7] - 1));
      return fc_maxpool_fwd(P<const float>(c, op[2]), reinterpret_cast<const int*>(m[3]), m[1], (int)m[2], (int)op[4], P<float>(c, op[5]),
                            P<int>(c, op[6]), st);
    }
    case OP_CONV: {  // in, img, map (-1: dense GEMM over n rows), dir, out, n(dim, dense only), Cin, Cout, statistics table + 1 | 0,
                     // [BatchNorm-backward form of the table:] bn_x + 1 | 0, mean, var, gamma, beta, eps, act, add + 1 | 0, bn_y + 1 | 0
      const int Cin = (int)op[8], Cout = (int)op[9];
      const int fl = c.flags | CONV_X6;
      float* stats = (op[10] > 0 && stats_blocks_of(c, op) > 0) ? P<float>(c, op[10] - 1) : nullptr;
      const float* bnx = (stats && op[11] > 0) ? P<const float>(c, op[11] - 1) : nullptr;
      const float *bmean = P<const float>(c, op[12]), *bvar = P<const float>(c, op[13]), *bga = P<const float>(c, op[14]),
                  *bbe = P<const float>(c, op[15]);
      const float beps = (float)as_double(op[16]);
      const int bact = (int)op[17];
      const float* badd = op[18] > 0 ? P<const float>(c, op[18] - 1) : nullptr;     // second gradient contribution / the layer's output
      const float* bny = op[19] > 0 ? P<const float>(c, op[19] - 1) : nullptr;
      const float* in = P<const float>(c, op[2]);
      const float* img = P<const float>(c, op[3]);
      float* out = P<float>(c, op[6]);
      // word 20: the amax word of `in` + 1 (h3 split; 0: the entry point makes its own pass) — consumed by the ONE call below
      if (!c.dry && op[20] > 0) fc_conv_amax_hint(P<const unsigned>(c, op[20] - 1), nullptr);
      if (op[4] < 0) {
        const int64_t n = c.dims[op[7]];
        if (!want_ws(c, s, fc_conv_fwd_ws_bytes(n, 1, Cin, Cout, fl))) return 0;
        if (bnx)
          return fc_conv_fwd_bn_bwd_stats(in, img, nullptr, nullptr, out, n, n, 1, Cin, Cout, fl, c.ws[s], c.ws_bytes[s], stats, bnx, bmean, bvar,
                                          bga, bbe, beps, bact, badd, bny, st);
        return fc_conv_fwd_stats(in, img, nullptr, nullptr, out, n, n, 1, Cin, Cout, fl, c.ws[s], c.ws_bytes[s], stats, st);
      }
      const int64_t* m = c.maps + op[4] * MAPW;
      const bool bwd = op[5] != 0;
      const int64_t n_in = bwd ? m[1] : m[0], n_out = bwd ? m[0] : m[1];
      const int K = (int)m[2];
      if (m[19] & (bwd ? 2 : 1)) {                       // per offset over the exact pair lists
        const int b = bwd ? 14 : 9;
        const int *pi = reinterpret_cast<const int*>(m[b]), *pc = reinterpret_cast<const int*>(m[b + 3]),
                  *pp = reinterpret_cast<const int*>(m[b + 2]);
        if (!want_ws(c, s, fc_conv_fwd_pairs_ws_bytes(n_out, K, Cout))) return 0;
        if (bnx)
          return fc_conv_fwd_pairs_tiles_bn_bwd_stats(in, img, pi, pc, pp, out, n_in, n_out, K, Cin, Cout, m[b + 4], fl, c.ws[s], c.ws_bytes[s],
                                                      stats, bnx, bmean, bvar, bga, bbe, beps, bact, badd, bny, st);
        return fc_conv_fwd_pairs_tiles_stats(in, img, pi, pc, pp, out, n_in, n_out, K, Cin, Cout, m[b + 4], fl, c.ws[s], c.ws_bytes[s], stats,
                                             st);
      }
      const int *tab = reinterpret_cast<const int*>(m[bwd ? 7 : 5]), *oidx = reinterpret_cast<const int*>(m[bwd ? 8 : 6]);
      if (!want_ws(c, s, fc_conv_fwd_ws_bytes(n_out, K, Cin, Cout, fl))) return 0;
      if (bnx)
        return fc_conv_fwd_bn_bwd_stats(in, img, tab, oidx, out, n_in, n_out, K, Cin, Cout, fl, c.ws[s], c.ws_bytes[s], stats, bnx, bmean, bvar,
                                        bga, bbe, beps, bact, badd, bny, st);
      return fc_conv_fwd_stats(in, img, tab, oidx, out, n_in, n_out, K, Cin, Cout, fl, c.ws[s], c.ws_bytes[s], stats, st);
    }
    case OP_BN_FWD: {  // x, n(dim), C, eps, gamma, beta, res, act, momentum, y, mean, var, cnt, rmean, rvar, nbt, train
      const int64_t n = c.dims[op[3]];
      const int C = (int)op[4];
      const float eps = (float)as_double(op[5]), mom = (float)as_double(op[10]);
      if (!op[18]) {      // eval mode: the running statistics are the statistics
        if (c.dry) return 0;
        if (op[21] > 0) fc_amax_out_hint(P<unsigned>(c, op[21] - 1));
        return fc_norm_act_fwd(P<const float>(c, op[2]), nullptr, 0, n, C, P<const float>(c, op[15]), P<const float>(c, op[16]), eps,
                               P<const float>(c, op[6]), P<const float>(c, op[7]), P<const float>(c, op[8]), (int)op[9], P<float>(c, op[11]),
                               st);
      }
      // training: statistics from the producer's epilogue when the operator names one (word 19 = producer index + 1, word 20 =
      // column groups per channel) and that launch has a statistics epilogue; else computed from x (fc_bn_train_fwd)
      const float* part = nullptr;
      int64_t nbp = 0;
      if (op[19] > 0) {
        const int64_t* pop = c.ops + (op[19] - 1) * OPW;
        nbp = pop[10] > 0 ? stats_blocks_of(c, pop) : 0;
        if (nbp > 0) part = P<const float>(c, pop[10] - 1);
      }
      if (!want_ws(c, s, fc_bn_train_ws_bytes(n, C))) return 0;
      if (op[21] > 0) fc_amax_out_hint(P<unsigned>(c, op[21] - 1));         // word 21: amax word of y + 1 | 0 (h3: y feeds a convolution)
      return fc_bn_train_fwd(P<const float>(c, op[2]), n, C, eps, P<const float>(c, op[6]), P<const float>(c, op[7]),
                             P<const float>(c, op[8]), (int)op[9], mom, P<float>(c, op[11]), P<float>(c, op[12]), P<float>(c, op[13]),
                             P<float>(c, op[14]), P<float>(c, op[15]), P<float>(c, op[16]), P<long long>(c, op[17]), part, nbp,
                             op[20] > 0 ? (int)op[20] : 1, c.bn_small_elems, c.ws[s], c.ws_bytes[s], st);
    }
    case OP_UNION_FWD: {  // fa, fb, rows, n_a(dim), n_b(dim), n_union(dim), C, out:  out[:n_a] = fa, rest 0, out[rows[i]] += fb[i]
      if (c.dry) return 0;
      const int64_t n_a = c.dims[op[5]], n_b = c.dims[op[6]], n_u = c.dims[op[7]];
      const int C = (int)op[8];
      float* out = P<float>(c, op[9]);
      FC_HIP(hipMemcpyAsync(out, P<const float>(c, op[2]), sizeof(float) * n_a * C, hipMemcpyDeviceToDevice, st));
      if (n_u > n_a) FC_HIP(hipMemsetAsync(out + n_a * C, 0, sizeof(float) * (n_u - n_a) * C, st));
      return fc_scatter_rows_add(P<const float>(c, op[3]), P<const int>(c, op[4]), n_b, C, out, st);
    }
    case OP_HEAD_FWD: {  // y, ld, bias, scale, n(dim), n_reg, n_cls, cent, bbox, cls, cmax
      if (c.dry) return 0;
      return fc_head_split_fwd(P<const float>(c, op[2]), (int)op[3], P<const float>(c, op[4]), P<const float>(c, op[5]), c.dims[op[6]],
                               (int)op[7], (int)op[8], P<float>(c, op[9]), P<float>(c, op[10]), P<float>(c, op[11]), P<float>(c, op[12]), st);
    }
    case OP_RECORD:
      if (c.dry) return 0;
      FC_HIP(hipEventRecord(g_events[op[2]], st));
      return 0;
    case OP_WAIT:
      if (c.dry) return 0;
      FC_HIP(hipStreamWaitEvent(st, g_events[op[2]], 0));
      return 0;
    case OP_HEAD_BWD: {  // y, ld, scale, bbox, g_cent, g_bbox, g_cls, n(dim), n_reg, n_cls, gy, gs_row | -1, g_scale, bias_part
      if (op[13] < 0) {    // with the scale / class-bias reductions of this level (fc_head_split_bwd_sums)
        const int64_t n = c.dims[op[9]];
        if (!want_ws(c, s, fc_head_split_bwd_sums_ws_bytes(n))) return 0;
        if (op[16] > 0) fc_amax_out_hint(P<unsigned>(c, op[16] - 1));         // word 16: amax slot of gy + 1 | 0
        return fc_head_split_bwd_sums(P<const float>(c, op[2]), (int)op[3], P<const float>(c, op[4]), P<const float>(c, op[5]),
                                      P<const float>(c, op[6]), P<const float>(c, op[7]), P<const float>(c, op[8]), n, (int)op[10],
                                      (int)op[11], P<float>(c, op[12]), P<float>(c, op[15]), P<float>(c, op[14]), c.ws[s],
                                      c.ws_bytes[s], st);
      }
      if (c.dry) return 0;
      return fc_head_split_bwd(P<const float>(c, op[2]), (int)op[3], P<const float>(c, op[4]), P<const float>(c, op[5]),
                               P<const float>(c, op[6]), P<const float>(c, op[7]), P<const float>(c, op[8]), c.dims[op[9]], (int)op[10],
                               (int)op[11], P<float>(c, op[12]), P<float>(c, op[13]), st);
    }
    case OP_WGRAD: {  // in, gout, map (-1: dense), gW, n(dim, dense only), Cin, Cout, amax word of in + 1 | 0, of gout + 1 | 0
      const int Cin = (int)op[7], Cout = (int)op[8];
      const int fl = c.flags | WGRAD_X6;
      if (!c.dry && (op[9] > 0 || op[10] > 0))
        fc_conv_amax_hint(op[9] > 0 ? P<const unsigned>(c, op[9] - 1) : nullptr, op[10] > 0 ? P<const unsigned>(c, op[10] - 1) : nullptr);
      if (op[4] < 0) {
        const int64_t n = c.dims[op[6]];
        if (!want_ws(c, s, fc_conv_wgrad_ws_bytes(n, 1, Cin, Cout, fl))) return 0;
        return fc_conv_wgrad(P<const float>(c, op[2]), P<const float>(c, op[3]), nullptr, nullptr, P<float>(c, op[5]), n, n, 1, Cin, Cout, fl,
                             c.ws[s], c.ws_bytes[s], st);
      }
      const int64_t* m = c.maps + op[4] * MAPW;
      const int K = (int)m[2];
      if (!want_ws(c, s, fc_conv_wgrad_ws_bytes(m[1], K, Cin, Cout, fl))) return 0;
      if ((m[19] & 4) && Cin % 64 == 0 && Cout % 64 == 0)
        return fc_conv_wgrad_pairs(P<const float>(c, op[2]), P<const float>(c, op[3]), reinterpret_cast<const int*>(m[9]),
                                   reinterpret_cast<const int*>(m[10]), reinterpret_cast<const int*>(m[12]), P<float>(c, op[5]), m[0], m[1],
                                   K, Cin, Cout, fl, c.ws[s], c.ws_bytes[s], st);
      return fc_conv_wgrad(P<const float>(c, op[2]), P<const float>(c, op[3]), reinterpret_cast<const int*>(m[3]), nullptr,
                           P<float>(c, op[5]), m[0], m[1], K, Cin, Cout, fl, c.ws[s], c.ws_bytes[s], st);
    }
    case OP_BN_BWD: {  // x, y, gy, n(dim), C, mean, var, cnt, eps, gamma, beta, act, gx, gres, sums, gy2 + 1 | 0, producer of gy + 1 | 0
      const int64_t n = c.dims[op[5]];
      const int C = (int)op[6];
      const float eps = (float)as_double(op[10]);
      // the backward-data convolution that wrote gy may have left this layer's two reductions in its epilogue (word 18: its index
      // + 1 in this list; fc_conv_fwd_bn_bwd_stats)
      const float* part = nullptr;
      int64_t nbp = 0;
      if (op[18] > 0) {
        const int64_t* pop = c.ops + (op[18] - 1) * OPW;
        nbp = (pop[10] > 0 && pop[11] > 0) ? stats_blocks_of(c, pop) : 0;
        if (nbp > 0) part = P<const float>(c, pop[10] - 1);
      }
      if (!want_ws(c, s, fc_bn_train_ws_bytes(n, C))) return 0;
      if (op[19] > 0) fc_amax_out_hint(P<unsigned>(c, op[19] - 1));         // word 19: amax word of gx + 1 | 0
      return fc_bn_train_bwd(P<const float>(c, op[2]), P<const float>(c, op[3]), P<const float>(c, op[4]),
                             op[17] > 0 ? P<const float>(c, op[17] - 1) : nullptr, n, C, P<const float>(c, op[7]), P<const float>(c, op[8]),
                             P<const float>(c, op[9]), eps, P<const float>(c, op[11]), P<const float>(c, op[12]), (int)op[13],
                             P<float>(c, op[14]), P<float>(c, op[15]), P<float>(c, op[16]), part, nbp, c.bn_small_elems, c.ws[s],
                             c.ws_bytes[s], st);
    }
    case OP_NORM_BWD: {  // x, y, gy, seg, n(dim), C, nseg(dim), mean, var, cnt, eps, gamma, beta, act, gx, gres, sums
      const int64_t n = c.dims[op[6]];
      const int C = (int)op[7], nseg = (int)c.dims[op[8]];
      if (!want_ws(c, s, fc_norm_act_bwd_ws_bytes(n, C, nseg))) return 0;
      return fc_norm_act_bwd(P<const float>(c, op[2]), P<const float>(c, op[3]), P<const float>(c, op[4]), P<const int>(c, op[5]),
                             op[5] >= 0 ? 4 : 0, n, C, nseg, P<const float>(c, op[9]), P<const float>(c, op[10]), P<const float>(c, op[11]),
                             (float)as_double(op[12]), P<const float>(c, op[13]), P<const float>(c, op[14]), (int)op[15], P<float>(c, op[16]),
                             P<float>(c, op[17]), P<float>(c, op[18]), c.ws[s], c.ws_bytes[s], st);
    }
    case OP_MAXPOOL_BWD: {  // gout, arg, map, C, gin
      const int64_t* m = c.maps + op[4] * MAPW;
      if (c.dry) return 0;
      const int C = (int)op[5];
      FC_HIP(hipMemsetAsync(P<float>(c, op[6]), 0, sizeof(float) * m[0] * C, st));
      return fc_maxpool_bwd(P<const float>(c, op[2]), P<const int>(c, op[3]), m[1], C, P<float>(c, op[6]), st);
    }
    case OP_STEM_WGRAD: {  // col, gout, map, gW
      const int64_t* m = c.maps + op[4] * MAPW;
      if (!want_ws(c, s, fc_stem_conv_wgrad_ws_bytes(m[1], (int)m[2]))) return 0;
      return fc_stem_conv_wgrad(P<const float>(c, op[2]), P<const float>(c, op[3]), P<float>(c, op[5]), m[1], (int)m[2], c.ws[s],
                                c.ws_bytes[s], st);
    }
    case OP_GATHER: {  // src, idx, n(dim), C, dst
      if (c.dry) return 0;
      return fc_gather_rows(P<const float>(c, op[2]), P<const int>(c, op[3]), c.dims[op[4]], (int)op[5], P<float>(c, op[6]), st);
    }
    case OP_ADD: {  // dst += src over n(dim) * C floats (C % 4 == 0)
      if (c.dry) return 0;
      const int64_t n4 = c.dims[op[4]] * op[5] / 4;
      if (n4 > 0) k_add_inplace<<<(unsigned)fc_cdiv(n4, 256), 256, 0, st>>>(P<float>(c, op[2]), P<const float>(c, op[3]), n4);
      FC_CHECK_LAUNCH();
      return 0;
    }
    case OP_SMALL_GRADS: {  // desc (device), first entry, n entries
      if (c.dry) return 0;
      if (op[4] > 0) k_small_grads<<<(unsigned)op[4], 128, 0, st>>>(P<const long long>(c, op[2]) + 8 * op[3], (int)op[4]);
      FC_CHECK_LAUNCH();
      return 0;
    }
    case OP_COL_SUM: {  // x (n, C), n(dim), C, dst (C)
      if (c.dry) return 0;
      k_col_sum<<<(unsigned)op[4], 1024, 0, st>>>(P<const float>(c, op[2]), c.dims[op[3]], (int)op[4], P<float>(c, op[5]));
      FC_CHECK_LAUNCH();
      return 0;
    }
    case OP_ROW_SUM: {  // x (n), n(dim), dst (1)
      if (c.dry) return 0;
      k_col_sum<<<1, 1024, 0, st>>>(P<const float>(c, op[2]), c.dims[op[3]], 1, P<float>(c, op[4]));
      FC_CHECK_LAUNCH();
      return 0;
    }
    case OP_PERMUTE_GENT: {  // src (Cin, 8 Cout), dst (8, Cin, Cout), Cin, Cout
      if (c.dry) return 0;
      const int64_t total = 8 * op[4] * op[5];
      k_permute_gent<<<(unsigned)fc_cdiv(total, 256), 256, 0, st>>>(P<const float>(c, op[2]), P<float>(c, op[3]), (int)op[4], (int)op[5]);
      FC_CHECK_LAUNCH();
      return 0;
    }
    case OP_HEAD_WFIN: {  // part, nl, R, ld, n_reg, n_cls, g_cent, g_reg, g_cls, bias_part | -1, g_bias | -1
      if (c.dry) return 0;
      const int total = (int)(op[4] * (1 + op[6] + op[7]));
      k_head_wfin<<<(unsigned)fc_cdiv(total, 256), 256, 0, st>>>(P<const float>(c, op[2]), (int)op[3], (int)op[4], (int)op[5], (int)op[6],
                                                                 (int)op[7], P<float>(c, op[8]), P<float>(c, op[9]), P<float>(c, op[10]),
                                                                 P<const float>(c, op[11]), P<float>(c, op[12]));
      FC_CHECK_LAUNCH();
      return 0;
    }
    case OP_AMAX: {  // x, n(dim), C, slot: max |x| over n * C floats -> slot word 0 (fc_amax)
      if (c.dry) return 0;
      return fc_amax(P<const float>(c, op[2]), c.dims[op[3]] * op[4], P<unsigned>(c, op[5]), st);
    }
    case OP_CLEAR: {  // dst, bytes: zero-fill (the amax words a pass's producers fold into)
      if (c.dry) return 0;
      FC_HIP(hipMemsetAsync(P<void>(c, op[2]), 0, (size_t)op[3], st));
      return 0;
    }
    case OP_COPY: {  // dst, src, n(dim), C
      if (c.dry) return 0;
      FC_HIP(hipMemcpyAsync(P<float>(c, op[2]), P<const float>(c, op[3]), sizeof(float) * c.dims[op[4]] * op[5], hipMemcpyDeviceToDevice, st));
      return 0;
    }
    default:
      return FC_EINVAL;
  }
}

}  // namespace

extern "C" {

int fc_exec_op_words(void) { return OPW; }
int fc_exec_map_words(void) { return MAPW; }

// Runs operators [op_begin, op_end) of `ops` (HOST array, fc_exec_op_words() int64 per operator; layouts: fcaf3d_amd/executor.py).
// addr / dims / maps: HOST arrays of device addresses, row counts and kernel-map descriptors the operators index.
// streams / ws / ws_bytes: 3 entries each (0 main, 1 head branch, 2 weight gradients).  A first pass sizes the scratch space of
// every operator; if a stream's workspace is too small NOTHING is launched, ws_need[3] holds the required sizes and the call
// returns -2.  cfg[0] = bn_small_elems (functional.BN_SMALL_ELEMS), cfg[1] = kernel-variant flags (functional.FLAGS), cfg[2] != 0:
// probe (below), cfg[3] != 0: the sizing pass ONLY (ws_need is filled, -2 if a workspace is too small, nothing is launched).
int fc_exec(const int64_t* ops, int64_t op_begin, int64_t op_end, const int64_t* addr, const int64_t* dims, const int64_t* maps,
            const int64_t* streams, const int64_t* ws, const int64_t* ws_bytes, int64_t* ws_need, const int64_t* cfg) {
  if (!ops || op_begin < 0 || op_end < op_begin) return FC_EINVAL;
  int rc = ensure_events();
  if (rc) return rc;
  Ctx c;
  c.ops = ops;
  c.addr = addr; c.dims = dims; c.maps = maps;
  for (int i = 0; i < NSTREAM; ++i) {
    c.streams[i] = reinterpret_cast<hipStream_t>(streams[i]);
    c.ws[i] = reinterpret_cast<void*>(ws[i]);
    c.ws_bytes[i] = ws_bytes[i];
    c.need[i] = 0;
  }
  c.bn_small_elems = cfg[0];
  c.flags = (int)cfg[1];
  c.probe = cfg[2] != 0;
  c.dry = true;
  for (int64_t i = op_begin; i < op_end; ++i) {
    rc = run_op(c, ops + i * OPW);
    if (rc) return rc;
  }
  bool ok = true;
  for (int i = 0; i < NSTREAM; ++i) {
    if (ws_need) ws_need[i] = c.need[i];
    if (c.need[i] > c.ws_bytes[i]) ok = false;
  }
  if (!ok) return FC_EWS;
  if (cfg[3]) return 0;
  c.dry = false;
  for (int64_t i = op_begin; i < op_end; ++i) {
    rc = run_op(c, ops + i * OPW);
    if (rc) return rc;
  }
  return 0;
}

// Probe read-out (cfg[2] != 0 in fc_exec: a HIP-event pair brackets every convolution operator, on the stream it is launched
// on).  Call after the device has drained: ms[i] = duration of record i, meta[8 i ..] = {map index or -1, direction, n_in, n_out,
// K, Cin, Cout, 1 if the per-offset pair-list route ran}; returns the number of records (at most cap are written) and forgets them.
int64_t fc_exec_probe_read(float* ms, int64_t* meta, int64_t cap) {
  const int64_t n = (int64_t)g_probe.size();
  for (int64_t i = 0; i < n; ++i) {
    ProbeRec& r = g_probe[i];
    if (i < cap) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) t = -1.f;
      ms[i] = t;
      __builtin_memcpy(meta + 8 * i, r.meta, sizeof r.meta);
    }
    g_probe_pool.push_back(r.a);
    g_probe_pool.push_back(r.b);
  }
  g_probe.clear();
  return n;
}

}  // extern "C"
