// FCAF3D target assignment for a whole batch in four launches on gfx950 (integer atomics only —
// deterministic), replacing the ~80 torch elementwise launches PER SCENE of
// Fcaf3DAssigner.assign (mmdet3d/models/dense_heads/fcaf3d_neck_with_head.py:394-466) and
// compute_centerness (:377-384), incl. rotation_3d_in_axis(axis=2) (core/bbox/structures/utils.py:21-61).
//
//   k_count : per location, per GT box of its scene: inside test -> counts[scene][box][level]++
//   k_best  : per box: best pyramid level = last level before the first one with < `limit` inside locations
//   k_kth   : per box: (topk+1)-th largest centerness among its candidates (inside & on best level)
//   k_final : per location: among boxes where it is a top-k candidate pick the smallest volume (first wins ties)
#include "fc_common.h"
#pragma clang fp contract(off)     // the same products at every call site: `centerness > kth` compares equal values

#define FLOAT_MAX_VOL 1e8f

// The small per-box tables (counts, best, kth, trig) are handed from one launch to the next through the workspace.  Read them
// past the CU's vector L1 (agent-scope loads, `global_load ... sc1`): see the note at fc_assign_targets.
__device__ static inline int ld_i(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline float ld_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Face6 { float d[6]; };

// distances of point p to the 6 faces of box b = [cx,cy,cz,w,l,h,yaw] in the box frame
// (ca, sa) = cos / sin of -yaw: computed ONCE per box by box_trig() at every call site (r3: they were re-evaluated for
// every (location, box) pair — 15 M cosf + sinf per step), the same function everywhere so that `centerness > kth`
// still compares bit-identical values
__device__ static inline void box_trig(const float* __restrict__ b, float* ca, float* sa) {
  const float a = -b[6];
  *ca = cosf(a);
  *sa = sinf(a);
}

__device__ static inline Face6 face_distances(const float* __restrict__ b, float ca, float sa, float px, float py, float pz) {
  float sx = px - b[0], sy = py - b[1], sz = pz - b[2];
  float rx = sx * ca + sy * sa;
  float ry = -sx * sa + sy * ca;
  float cx = b[0] + rx, cy = b[1] + ry, cz = b[2] + sz;
  Face6 f;
  f.d[0] = cx - b[0] + b[3] / 2;
  f.d[1] = b[0] + b[3] / 2 - cx;
  f.d[2] = cy - b[1] + b[4] / 2;
  f.d[3] = b[1] + b[4] / 2 - cy;
  f.d[4] = cz - b[2] + b[5] / 2;
  f.d[5] = b[2] + b[5] / 2 - cz;
  return f;
}

__device__ static inline bool is_inside(const Face6& f) {
  float m = fminf(fminf(fminf(f.d[0], f.d[1]), fminf(f.d[2], f.d[3])), fminf(f.d[4], f.d[5]));
  return m > 0.f;
}

__device__ static inline float centerness_of(const Face6& f) {
  float v = fminf(f.d[0], f.d[1]) / fmaxf(f.d[0], f.d[1]);
  v = v * fminf(f.d[2], f.d[3]) / fmaxf(f.d[2], f.d[3]);
  v = v * fminf(f.d[4], f.d[5]) / fmaxf(f.d[4], f.d[5]);
  return sqrtf(v);
}

__global__ void k_count(const float* __restrict__ pts, const int* __restrict__ scene, const int* __restrict__ level, int64_t N,
                        const float* __restrict__ boxes, const int* __restrict__ box_count, int M, int L,
                        const float* __restrict__ trig, int* __restrict__ counts) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int s = scene[i], l = level[i];
  float px = pts[i * 3], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
  int m = box_count[s];
  for (int j = 0; j < m; ++j) {
    const float* b = boxes + ((int64_t)s * M + j) * 7;
    Face6 f = face_distances(b, ld_f(&trig[2 * (s * M + j)]), ld_f(&trig[2 * (s * M + j) + 1]), px, py, pz);
    if (is_inside(f)) atomicAdd(&counts[((int64_t)s * M + j) * L + l], 1);
  }
}

// (cos, sin) of -yaw of every padded box slot, once per step
__global__ void k_box_trig(const float* __restrict__ boxes, int BM, float* __restrict__ trig, int* __restrict__ counts, int L) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= BM) return;
  box_trig(boxes + (int64_t)t * 7, &trig[2 * t], &trig[2 * t + 1]);
  for (int l = 0; l < L; ++l) counts[(int64_t)t * L + l] = 0;      // (the per-level inside counters k_count adds to)
}

__global__ void k_best(const int* __restrict__ counts, int BM, int L, int limit, int* __restrict__ best) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= BM) return;
  int first_starved = -1;
  for (int l = 0; l < L; ++l)
    if (ld_i(&counts[(int64_t)t * L + l]) < limit) { first_starved = l; break; }
  int b;
  if (first_starved < 0) b = L - 1;
  else b = first_starved - 1 < 0 ? 0 : first_starved - 1;
  best[t] = b;
}

// one block per (scene, box); rows of (level, scene) are order[seg_start[l*B+s] .. seg_start[l*B+s+1]).
// Pass 1 collects the candidates' centerness values into LDS (one scan of the scene's rows on that level);
// the (topk+1)-th largest is then peeled off the LDS list value by value (with multiplicity).  More than
// KTH_CAP candidates (only possible for a huge box on a fine level) falls back to rescanning global memory.
#define KTH_CAP 8192
__global__ __launch_bounds__(256) void k_kth(const float* __restrict__ pts, const float* __restrict__ boxes,
                                             const int* __restrict__ box_count, const int* __restrict__ best,
                                             const int* __restrict__ order, const int* __restrict__ seg_start, int B, int M,
                                             int L, int topk, const float* __restrict__ trig, float* __restrict__ kth) {
  const int s = blockIdx.x / M, j = blockIdx.x % M;
  if (j >= box_count[s]) return;
  const float* b = boxes + ((int64_t)s * M + j) * 7;
  const float ca = ld_f(&trig[2 * (s * M + j)]), sa = ld_f(&trig[2 * (s * M + j) + 1]);
  const int l = ld_i(&best[s * M + j]);
  const int r0 = seg_start[l * B + s], r1 = seg_start[l * B + s + 1];
  // n_scene = all locations of the scene over all levels (torch.topk is taken over all of them, padded with -1)
  int n_scene = 0;
  for (int q = 0; q < L; ++q) n_scene += seg_start[q * B + s + 1] - seg_start[q * B + s];
  const int rank = min(topk + 1, n_scene);
  __shared__ float cand[KTH_CAP];
  __shared__ int ncand_s;
  __shared__ float red_v[4];
  __shared__ int red_c[4];
  if (threadIdx.x == 0) ncand_s = 0;
  __syncthreads();
  // four rows in flight per thread: their `order` entries, then their coordinates, are requested together (r3: one
  // dependent order -> point chain per iteration over up to 55k rows made this launch 142 us)
  for (int tb = r0 + threadIdx.x; tb < r1; tb += 4 * 256) {
    int idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) idx[u] = order[tb + 256 * u < r1 ? tb + 256 * u : tb];
    float px[4], py[4], pz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      px[u] = pts[(int64_t)idx[u] * 3]; py[u] = pts[(int64_t)idx[u] * 3 + 1]; pz[u] = pts[(int64_t)idx[u] * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (tb + 256 * u >= r1) continue;
      Face6 f = face_distances(b, ca, sa, px[u], py[u], pz[u]);
      if (!is_inside(f)) continue;
      int p = atomicAdd(&ncand_s, 1);
      if (p < KTH_CAP) cand[p] = centerness_of(f);
    }
  }
  __syncthreads();
  const int ncand = ncand_s;
  const bool in_lds = ncand <= KTH_CAP;
  float prev = INFINITY;        // values >= prev are already accounted for
  int seen = 0;
  float answer = -1.f;
  for (int round = 0; round <= topk; ++round) {
    // largest candidate value strictly below `prev`, and its multiplicity
    float best_v = -2.f;
    int cnt = 0;
    if (in_lds) {
      for (int t = threadIdx.x; t < ncand; t += 256) {
        float c = cand[t];
        if (!(c < prev)) continue;
        if (c > best_v) { best_v = c; cnt = 1; }
        else if (c == best_v) ++cnt;
      }
    } else {
      for (int t = r0 + threadIdx.x; t < r1; t += 256) {
        int i = order[t];
        Face6 f = face_distances(b, ca, sa, pts[(int64_t)i * 3], pts[(int64_t)i * 3 + 1], pts[(int64_t)i * 3 + 2]);
        if (!is_inside(f)) continue;
        float c = centerness_of(f);
        if (!(c < prev)) continue;
        if (c > best_v) { best_v = c; cnt = 1; }
        else if (c == best_v) ++cnt;
      }
    }
    // wave reduce (max value, summed multiplicity of that value)
    for (int off = 32; off > 0; off >>= 1) {
      float ov = __shfl_xor(best_v, off, 64);
      int oc = __shfl_xor(cnt, off, 64);
      if (ov > best_v) { best_v = ov; cnt = oc; }
      else if (ov == best_v) cnt += oc;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red_v[threadIdx.x >> 6] = best_v; red_c[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    float bv = red_v[0];
    int bc = red_c[0];
    for (int w = 1; w < 4; ++w) {
      if (red_v[w] > bv) { bv = red_v[w]; bc = red_c[w]; }
      else if (red_v[w] == bv) bc += red_c[w];
    }
    if (bv < 0.f) break;                    // candidates exhausted: the rank-th value is the -1 padding
    seen += bc;
    if (seen >= rank) { answer = bv; break; }
    prev = bv;
  }
  if (threadIdx.x == 0) kth[s * M + j] = answer;
}

__global__ void k_final(const float* __restrict__ pts, const int* __restrict__ scene, const int* __restrict__ level, int64_t N,
                        const float* __restrict__ boxes, const long long* __restrict__ labels,
                        const int* __restrict__ box_count, const int* __restrict__ best, const float* __restrict__ kth, int M,
                        const float* __restrict__ trig, float* __restrict__ ct_out, float* __restrict__ bt_out, long long* __restrict__ lab_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int s = scene[i], l = level[i];
  float px = pts[i * 3], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
  int m = box_count[s];
  float best_vol = FLOAT_MAX_VOL;
  int owner = 0;
  float owner_c = 0.f;
  for (int j = 0; j < m; ++j) {
    const float* b = boxes + ((int64_t)s * M + j) * 7;
    if (ld_i(&best[s * M + j]) != l) continue;
    Face6 f = face_distances(b, ld_f(&trig[2 * (s * M + j)]), ld_f(&trig[2 * (s * M + j) + 1]), px, py, pz);
    if (!is_inside(f)) continue;
    float c = centerness_of(f);
    if (!(c > ld_f(&kth[s * M + j]))) continue;
    float vol = b[3] * b[4] * b[5];
    if (vol < best_vol) { best_vol = vol; owner = j; owner_c = c; }
  }
  bool pos = best_vol != FLOAT_MAX_VOL;
  lab_out[i] = pos ? labels[(int64_t)s * M + owner] : -1;
  ct_out[i] = pos ? owner_c : 0.f;
  const float* ob = boxes + ((int64_t)s * M + owner) * 7;
  for (int e = 0; e < 7; ++e) bt_out[i * 7 + e] = m > 0 ? ob[e] : 0.f;
}

extern "C" {

int64_t fc_assign_ws_bytes(int B, int M, int L) {
  return (int64_t)B * M * (L + 4) * 4 + 256;
}

// r3 finding (tools/trace_det.py): with this call on the coordinate stream, overlapping the main stream's split-bf16
// convolutions of the previous step, ~30 % of the calls returned a few dozen wrong rows — k_final had read an OLD kth / best
// entry although k_kth had long finished: the tables sit at the same workspace address in every call, a CU's vector L1 keeps
// the line from the previous call, and under that overlap the invalidate at the kernel boundary did not cover every CU the
// late-placed waves landed on (never seen beside the fp32 kernels or on an idle chip: 0 of 60 calls).  The consumers now
// read the tables past the L1 (ld_i / ld_f: 0 of 36 calls wrong), and the counters are zeroed by k_box_trig, not a memset.
// points (N,3); scene/level (N) int32; boxes (B,M,7) [cx,cy,cz,w,l,h,yaw] gravity centre, padded; labels (B,M) int64;
// box_count (B); order (N) = location rows grouped by (level, scene); seg_start (L*B+1) offsets into `order`.
// outputs: centerness targets (N) (0 for background), box targets (N,7), labels (N) int64 (-1 = background).
int fc_assign_targets(const float* points, const int* scene, const int* level, int64_t N, const float* boxes,
                      const long long* labels, const int* box_count, int B, int M, int L, const int* order,
                      const int* seg_start, int limit, int topk, float* centerness_t, float* bbox_t, long long* labels_out,
                      void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (N < 0 || B < 1 || M < 1 || L < 1 || topk < 0) return FC_EINVAL;
  if (ws_bytes < fc_assign_ws_bytes(B, M, L)) return FC_EWS;
  if (N == 0) return FC_OK;
  int* counts = (int*)ws;
  int* best = counts + (int64_t)B * M * L;
  float* kth = (float*)(best + (int64_t)B * M);
  float* trig = kth + (int64_t)B * M;
  unsigned g = (unsigned)fc_cdiv(N, 256);
  k_box_trig<<<(unsigned)fc_cdiv(B * M, 64), 64, 0, stream>>>(boxes, B * M, trig, counts, L);
  FC_CHECK_LAUNCH();
  k_count<<<g, 256, 0, stream>>>(points, scene, level, N, boxes, box_count, M, L, trig, counts);
  FC_CHECK_LAUNCH();
  k_best<<<(unsigned)fc_cdiv(B * M, 64), 64, 0, stream>>>(counts, B * M, L, limit, best);
  FC_CHECK_LAUNCH();
  k_kth<<<(unsigned)(B * M), 256, 0, stream>>>(points, boxes, box_count, best, order, seg_start, B, M, L, topk, trig, kth);
  FC_CHECK_LAUNCH();
  k_final<<<g, 256, 0, stream>>>(points, scene, level, N, boxes, labels, box_count, best, kth, M, trig, centerness_t, bbox_t,
                                 labels_out);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"
