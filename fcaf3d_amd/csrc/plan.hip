// The coordinate phase of one step as native host loops (r6): collate + voxel hash + the whole pyramid of coordinate sets,
// then every kernel map and derived table the network body will use — two C-ABI calls per step instead of ~420 launches
// driven from Python with ~25 blocking count read-backs (profiles/r5_notes.md section 16: the step was host-bound on them
// whenever a round trip on the coordinate stream took 0.3 ms longer than usual).
//
//   fc_plan_levels  points of B scenes -> [cm0, m1, m2, L1..Lnl]: the collate of single_stage_sparse.py:34-37 and the strided
//                   output sets of me_resnet.py:19-24, :56-62, built as ONE chain with DEVICE-resident row counts (every kernel
//                   is launched for the upper bound and reads the live count from the previous set's meta word), then ONE
//                   read-back of all counts (rows per set, rows per set and scene, range flags).
//   fc_plan_maps    with those counts: every kernel map of the backbone and the neck (generated sets by index arithmetic),
//                   mask-sorted tables (native LSD radix argsort), pair lists, transposed tables, union rows, the head's
//                   location / scene / level arrays — enqueued back to back, then ONE read-back (pair-list counts, union hits).
//
// Results are the sets, tables and row orders of the per-operator path (fcaf3d_amd/sparse.py -> csrc/coords.hip), bit for
// bit: tests/test_gpu_plan.py compares every buffer.  Reference: what this replaces is ME's CoordinateManager work inside
// ME.SparseTensor(...) and every ME.Minkowski* call of extract_feat (mmdet3d/models/detectors/single_stage_sparse.py:32-40).
#include "fc_common.h"
#include "../../include/fcaf3d_hip.h"
#include <string.h>
#include <vector>

namespace {

// ---- layout of the host tables (mirrored in fcaf3d_amd/plan.py) ------------------------------------------------------------
enum : int {                        // cfg words
  C_B = 0, C_NL, C_NFEAT, C_VS /*double bits*/, C_FEATDIV /*double bits*/, C_TOTAL, C_BACKWARD, C_SORT_MIN, C_PAIR_ROWS, C_PTS_THR,
  C_TARGETS, C_COORDS_IN, C_FEATS_IN, C_PT_STRIDE, C_NECK, C_VS_HEAD /*double bits: the head's voxel size*/, C_PROBE, CFGW = 24
};
constexpr int HDR = 16;             // out[0..15]: header
enum : int { H_S = 0, H_NEED2, H_PRUNE, H_NMAPS, H_STRUCT, H_NALL, H_F0, H_TGT_PTS, H_TGT_SCENE, H_TGT_LEVEL, H_TGT_ORDER, H_TGT_SEG,
             H_NHEAD, H_BAD };
constexpr int SETW = 8;             // per set: coords, n, stride, keys, vals, cap, gen_parent (set index or -1), union rows
enum : int { S_COORDS = 0, S_N, S_STRIDE, S_KEYS, S_VALS, S_CAP, S_PARENT, S_ROWS };
constexpr int MAPR = 64;            // per map record
enum : int {
  MW_IN = 0, MW_OUT, MW_K, MW_NIN, MW_NOUT, MW_NBR, MW_NBRT, MW_SORT, MW_SORTI, MW_SORTT, MW_SORTTI, MW_PI, MW_PO, MW_POS, MW_CNT, MW_TILES,
  MW_TPI, MW_TPO, MW_TPOS, MW_TCNT, MW_TTILES, MW_FLAGS /*1 sort_rows, 2 use_pairs, 4 dense*/, MW_DESC_F = 24 /*20 words: desc(conv, backward=false)*/,
  MW_DESC_B = 44 /*20 words: desc(conv, backward=true)*/
};
constexpr int MAXSETS = 24, MAXLV = 8;

inline double as_double(int64_t v) { double d; memcpy(&d, &v, 8); return d; }
inline int64_t next_pow2(int64_t n) { int64_t p = 2; while (p < n) p *= 2; return p; }

struct Bump {                       // bump allocation inside a caller-owned arena; base == nullptr: sizing pass
  char* base; int64_t off;
  void* take(int64_t bytes) {
    off = fc_align(off, 256);
    void* p = base ? base + off : nullptr;
    off += bytes > 0 ? bytes : 0;
    return p;
  }
  template <typename T> T* arr(int64_t n) { return (T*)take(n * (int64_t)sizeof(T)); }
};

// one event per thread for the read-backs (created once, never destroyed)
thread_local hipEvent_t t_ev = nullptr;
int sync_readback(void* host, const void* dev, int64_t bytes, hipStream_t s) {
  if (!t_ev) FC_HIP(hipEventCreateWithFlags(&t_ev, hipEventDisableTiming));
  FC_HIP(hipMemcpyAsync(host, dev, (size_t)bytes, hipMemcpyDeviceToHost, s));
  FC_HIP(hipEventRecord(t_ev, s));
  FC_HIP(hipEventSynchronize(t_ev));
  return 0;
}

// ---- probe: HIP-event brackets around every launch (group) of a plan, with its compulsory bytes — bench.py's `roofline.hbm_kernels`
// (cfg[C_PROBE] != 0; one thread at a time; read out by fc_plan_probe_read after the device has drained) ----------------------------
enum : int { PK_TABLES = 0, PK_VOXELIZE, PK_FLAGS, PK_FINALIZE, PK_GEN, PK_KMAPS, PK_CHILDREN, PK_FILL, PK_TRANSPOSE, PK_MASKS, PK_RADIX,
             PK_PERMUTE, PK_PAIRS, PK_GENROWS, PK_HEAD, PK_KINDS };
struct PlanProbe { int kind; double bytes; hipEvent_t a, b; };
std::vector<PlanProbe> g_pp;
std::vector<hipEvent_t> g_pp_pool;
hipEvent_t pp_event() {
  if (!g_pp_pool.empty()) { hipEvent_t e = g_pp_pool.back(); g_pp_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
struct Bracket {                     // RAII: records the closing event when the launch (group) has been enqueued
  hipStream_t s; bool on; size_t idx;
  Bracket(bool probe, int kind, double bytes, hipStream_t st) : s(st), on(probe), idx(0) {
    if (!on) return;
    PlanProbe p{kind, bytes, pp_event(), pp_event()};
    if (!p.a || !p.b) { on = false; return; }
    (void)hipEventRecord(p.a, s);
    idx = g_pp.size();
    g_pp.push_back(p);
  }
  ~Bracket() { if (on) (void)hipEventRecord(g_pp[idx].b, s); }
};

// ---- batched launches ------------------------------------------------------------------------------------------------------------
// Most of the phase is many small independent jobs of one kind (ten kernel maps, thirteen pair lists, five argsorts ...): each kind
// is ONE launch over the concatenated blocks of its jobs; a block finds its job by a scan of the (<= 24) first-block entries, which
// sit in the kernel arguments (scalar loads).
// (r6: a 2.1 KB argument block — 24 jobs of 80 bytes — made the launches misbehave on this stack, a 1.7 KB one did not: the
// batches stay below 1.5 KB, static_assert below; four pyramid levels need at most 16 jobs of a kind)
constexpr int MAXJ = 16;
template <typename J> struct Batch {
  J j[MAXJ];
  int64_t blk0[MAXJ + 1];
  int count;
  bool overflow;
  Batch() : count(0), overflow(false) { blk0[0] = 0; }
  void add(const J& job, int64_t blocks) {
    if (count >= MAXJ) { overflow = true; return; }
    j[count] = job;
    blk0[count + 1] = blk0[count] + (blocks > 0 ? blocks : 0);
    ++count;
  }
  int64_t blocks() const { return blk0[count]; }
};
template <typename J> __device__ __forceinline__ int batch_find(const Batch<J>& b, int64_t blk, int64_t* local) {
  int i = 0;
  while (i + 1 < b.count && blk >= b.blk0[i + 1]) ++i;
  *local = blk - b.blk0[i];
  return i;
}

// ---- stage 1 kernels: device-resident counts ---------------------------------------------------------------------------------
// meta (device ints): per set s: [8s + 0] rows, [8s + 2] range flag; then rows per (set, scene).
constexpr int METAW = 8;
constexpr int CH = 32;              // scenes per launch of the point kernel
struct SceneArgs { const float* p[CH]; int n[CH]; int off[CH]; int b0, stride, nfeat; float vs, feat_div; };

__device__ inline int4 quant(int4 c, int q) {
  if (q > 1) { c.y = fc_floor_div(c.y, q) * q; c.z = fc_floor_div(c.z, q) * q; c.w = fc_floor_div(c.w, q) * q; }
  return c;
}
__device__ inline bool out_of_range(int4 c) {
  return c.x < 0 || c.x > 32767 || c.y < -FC_COORD_LIMIT || c.y > FC_COORD_LIMIT || c.z < -FC_COORD_LIMIT || c.z > FC_COORD_LIMIT ||
         c.w < -FC_COORD_LIMIT || c.w > FC_COORD_LIMIT;
}
// capacity of the table of a set whose INPUT has n rows: next_pow2(2 n) — the per-operator path's rule (sparse.CoordMap.from_coords)
__host__ __device__ inline unsigned long long table_mask(int64_t n_in) {
  int64_t cap = 2;
  while (cap < 2 * n_in) cap *= 2;
  return (unsigned long long)(cap - 1);
}
__device__ inline void hash_insert(unsigned long long* keys, int* vals, unsigned long long mask, int4 c, int i, int* slot) {
  const unsigned long long key = fc_pack(c.x, c.y, c.z, c.w);
  unsigned long long h = fc_mix(key) & mask;
  while (true) {
    const unsigned long long prev = atomicCAS(&keys[h], FC_EMPTY_KEY, key);
    if (prev == FC_EMPTY_KEY || prev == key) {
      atomicMin(&vals[h], i);       // first occurrence (smallest row) wins — SURVEY.md Appendix A.2
      slot[i] = (int)h;
      return;
    }
    h = (h + 1) & mask;
  }
}

// all tables of the chain in one launch: keys (S, cap) / vals (S, cap) contiguous, two key slots per thread
__global__ void k_plan_tables_init(unsigned long long* __restrict__ keys, int* __restrict__ vals, int64_t total) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < total) {
    *reinterpret_cast<ulonglong2*>(keys + i) = make_ulonglong2(FC_EMPTY_KEY, FC_EMPTY_KEY);
    *reinterpret_cast<int2*>(vals + i) = make_int2(0x7fffffff, 0x7fffffff);
  } else if (i < total) {
    keys[i] = FC_EMPTY_KEY; vals[i] = 0x7fffffff;
  }
}

// the collate (single_stage_sparse.py:34-36: floor(xyz / voxel_size), features / 255, batch index = scene) fused with the hash
// insert of level 0 — blockIdx.y = scene of the chunk.  (true fp32 division, as fc_voxelize.)
__global__ void k_plan_voxelize_insert(SceneArgs sc, int4* __restrict__ coords, float* __restrict__ feats, unsigned long long* keys,
                                       int* vals, unsigned long long mask, int* __restrict__ meta_s, int* __restrict__ slot) {
  const int j = blockIdx.y;
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= sc.n[j]) return;
  const float* p = sc.p[j] + (int64_t)l * sc.stride;
  const int i = sc.off[j] + l;
  int4 c;
  c.x = sc.b0 + j;
  c.y = (int)floorf(p[0] / sc.vs); c.z = (int)floorf(p[1] / sc.vs); c.w = (int)floorf(p[2] / sc.vs);
  coords[i] = c;
  for (int f = 0; f < sc.nfeat; ++f) feats[(int64_t)i * sc.nfeat + f] = p[3 + f] / sc.feat_div;
  if (out_of_range(c)) meta_s[2] = 1;
  hash_insert(keys, vals, mask, c, i, slot);
}

// level 0 from a pre-voxelised coordinate array
__global__ void k_plan_insert(const int4* __restrict__ coords, int64_t n, unsigned long long* keys, int* vals, unsigned long long mask,
                              int* __restrict__ meta_s, int* __restrict__ slot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  if (out_of_range(c)) meta_s[2] = 1;
  hash_insert(keys, vals, mask, c, (int)i, slot);
}

// winners: flags + the count of every 256-row sub-block (blocksums, what k_plan_finalize works in) and of every 1 024-row block
// (coarse).  No scan here: every block of k_plan_finalize adds up the (L2-resident, <= a few thousand) counts in front of it — and all
// of them, for the set's row count — itself.  (Until r6's last take the LAST block to finish scanned the counts in place: a
// device-scope fence and a same-line atomic per block, 28 us for an EMPTY set and ~60 of the ~100 us of a full one.)
__global__ __launch_bounds__(256) void k_plan_flags(const int* __restrict__ slot, const int* __restrict__ vals, const int* __restrict__ n_dev,
                                                    int64_t n_host, unsigned char* __restrict__ flags, int* __restrict__ blocksums,
                                                    int* __restrict__ coarse) {
  const int64_t n = n_dev ? *n_dev : n_host;
  if ((int64_t)blockIdx.x * 1024 >= n) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ int wsr[4][4];
  // the four dependent read pairs of a thread (slot, table value) are issued together
  int f[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = (int64_t)blockIdx.x * 1024 + r * 256 + threadIdx.x;
    f[r] = i < n ? (vals[slot[i]] == (int)i) : 0;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = (int64_t)blockIdx.x * 1024 + r * 256 + threadIdx.x;
    if (i < n) flags[i] = (unsigned char)f[r];
    const unsigned long long bal = __ballot(f[r]);
    if (lane == 0) wsr[r][w] = __popcll(bal);
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int r = threadIdx.x;
    blocksums[4 * blockIdx.x + r] = wsr[r][0] + wsr[r][1] + wsr[r][2] + wsr[r][3];
  } else if (threadIdx.x == 64) {
    int t = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) t += wsr[r][0] + wsr[r][1] + wsr[r][2] + wsr[r][3];
    coarse[blockIdx.x] = t;
  }
}

// positions + the set itself: winner row i -> out row p: coordinates (with its feature row for level 0), table value = p,
// rows-per-scene counters (rows of one scene are consecutive: one atomic per wave and scene) — and, fused, the hash insert of the
// NEXT set of the chain: row p of this set, quantised to the next stride, goes into the next table with value p (first occurrence
// = smallest p, as an insert pass over the finished set would give).  The block's first output row = the winners in front of it =
// the coarse counts of the 1 024-row blocks before its own + the sub-block counts inside that one; the set's row count (the next
// table's mask; block 0 leaves it in meta_s[0] for the kernels behind) = all coarse counts.
__global__ __launch_bounds__(256) void k_plan_finalize(const int4* __restrict__ coords, const int* __restrict__ n_dev, int64_t n_host, int q,
                                                       const unsigned char* __restrict__ flags, const int* __restrict__ blocksums,
                                                       const int* __restrict__ coarse, const int* __restrict__ slot, int* vals,
                                                       int4* __restrict__ out_coords, const float* __restrict__ feats_in,
                                                       float* __restrict__ feats_out, int nfeat, int* __restrict__ scene_cnt, int B,
                                                       int* __restrict__ meta_s, unsigned long long* next_keys, int* next_vals,
                                                       int* __restrict__ next_slot) {
  __shared__ int ws[4];
  __shared__ int red[2][4];
  const int64_t n = n_dev ? *n_dev : n_host;
  const int64_t base = (int64_t)blockIdx.x * 256;
  if (base >= n) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i = base + threadIdx.x;
  const int f = i < n ? (int)flags[i] : 0;
  const unsigned long long bal = __ballot(f);
  if (lane == 0) ws[w] = __popcll(bal);
  const int nc = (int)((n + 1023) >> 10), mine = (int)(blockIdx.x >> 2);
  int before = 0, all = 0;
  for (int j = threadIdx.x; j < nc; j += 256) {
    const int v = coarse[j];
    all += v;
    if (j < mine) before += v;
  }
  if (threadIdx.x < (int)(blockIdx.x & 3)) before += blocksums[4 * mine + threadIdx.x];
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o, 64);
    all += __shfl_xor(all, o, 64);
  }
  if (lane == 0) { red[0][w] = before; red[1][w] = all; }
  __syncthreads();
  int pre = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  const int total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  if (blockIdx.x == 0 && threadIdx.x == 0) meta_s[0] = total;
  const unsigned long long nmask = next_keys ? table_mask(total) : 0ull;
  for (int k = 0; k < w; ++k) pre += ws[k];
  int b = -1;
  if (f) {
    const int p = pre + __popcll(bal & ((1ull << lane) - 1ull));
    const int4 c = quant(coords[i], q);
    out_coords[p] = c;
    vals[slot[i]] = p;                             // the table now maps key -> row of the new set
    if (feats_out)
      for (int k = 0; k < nfeat; ++k) feats_out[(int64_t)p * nfeat + k] = feats_in[i * nfeat + k];
    b = c.x;
    if (next_keys) hash_insert(next_keys, next_vals, nmask, quant(c, 2 * q), p, next_slot);
  }
  unsigned long long rem = bal;                    // rows per scene
  while (rem) {
    const int leader = __ffsll((long long)rem) - 1;
    const int lb = __shfl(b, leader, 64);
    const unsigned long long same = __ballot(f && b == lb);
    if (lane == leader && lb >= 0 && lb < B) atomicAdd(&scene_cnt[lb], __popcll(same));
    rem &= ~same;
  }
}

// ---- stage 2 kernels (batched) ---------------------------------------------------------------------------------------------------
// kernel maps with the offsets computed in the kernel (x fastest; centred for odd kernels, {0..k-1} for even — Appendix A.3):
// nbr[k][o] = row of out_coords[o] + offset_k * stride in the table, or -1.  job blocks = K * ceil(n_out / 256), k-major.
struct KMapJob { const int4* oc; const unsigned long long* keys; const int* vals; unsigned long long mask; int* nbr; int64_t n_out; int K, ks, stride, nbx; };
__global__ void k_plan_kernel_maps(Batch<KMapJob> bt) {
  int64_t lb;
  const KMapJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const int k = (int)(lb / j.nbx);
  const int64_t o = (lb % j.nbx) * 256 + threadIdx.x;
  if (o >= j.n_out) return;
  const int ks = j.ks, c0 = (ks & 1) ? ks / 2 : 0;
  const int dx = (k % ks - c0) * j.stride, dy = ((k / ks) % ks - c0) * j.stride, dz = (k / (ks * ks) - c0) * j.stride;
  const int4 c = j.oc[o];
  j.nbr[(int64_t)k * j.n_out + o] = fc_lookup(j.keys, j.vals, j.mask, fc_pack(c.x, c.y + dx, c.z + dy, c.w + dz));
}

// kernel maps from a set of stride s onto the set of stride 2 s (the stem's and the levels' k3 s2 convolutions, the k2 s2 max-pool),
// built from the INPUT side (r6, last take): row i of the input lies at offset k of output row o exactly when out = in - offset_k is a
// voxel of the output set — per axis ONE candidate output if the coordinate is a multiple of 2 s (offset 0), two if not (offsets
// +s / -s), one for the k2 kernel — so 1 ... 8 probes of the OUTPUT set's table per input row (3.4 on average) instead of 27 probes
// of the input table per output row, 14 % of which hit: the stem's map 18.9 M probes -> 2.7 M.  The table (pre-filled with -1 by the
// launch in front) gets nbr[k][o] = i; every entry has one writer (input rows are unique voxels).  Same tables bit for bit.
struct SMapJob { const int4* ic; const unsigned long long* okeys; const int* ovals; unsigned long long omask; int* nbr; int64_t n_in, n_out; int ks, stride, nbx; };
__global__ void k_plan_strided_maps(Batch<SMapJob> bt) {
  int64_t lb;
  const SMapJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  // eight threads per input row, one per candidate (a, b, c): the probes of a row are in flight together
  const int64_t t = lb * 256 + threadIdx.x;
  const int64_t i = t >> 3;
  if (i >= j.n_in) return;
  const int a = (int)t & 1, b = ((int)t >> 1) & 1, c = ((int)t >> 2) & 1;
  const int4 p = j.ic[i];
  const int s = j.stride, s2 = 2 * j.stride, ks = j.ks;
  const int pc[3] = {p.y, p.z, p.w};
  int nc[3], cc[3][2], ki[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int q = fc_floor_div(pc[d], s2) * s2, r = pc[d] - q;      // r = 0 or s
    cc[d][0] = q; cc[d][1] = q + s2;
    if (ks == 2) { nc[d] = 1; ki[d][0] = r / s; ki[d][1] = 0; }     // offsets {0, s}: out = q
    else if (r == 0) { nc[d] = 1; ki[d][0] = 1; ki[d][1] = 0; }     // offset 0
    else { nc[d] = 2; ki[d][0] = 2; ki[d][1] = 0; }                 // out = q: offset +s (index 2); out = q + 2 s: offset -s (index 0)
  }
  if (a >= nc[0] || b >= nc[1] || c >= nc[2]) return;
  const int o = fc_lookup(j.okeys, j.ovals, j.omask, fc_pack(p.x, cc[0][a], cc[1][b], cc[2][c]));
  if (o >= 0) j.nbr[(int64_t)(ki[0][a] + ks * ki[1][b] + ks * ks * ki[2][c]) * j.n_out + o] = (int)i;
}

// row copies / fills of (K, n) tables.  mode 0: dst[k] = src[K - 1 - k] — the transposed table of a map of a set onto ITSELF with
// a centred odd kernel (offset k reversed is offset K - 1 - k: identical to fc_kernel_map_transpose); mode 1: dst = -1;
// mode 2: scatter dst[k][src[k][o]] = o (src (K, n), dst (K, n2)) — fc_kernel_map_transpose's second half; mode 3: dst = -1 as one flat
// run of K n entries, 16 bytes per thread
struct RowJob { const int* src; int* dst; int64_t n, n2; int K, mode, nbx; };
__global__ void k_plan_rows(Batch<RowJob> bt) {
  int64_t lb;
  const RowJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  if (j.mode == 3) {                                 // dst[0 .. K n) = -1, four entries per thread (dst is 16-byte aligned: bump allocator)
    const int64_t tot = (int64_t)j.K * j.n, e = (lb * 256 + threadIdx.x) * 4;
    if (e + 3 < tot) *reinterpret_cast<int4*>(j.dst + e) = make_int4(-1, -1, -1, -1);
    else for (int64_t q = e; q < tot; ++q) j.dst[q] = -1;
    return;
  }
  // four entries per thread, 256 apart (coalesced), their loads in flight together: at 4 bytes per thread a full chip holds ~2 MB in
  // flight — 1 TB/s at 2 us of latency (r6: the fills went 28 -> 10 us as 16-byte stores; the same for the copies and the scatters).
  // job blocks = K * ceil(nbx / 4)
  const int per = (j.nbx + 3) >> 2;
  const int k = (int)(lb / per);
  const int64_t i0 = (lb % per) * 1024 + threadIdx.x;
  int v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = i0 + 256 * u;
    v[u] = -1;
    if (i < j.n && j.mode != 1) v[u] = j.src[(int64_t)(j.mode == 0 ? j.K - 1 - k : k) * j.n + i];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = i0 + 256 * u;
    if (i >= j.n) continue;
    if (j.mode == 2) { if (v[u] >= 0) j.dst[(int64_t)k * j.n2 + v[u]] = (int)i; }
    else j.dst[(int64_t)k * j.n + i] = v[u];
  }
}

// generated sets straight from the coarsest level (MinkowskiGenerativeConvolutionTranspose k2 s2 applied `depth` times, row 8i + k
// each time — Appendix A.4): row t = ((i * 8 + k_depth-1) * 8 + ...) + k_0 sits at coarse[i] + sum_d bits(k_d) * (stride << d)
struct GenJob { const int4* coarse; int4* out; int64_t n; int depth, stride, nbx; };
__global__ void k_plan_gen_coords(Batch<GenJob> bt) {
  int64_t lb;
  const GenJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const int64_t t = lb * 256 + threadIdx.x;
  if (t >= j.n) return;
  int4 c = j.coarse[t >> (3 * j.depth)];
  for (int d = 0; d < j.depth; ++d) {
    const int k = (int)(t >> (3 * d)) & 7;
    const int h = j.stride << d;
    c.y += (k & 1) ? h : 0; c.z += (k & 2) ? h : 0; c.w += (k & 4) ? h : 0;
  }
  j.out[t] = c;
}

// row of each voxel of a backbone level (stride T) in the generated set `depth` levels below the coarsest level: the generated
// sets hold ALL descendants of the coarsest set (child k of row i at 8i + k), so the row follows from ONE probe of the coarsest
// table and `depth` child-bit triples — identical to fc_child_rows on the parent set's table.  *n_found counts the hits.
struct GenRowsJob { const int4* q; int* rows; int* n_found; int64_t n; int T, depth; };
__global__ void k_plan_gen_rows(Batch<GenRowsJob> bt, const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                unsigned long long mask) {
  int64_t lb;
  const GenRowsJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const int64_t i = lb * 256 + threadIdx.x;
  int hit = 0;
  if (i < j.n) {
    const int4 c = j.q[i];
    const int T = j.T, P = T << j.depth;           // stride of the coarsest set
    const int px = fc_floor_div(c.y, P) * P, py = fc_floor_div(c.z, P) * P, pz = fc_floor_div(c.w, P) * P;
    int r = fc_lookup(keys, vals, mask, fc_pack(c.x, px, py, pz));
    if (r >= 0) {
      const int rx = (c.y - px) / T, ry = (c.z - py) / T, rz = (c.w - pz) / T;       // in [0, 2^depth)
      for (int d = j.depth - 1; d >= 0; --d) r = 8 * r + (((rx >> d) & 1) | (((ry >> d) & 1) << 1) | (((rz >> d) & 1) << 2));
      hit = 1;
    }
    j.rows[i] = r;
  }
  const unsigned long long bal = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(j.n_found, __popcll(bal));
}

// the head's per-location arrays over all levels (finest first): location = voxel corner * voxel size
// (fcaf3d_neck_with_head.py:276-277), scene, level, identity order (rows of every set are grouped by scene)
struct HeadArgs { const int4* coords[MAXLV]; int64_t off[MAXLV + 1]; int nl; float vs; };
__global__ void k_plan_head_arrays(HeadArgs h, float* __restrict__ pts, int* __restrict__ scene, int* __restrict__ level, int* __restrict__ order) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= h.off[h.nl]) return;
  int l = 0;
  while (l + 1 < h.nl && t >= h.off[l + 1]) ++l;
  const int4 c = h.coords[l][t - h.off[l]];
  pts[3 * t] = (float)c.y * h.vs; pts[3 * t + 1] = (float)c.z * h.vs; pts[3 * t + 2] = (float)c.w * h.vs;
  scene[t] = c.x;
  level[t] = l;
  order[t] = (int)t;
}

// occupancy masks of the rows of (27, n) tables (fc_nbr_row_masks), all tables that get a mask-sorted copy in one launch
struct SortJob { const int* tab; int* masks; int* order; int* sorted; int* ka; int* kb; int* va; int* hist; int64_t n; int nblk, nbx; };
__global__ void k_plan_row_masks(Batch<SortJob> bt) {
  int64_t lb;
  const SortJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const int64_t o = lb * 256 + threadIdx.x;
  if (o >= j.n) return;
  unsigned int m = 0;
  for (int k = 0; k < 27; ++k)
    if (j.tab[(int64_t)k * j.n + o] >= 0) m |= 1u << k;
  j.masks[o] = (int)m;
}
// sorted[k][t] = tab[k][order[t]] (fc_permute_nbr); job blocks = 27 * ceil(n / 256)
__global__ void k_plan_permute(Batch<SortJob> bt) {
  int64_t lb;
  const SortJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const int k = (int)(lb / j.nbx);
  const int64_t t = (lb % j.nbx) * 256 + threadIdx.x;
  if (t >= j.n) return;
  j.sorted[(int64_t)k * j.n + t] = j.tab[(int64_t)k * j.n + j.order[t]];      // (four rows per thread: no faster — a gather of 4-byte entries)
}

// ---- LSD radix argsort of 27-bit keys (the occupancy masks), stable: 3 passes of 9 bits; all sort jobs of a plan per launch -------
constexpr int RB = 9, RBINS = 1 << RB, RTILE = 4096;
struct RadixIO { const int* kin; const int* vin; int* kout; int* vout; };
// p0: (masks, identity) -> (ka, order); p1: (ka, order) -> (kb, va); p2: (kb, va) -> (ka, order): the result lands in `order`.
// PASS is a template parameter: with a run-time pass hipcc 7.2 left the key pointer of the third pass undefined in k_radix_scatter
// (the switch lowering assigned it on one of two default paths only: it read job 0's `tab` — found with a null `tab`, r6_notes.md)
template <int PASS> __device__ __forceinline__ RadixIO radix_io(const SortJob& j) {
  if (PASS == 0) return {j.masks, nullptr, j.ka, j.order};
  if (PASS == 1) return {j.ka, j.order, j.kb, j.va};
  return {j.kb, j.va, j.ka, j.order};
}
template <int PASS> __global__ __launch_bounds__(256) void k_radix_hist(Batch<SortJob> bt) {
  constexpr int pass = PASS;
  __shared__ int h[RBINS];
  int64_t lb;
  const SortJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const int* keys = radix_io<PASS>(j).kin;
  const int shift = pass * RB;
  for (int b = threadIdx.x; b < RBINS; b += 256) h[b] = 0;
  __syncthreads();
  const int64_t base = lb * RTILE;
  for (int r = 0; r < RTILE / 256; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < j.n) atomicAdd(&h[(keys[i] >> shift) & (RBINS - 1)], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < RBINS; b += 256) j.hist[lb * RBINS + b] = h[b];
}

// one block of RBINS threads per job: hist[blk][d] -> global start of (d, blk) in (digit, block) order
__global__ __launch_bounds__(RBINS) void k_radix_scan(Batch<SortJob> bt) {
  __shared__ int tot[RBINS];
  const SortJob& j = bt.j[blockIdx.x];
  int* hist = j.hist;
  const int nblk = j.nblk;
  const int d = threadIdx.x;
  int run = 0;
  for (int b = 0; b < nblk; ++b) {
    const int v = hist[(int64_t)b * RBINS + d];
    hist[(int64_t)b * RBINS + d] = run;
    run += v;
  }
  tot[d] = run;
  __syncthreads();
  int x = run;                                     // inclusive scan of the digit totals (Hillis-Steele over RBINS threads)
  for (int o = 1; o < RBINS; o <<= 1) {
    const int t = d >= o ? tot[d - o] : 0;
    __syncthreads();
    x += t;
    tot[d] = x;
    __syncthreads();
  }
  const int dbase = x - run;
  for (int b = 0; b < nblk; ++b) hist[(int64_t)b * RBINS + d] += dbase;
}

// rank inside the tile in row order (stable) and scatter.  vin == nullptr: the identity (first pass).
template <int PASS> __global__ __launch_bounds__(256) void k_radix_scatter(Batch<SortJob> bt) {
  constexpr int pass = PASS;
  __shared__ int offs[RBINS];
  int64_t lb;
  const SortJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const RadixIO io = radix_io<PASS>(j);
  const int shift = pass * RB;
  const int64_t n = j.n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int b = threadIdx.x; b < RBINS; b += 256) offs[b] = j.hist[lb * RBINS + b];
  __syncthreads();
  const int64_t base = lb * RTILE;
  for (int r = 0; r < RTILE / 256; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    const bool live = i < n;
    const int key = live ? io.kin[i] : 0;
    const int d = (key >> shift) & (RBINS - 1);
    unsigned long long peers = __ballot(live);     // lanes of this wave with the same digit
    for (int b = 0; b < RB; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    const int cnt = __popcll(peers);
    int dst = 0;
    for (int ww = 0; ww < 4; ++ww) {               // waves in row order
      if (w == ww && live && rank == 0) { dst = offs[d]; offs[d] = dst + cnt; }
      __syncthreads();
    }
    const int leader = __ffsll((long long)peers) - 1;
    dst = __shfl(dst, live ? leader : 0, 64);
    if (live) {
      io.kout[dst + rank] = key;
      io.vout[dst + rank] = io.vin ? io.vin[i] : (int)i;
    }
  }
}

int64_t argsort27_ws_bytes(int64_t n) { return fc_align(4 * n, 256) * 3 + fc_align(4 * fc_cdiv(n > 0 ? n : 1, RTILE) * RBINS, 256); }
// masks must be filled already (job.masks); ws of every job: ka, kb, va, hist
int argsort27_batch(const Batch<SortJob>& bt, hipStream_t s) {
  static_assert(sizeof(Batch<SortJob>) <= 1536, "batch descriptor too large for the kernel-argument block");
  if (bt.overflow) return FC_EINVAL;
  if (!bt.count || !bt.blocks()) return 0;
  Batch<SortJob> tiles = bt;                       // blocks = RTILE tiles
  tiles.count = 0; tiles.blk0[0] = 0;
  for (int i = 0; i < bt.count; ++i) tiles.add(bt.j[i], bt.j[i].nblk);
  const unsigned g = (unsigned)tiles.blocks(), nj = (unsigned)tiles.count;
#define FC_RADIX_PASS(P)                                  \
  k_radix_hist<P><<<g, 256, 0, s>>>(tiles);               \
  FC_CHECK_LAUNCH();                                      \
  k_radix_scan<<<nj, RBINS, 0, s>>>(tiles);               \
  FC_CHECK_LAUNCH();                                      \
  k_radix_scatter<P><<<g, 256, 0, s>>>(tiles);            \
  FC_CHECK_LAUNCH();
  FC_RADIX_PASS(0)
  FC_RADIX_PASS(1)
  FC_RADIX_PASS(2)
#undef FC_RADIX_PASS
  return 0;
}
void sort_job_ws(SortJob& j, void* ws) {
  char* w = (char*)ws;
  j.ka = (int*)w; w += fc_align(4 * j.n, 256);
  j.kb = (int*)w; w += fc_align(4 * j.n, 256);
  j.va = (int*)w; w += fc_align(4 * j.n, 256);
  j.hist = (int*)w;
  j.nblk = (int)fc_cdiv(j.n > 0 ? j.n : 1, RTILE);
  j.nbx = (int)fc_cdiv(j.n > 0 ? j.n : 1, 256);
}

// ---- exact pair lists (fc_kernel_map_pairs) of many (27, n) tables per launch: count pass + fill pass ------------------------------
constexpr int PBLK = 1024;
struct PairJob { const int* tab; int* pi; int* po; int* pos; int* cnt; int* blk_cnt; int64_t n; int nblk; };
__global__ __launch_bounds__(PBLK) void k_plan_pairs_count(Batch<PairJob> bt) {
  int64_t lb;
  const PairJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const int k = (int)(lb / j.nblk), blk = (int)(lb % j.nblk);
  const int64_t row = (int64_t)blk * PBLK + threadIdx.x;
  const int present = row < j.n && j.tab[(int64_t)k * j.n + row] >= 0;
  const int c = __syncthreads_count(present);
  if (threadIdx.x == 0) j.blk_cnt[k * j.nblk + blk] = c;
}
__global__ __launch_bounds__(PBLK) void k_plan_pairs_fill(Batch<PairJob> bt) {
  __shared__ int wave_cnt[PBLK / 64];
  __shared__ int base_s;
  int64_t lb;
  const PairJob& j = bt.j[batch_find(bt, blockIdx.x, &lb)];
  const int k = (int)(lb / j.nblk), blk = (int)(lb % j.nblk), nblk = j.nblk;
  const int64_t n = j.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {                                   // pairs of offset k in the blocks before this one
    int a = 0;
    for (int b = lane; b < blk; b += 64) a += j.blk_cnt[k * nblk + b];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) base_s = a;
  }
  const int64_t row = (int64_t)blk * PBLK + tid;
  const int v = row < n ? j.tab[(int64_t)k * n + row] : -1;
  const unsigned long long bal = __ballot(v >= 0);
  if (lane == 0) wave_cnt[wave] = __popcll(bal);
  __syncthreads();
  int pre = base_s;
  for (int w = 0; w < wave; ++w) pre += wave_cnt[w];
  const int p = pre + __popcll(bal & ((1ull << lane) - 1ull));
  if (v >= 0) {
    j.pi[(int64_t)k * n + p] = v;
    j.po[(int64_t)k * n + p] = (int)row;
  }
  if (row < n) j.pos[(int64_t)k * n + row] = v >= 0 ? p : -1;
  if (blk == nblk - 1 && tid == PBLK - 1) j.cnt[k] = pre + __popcll(bal);
}

int live_tiles(const int* cnt, int K) {
  int64_t t = 0;
  for (int k = 0; k < K; ++k) t += (cnt[k] + 127) / 128;
  return (int)t;
}

template <typename J, typename F> int launch_batch(const Batch<J>& bt, int threads, hipStream_t s, F kernel) {
  static_assert(sizeof(Batch<J>) <= 1536, "batch descriptor too large for the kernel-argument block");
  if (bt.overflow) return FC_EINVAL;
  if (!bt.count || !bt.blocks()) return 0;
  kernel<<<(unsigned)bt.blocks(), threads, 0, s>>>(bt);
  FC_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" {

int fc_abi_version(void) { return FC_ABI_VERSION; }
int fc_plan_cfg_words(void) { return CFGW; }
int fc_plan_out_words(int B, int nl) { return HDR + SETW * MAXSETS + MAPR * (2 + 4 * nl) + MAXSETS * (B > 0 ? B : 1) + 64; }

// stage 1 arena: raw collate, scratch of the chain, the sets of [cm0, m1, m2, L1..Lnl] with their tables, the meta block
static int64_t stage1_layout(int64_t T, int B, int nl, int nfeat, char* base, void** p /*pointers out*/, int S) {
  Bump a{base, 0};
  int i = 0;
  p[i++] = a.arr<int>(4 * T);                                   // 0 coords_raw
  p[i++] = a.arr<float>((int64_t)nfeat * T);                    // 1 feats_raw
  p[i++] = a.arr<int>(T);                                       // 2 slot (even sets)
  p[i++] = a.take(T);                                           // 3 flags
  p[i++] = a.arr<int>(5 * fc_cdiv(T > 0 ? T : 1, 1024) + 8);    // 4 blocksums (one per 256 rows), then one per 1 024 rows
  p[i++] = a.arr<int>((int64_t)METAW * S + (int64_t)S * B + 64); // 5 meta
  p[i++] = a.arr<float>((int64_t)nfeat * T);                    // 6 F0
  p[i++] = a.arr<int>(T);                                       // 7 slot (odd sets)
  const int64_t cap = next_pow2(T > 0 ? 2 * T : 2);
  p[i++] = a.arr<unsigned long long>(cap * S);                  // 8 keys of all sets (S, cap)
  p[i++] = a.arr<int>(cap * S);                                 // 9 vals of all sets (S, cap)
  for (int s = 0; s < S; ++s) p[i++] = a.arr<int>(4 * T);       // 10 + s: coords of set s
  return fc_align(a.off, 256);
}

int64_t fc_plan_stage1_bytes(int64_t total_points, int B, int nl, int nfeat) {
  if (total_points < 0 || B < 1 || nl < 1 || nl > MAXLV || nfeat < 0) return -1;
  void* p[10 + MAXSETS];
  return stage1_layout(total_points, B, nl, nfeat, nullptr, p, 3 + nl);
}

int fc_plan_levels(const int64_t* cfg, const int64_t* scenes, void* arena1, int64_t arena1_bytes, int64_t* out, int* counts_host,
                   hipStream_t stream) {
  const int B = (int)cfg[C_B], nl = (int)cfg[C_NL], nfeat = (int)cfg[C_NFEAT];
  const int64_t T = cfg[C_TOTAL];
  if (B < 1 || B > 32767 || nl < 1 || nl > MAXLV || nfeat < 0 || T < 0 || T > 0x7fffffffLL / 8 || !arena1 || !out || !counts_host) return FC_EINVAL;
  const int S = 3 + nl;
  void* p[10 + MAXSETS];
  if (arena1_bytes < stage1_layout(T, B, nl, nfeat, (char*)arena1, p, S)) return FC_EWS;
  int4* coords_raw = (int4*)p[0];
  float* feats_raw = (float*)p[1];
  int* slots[2] = {(int*)p[2], (int*)p[7]};
  unsigned char* flags = (unsigned char*)p[3];
  int* blocksums = (int*)p[4];
  int* coarse = blocksums + 4 * fc_cdiv(T > 0 ? T : 1, 1024) + 4;
  int* meta = (int*)p[5];
  float* F0 = (float*)p[6];
  const int64_t cap0 = next_pow2(T > 0 ? 2 * T : 2);
  unsigned long long* keys_all = (unsigned long long*)p[8];
  int* vals_all = (int*)p[9];
  const int64_t nmeta = (int64_t)METAW * S + (int64_t)S * B;
  FC_HIP(hipMemsetAsync(meta, 0, (size_t)nmeta * sizeof(int), stream));
  const bool probe = cfg[C_PROBE] != 0;
  size_t stage1_pp[MAXSETS] = {};
  const float vs = (float)as_double(cfg[C_VS]), fdiv = (float)as_double(cfg[C_FEATDIV]);
  if (!cfg[C_COORDS_IN] && !(vs > 0.f)) return FC_EINVAL;
  if (T > 0) {
    {
      Bracket br(probe, PK_TABLES, 12.0 * (double)(cap0 * S), stream);
      k_plan_tables_init<<<(unsigned)fc_cdiv(cap0 * S, 512), 256, 0, stream>>>(keys_all, vals_all, cap0 * S);
      FC_CHECK_LAUNCH();
    }
    // ---- level 0: the collate + insert ----
    const int4* src0 = coords_raw;
    if (!cfg[C_COORDS_IN]) {
      int64_t off = 0;
      for (int b0 = 0; b0 < B; b0 += CH) {
        SceneArgs sc;
        memset(&sc, 0, sizeof(sc));
        int maxn = 0;
        const int nb = B - b0 < CH ? B - b0 : CH;
        for (int j = 0; j < nb; ++j) {
          sc.p[j] = (const float*)scenes[3 * (b0 + j)];
          sc.n[j] = (int)scenes[3 * (b0 + j) + 1];
          if (sc.n[j] < 0 || (sc.n[j] > 0 && !sc.p[j]) || scenes[3 * (b0 + j) + 2] != cfg[C_PT_STRIDE]) return FC_EINVAL;
          sc.off[j] = (int)off;
          off += sc.n[j];
          if (sc.n[j] > maxn) maxn = sc.n[j];
        }
        if (off > T) return FC_EINVAL;
        sc.b0 = b0; sc.stride = (int)cfg[C_PT_STRIDE]; sc.nfeat = nfeat; sc.vs = vs; sc.feat_div = fdiv;
        if (sc.stride < 3 + nfeat) return FC_EINVAL;
        if (maxn > 0) {
          // points read, coords + features + slot written, one 12-byte table slot read and written per point
          Bracket br(probe, PK_VOXELIZE, (double)(off - sc.off[0]) * (4.0 * sc.stride + 16.0 + 4.0 * nfeat + 4.0 + 24.0), stream);
          dim3 grid((unsigned)fc_cdiv(maxn, 256), nb);
          k_plan_voxelize_insert<<<grid, 256, 0, stream>>>(sc, coords_raw, feats_raw, keys_all, vals_all, table_mask(T), meta, slots[0]);
          FC_CHECK_LAUNCH();
        }
      }
      if (off != T) return FC_EINVAL;
    } else {                                       // pre-voxelised input (an augmenting pipeline wrote coords / feats itself)
      src0 = (const int4*)cfg[C_COORDS_IN];
      feats_raw = (float*)cfg[C_FEATS_IN];
      if (!feats_raw && nfeat) return FC_EINVAL;
      Bracket br(probe, PK_VOXELIZE, (double)T * (16.0 + 4.0 + 24.0), stream);
      k_plan_insert<<<(unsigned)fc_cdiv(T, 256), 256, 0, stream>>>(src0, T, keys_all, vals_all, table_mask(T), meta, slots[0]);
      FC_CHECK_LAUNCH();
    }
    // ---- the chain: two launches per set ----
    const unsigned gB = (unsigned)fc_cdiv(T, 1024), gF = (unsigned)fc_cdiv(T, 256);
    for (int s = 0; s < S; ++s) {
      int* meta_s = meta + METAW * s;
      const int* n_in_dev = s ? meta + METAW * (s - 1) : nullptr;
      const int4* src = s ? (const int4*)p[10 + s - 1] : src0;
      int* vals = vals_all + cap0 * s;
      const size_t pp0 = g_pp.size();
      {
        Bracket br(probe, PK_FLAGS, 0.0, stream);        // bytes are filled in after the read-back (they depend on the live counts)
        k_plan_flags<<<gB, 256, 0, stream>>>(slots[s & 1], vals, n_in_dev, T, flags, blocksums, coarse);
        FC_CHECK_LAUNCH();
      }
      if (probe) stage1_pp[s] = pp0;
      const bool more = s + 1 < S;
      Bracket br(probe, PK_FINALIZE, 0.0, stream);
      k_plan_finalize<<<gF, 256, 0, stream>>>(src, n_in_dev, T, 1 << s, flags, blocksums, coarse, slots[s & 1], vals, (int4*)p[10 + s],
                                             s == 0 ? feats_raw : nullptr, s == 0 ? F0 : nullptr, nfeat,
                                             meta + METAW * S + (int64_t)s * B, B, meta_s, more ? keys_all + cap0 * (s + 1) : nullptr,
                                             more ? vals_all + cap0 * (s + 1) : nullptr, slots[(s + 1) & 1]);
      FC_CHECK_LAUNCH();
    }
  }
  int rc = sync_readback(counts_host, meta, nmeta * (int64_t)sizeof(int), stream);
  if (rc) return rc;
  if (probe && T > 0)
    for (int s = 0; s < S; ++s) {                   // compulsory bytes of the chain's launches, now that the counts are known
      const double n_in = s ? counts_host[METAW * (s - 1)] : (double)T, n_out = counts_host[METAW * s];
      g_pp[stage1_pp[s]].bytes = n_in * (4.0 + 4.0 + 1.0);                                     // slot, table value, flag
      g_pp[stage1_pp[s] + 1].bytes = n_in * (1.0 + 4.0) + n_out * (16.0 + 16.0 + 4.0 + (s == 0 ? 8.0 * nfeat : 0.0)) +
                                     (s + 1 < S ? n_out * (24.0 + 4.0) : 0.0);                 // ... + the next set's insert
    }
  // ---- host tables ----
  const int nw = fc_plan_out_words(B, nl);
  memset(out, 0, (size_t)nw * sizeof(int64_t));
  out[H_S] = S;
  out[H_F0] = (int64_t)F0;
  out[H_PRUNE] = -1;
  int bad = 0;
  for (int s = 0; s < S; ++s) {
    const int* m = counts_host + METAW * s;
    int64_t* o = out + HDR + SETW * s;
    o[S_COORDS] = (int64_t)p[10 + s];
    o[S_N] = m[0];
    o[S_STRIDE] = 1 << s;
    o[S_KEYS] = (int64_t)(keys_all + cap0 * s);
    o[S_VALS] = (int64_t)(vals_all + cap0 * s);
    o[S_CAP] = (int64_t)table_mask(s ? counts_host[METAW * (s - 1)] : T) + 1;
    o[S_PARENT] = -1;
    bad |= m[2];
  }
  out[H_BAD] = bad;
  return FC_OK;
}

// ---- stage 2 -------------------------------------------------------------------------------------------------------------------
struct MapRec { int in, out, K; int64_t n_in, n_out; bool dense, self; };

// Everything stage 2 builds, as one pass over a bump allocator: base == nullptr sizes the arena, otherwise the kernels are enqueued
// — one (or a few) launches per KIND of job.
static int stage2(const int64_t* cfg, int64_t* out, const int* counts_host, char* base, int64_t* need, int* cnt_host, hipStream_t stream) {
  const int B = (int)cfg[C_B], nl = (int)cfg[C_NL];
  const bool backward = cfg[C_BACKWARD] != 0, neck = cfg[C_NECK] != 0, targets = cfg[C_TARGETS] != 0;
  const int64_t sort_min = cfg[C_SORT_MIN], pair_rows = cfg[C_PAIR_ROWS], pts_thr = cfg[C_PTS_THR];
  const int S0 = 3 + nl;
  const bool run = base != nullptr;
  const bool probe = run && cfg[C_PROBE] != 0;
  Bump a{base, 0};
  int64_t* sets = out + HDR;
  auto SN = [&](int s) { return sets[SETW * s + S_N]; };
  auto SC = [&](int s) { return (const int4*)sets[SETW * s + S_COORDS]; };
  const int cs = S0 - 1;                            // the coarsest backbone level: every generated row follows from it
  // ---- neck sets: g_i = children of the head level above it, i = nl-2 .. 0 (set index S0 + (nl-2-i)) ----
  int S = S0;
  int head_set[MAXLV];                              // head level l (finest first) -> set index
  int prune = -1;
  Batch<GenJob> gen;
  if (neck) {
    int x = cs;
    head_set[nl - 1] = x;
    for (int i = nl - 2; i >= 0; --i) {
      const int g = S++;
      const int depth = nl - 1 - i;
      int64_t* o = sets + SETW * g;
      o[S_N] = 8 * SN(x);
      o[S_STRIDE] = sets[SETW * x + S_STRIDE] / 2;
      o[S_PARENT] = x;
      o[S_KEYS] = o[S_VALS] = o[S_CAP] = 0;
      o[S_COORDS] = (int64_t)a.arr<int>(4 * o[S_N]);
      o[S_ROWS] = (int64_t)a.arr<int>(SN(3 + i));
      head_set[i] = g;
      gen.add({SC(cs), (int4*)o[S_COORDS], o[S_N], depth, (int)o[S_STRIDE], 0}, fc_cdiv(o[S_N], 256));
      // pts_threshold (fcaf3d_neck_with_head.py:110-126): rows of a scene in g = its rows in the coarsest level * 8^depth
      if (pts_thr >= 0 && prune < 0) {
        const int* sc_cnt = counts_host + METAW * S0 + (int64_t)cs * B;
        for (int b = 0; b < B; ++b)
          if (((int64_t)sc_cnt[b] << (3 * depth)) > pts_thr) prune = i;
      }
      x = g;
      if (prune >= 0) break;                        // the sets below a pruned level depend on the network's scores
    }
  }
  out[H_S] = S;
  out[H_PRUNE] = prune;
  out[H_NHEAD] = neck ? nl : 0;
  // ---- maps ----
  MapRec mr[2 + 4 * MAXLV];
  int nm = 0;
  mr[nm++] = {0, 1, 27, SN(0), SN(1), false, false};           // stem conv k3 s2 (its own kernel: table only)
  mr[nm++] = {1, 2, 8, SN(1), SN(2), false, false};            // max-pool k2 s2
  for (int li = 1; li <= nl; ++li) {
    const int prev = 1 + li, mi = 2 + li;
    mr[nm++] = {prev, mi, 27, SN(prev), SN(mi), false, false}; // down  k3 s2
    mr[nm++] = {prev, mi, 1, SN(prev), SN(mi), false, false};  // ds    k1 s2 (row 13 of `down`)
    mr[nm++] = {mi, mi, 27, SN(mi), SN(mi), false, true};      // same  k3 s1
  }
  for (int g = S0; g < S; ++g) mr[nm++] = {g, g, 27, SN(g), SN(g), true, true};   // gsame (nl-2 .. ) k3 s1 on a generated set
  out[H_NMAPS] = nm;
  int64_t* maps = out + HDR + SETW * MAXSETS;
  // the pair-list counters of all maps sit in ONE block (read back together): 2 x 27 ints per map + union hit counters
  int* cnt_dev = a.arr<int>(64 * nm + MAXLV);

  Batch<KMapJob> kmaps;
  Batch<RowJob> fills, rows, prefills;              // fills (-1) run before the scatters; prefills: the tables of the strided maps, before k_plan_strided_maps
  Batch<SMapJob> smaps;
  Batch<SortJob> sorts;
  Batch<PairJob> pairs;
  struct Child { const int* pnbr; int64_t n_parent; int* nbr; } children[MAXLV];
  int nchildren = 0;

  // -- pass A: the tables themselves --
  for (int m = 0; m < nm; ++m) {
    const MapRec& r = mr[m];
    int64_t* o = maps + MAPR * m;
    o[MW_IN] = r.in; o[MW_OUT] = r.out; o[MW_K] = r.K; o[MW_NIN] = r.n_in; o[MW_NOUT] = r.n_out;
    const int64_t* din = sets + SETW * r.in;
    const bool is_ds = r.K == 1;
    const bool conv = m >= 2;
    int* nbr = nullptr;
    int* nbr_t = nullptr;
    if (is_ds) {                                    // k1 s2: the centre row of the k3 s2 table of the same pair of sets
      const int64_t* od = maps + MAPR * (m - 1);
      if (run) {
        nbr = (int*)od[MW_NBR] + 13 * r.n_out;
        nbr_t = backward ? (int*)od[MW_NBRT] + 13 * r.n_in : nullptr;
      }
    } else {
      nbr = a.arr<int>((int64_t)r.K * r.n_out);
      if (r.dense) {                                // generated set: from the parent's own k3 table (index arithmetic)
        const int par = (int)din[S_PARENT];
        const int64_t* op = nullptr;                // the parent's k3 s1 map: `same` of the coarsest level, or the gsame above
        for (int mm = 0; mm < m; ++mm)
          if (mr[mm].self && mr[mm].in == par) op = maps + MAPR * mm;
        if (!op) return FC_EINVAL;
        children[nchildren++] = {(const int*)op[MW_NBR], SN(par), nbr};
      } else {
        const int ks = r.K == 27 ? 3 : 2;
        const int nbx = (int)fc_cdiv(r.n_out, 256);
        const int64_t* dout = sets + SETW * r.out;
        if (r.in != r.out && dout[S_STRIDE] == 2 * din[S_STRIDE]) {      // stride s -> 2 s: from the input side (k_plan_strided_maps)
          smaps.add({SC(r.in), (const unsigned long long*)dout[S_KEYS], (const int*)dout[S_VALS], (unsigned long long)(dout[S_CAP] - 1), nbr,
                     r.n_in, r.n_out, ks, (int)din[S_STRIDE], (int)fc_cdiv(8 * r.n_in, 256)}, fc_cdiv(8 * r.n_in, 256));
          prefills.add({nullptr, nbr, r.n_out, r.n_out, r.K, 3, nbx}, fc_cdiv((int64_t)r.K * r.n_out, 1024));
        } else {
          kmaps.add({SC(r.out), (const unsigned long long*)din[S_KEYS], (const int*)din[S_VALS], (unsigned long long)(din[S_CAP] - 1), nbr,
                     r.n_out, r.K, ks, (int)din[S_STRIDE], nbx}, (int64_t)r.K * nbx);
        }
      }
      if (conv && backward) {
        nbr_t = a.arr<int>((int64_t)r.K * r.n_in);
        const int nbx = (int)fc_cdiv(r.n_in, 256);
        if (r.self) {
          rows.add({nbr, nbr_t, r.n_in, r.n_in, r.K, 0, nbx}, (int64_t)r.K * ((nbx + 3) / 4));
        } else {
          fills.add({nullptr, nbr_t, r.n_in, r.n_in, r.K, 3, nbx}, fc_cdiv((int64_t)r.K * r.n_in, 1024));
          const int nbo = (int)fc_cdiv(r.n_out, 256);
          rows.add({nbr, nbr_t, r.n_out, r.n_in, r.K, 2, nbo}, (int64_t)r.K * ((nbo + 3) / 4));
        }
      }
    }
    o[MW_NBR] = (int64_t)nbr;
    o[MW_NBRT] = (int64_t)nbr_t;
    int64_t* df = o + MW_DESC_F;
    int64_t* db = o + MW_DESC_B;
    df[0] = db[0] = r.n_in; df[1] = db[1] = r.n_out; df[2] = db[2] = r.K; df[3] = db[3] = (int64_t)nbr;
    const bool sort_rows = r.K == 27 && !r.dense && r.n_out >= sort_min;
    const bool use_pairs = r.K == 27 && !r.dense;
    o[MW_FLAGS] = (sort_rows ? 1 : 0) | (use_pairs ? 2 : 0) | (r.dense ? 4 : 0);
  }
  // -- pass B: the derived tables of every convolution route --
  for (int m = 2; m < nm; ++m) {
    const MapRec& r = mr[m];
    int64_t* o = maps + MAPR * m;
    int* nbr = (int*)o[MW_NBR];
    int* nbr_t = (int*)o[MW_NBRT];
    const bool sort_rows = (o[MW_FLAGS] & 1) != 0, use_pairs = (o[MW_FLAGS] & 2) != 0;
    auto pair_lists = [&](const int* tab, int64_t n, int base_word, int* cnt) {
      PairJob j;
      j.tab = tab; j.n = n; j.cnt = cnt;
      j.pi = a.arr<int>(27 * n); j.po = a.arr<int>(27 * n); j.pos = a.arr<int>(27 * n);
      j.nblk = (int)fc_cdiv(n > 0 ? n : 1, PBLK);
      j.blk_cnt = a.arr<int>(27 * (int64_t)j.nblk);
      o[base_word] = (int64_t)j.pi; o[base_word + 1] = (int64_t)j.po; o[base_word + 2] = (int64_t)j.pos; o[base_word + 3] = (int64_t)cnt;
      pairs.add(j, n > 0 ? 27 * (int64_t)j.nblk : 0);
    };
    auto sorted = [&](const int* tab, int64_t n, int w_tab, int w_idx) {
      SortJob j;
      memset(&j, 0, sizeof(j));
      j.tab = tab; j.n = n;
      j.order = a.arr<int>(n);
      j.sorted = a.arr<int>(27 * n);
      j.masks = a.arr<int>(n);
      sort_job_ws(j, a.take(argsort27_ws_bytes(n)));
      o[w_tab] = (int64_t)j.sorted; o[w_idx] = (int64_t)j.order;
      sorts.add(j, j.nbx);
    };
    int* cnt_f = cnt_dev + 64 * m;
    int* cnt_t = cnt_dev + 64 * m + 32;
    bool have_pairs = false;
    if (use_pairs && r.n_out <= pair_rows) {
      pair_lists(nbr, r.n_out, MW_PI, cnt_f);
      have_pairs = true;
    } else if (sort_rows) {
      sorted(nbr, r.n_out, MW_SORT, MW_SORTI);
    } else {
      o[MW_SORT] = (int64_t)nbr; o[MW_SORTI] = 0;
    }
    if (backward) {
      if (use_pairs && !have_pairs) pair_lists(nbr, r.n_out, MW_PI, cnt_f);      // the weight gradient reduces over exact pair lists
      if (use_pairs && r.n_in <= pair_rows) {
        pair_lists(nbr_t, r.n_in, MW_TPI, cnt_t);
      } else if (sort_rows) {
        sorted(nbr_t, r.n_in, MW_SORTT, MW_SORTTI);
      } else {
        o[MW_SORTT] = (int64_t)nbr_t; o[MW_SORTTI] = 0;
      }
    }
  }
  // ---- union rows of the backbone levels inside the generated sets (fcaf3d_neck_with_head.py:101) ----
  Batch<GenRowsJob> grows;
  for (int g = S0; g < S; ++g) {
    const int i = nl - 2 - (g - S0);                // backbone level index (0-based): set 3 + i
    grows.add({SC(3 + i), (int*)sets[SETW * g + S_ROWS], cnt_dev + 64 * nm + (g - S0), SN(3 + i), (int)sets[SETW * (3 + i) + S_STRIDE],
               nl - 1 - i}, fc_cdiv(SN(3 + i), 256));
  }
  // ---- the head's location arrays (training: what the target assignment and the loss read) ----
  int64_t n_all = 0;
  if (neck && prune < 0)
    for (int l = 0; l < nl; ++l) n_all += SN(head_set[l]);
  out[H_NALL] = n_all;
  float* pts = nullptr; int* scene = nullptr; int* level = nullptr; int* order = nullptr; int* seg = nullptr;
  const bool want_head = targets && neck && prune < 0;
  if (want_head) {
    pts = a.arr<float>(3 * n_all); scene = a.arr<int>(n_all); level = a.arr<int>(n_all); order = a.arr<int>(n_all);
    seg = a.arr<int>((int64_t)nl * B + 1);
    out[H_TGT_PTS] = (int64_t)pts; out[H_TGT_SCENE] = (int64_t)scene; out[H_TGT_LEVEL] = (int64_t)level;
    out[H_TGT_ORDER] = (int64_t)order; out[H_TGT_SEG] = (int64_t)seg;
  }
  *need = fc_align(a.off, 256);
  if (!run) return 0;

  // ---- launches, in dependency order ----
  FC_HIP(hipMemsetAsync(cnt_dev, 0, sizeof(int) * (64 * nm + MAXLV), stream));
  int rc;
  auto rows_bytes = [](const Batch<RowJob>& b) { double t = 0; for (int i = 0; i < b.count; ++i) t += (double)b.j[i].K * b.j[i].n * ((b.j[i].mode == 1 || b.j[i].mode == 3) ? 4.0 : 8.0); return t; };
  double by = 0;
  for (int i = 0; i < gen.count; ++i) by += 16.0 * gen.j[i].n;
  { Bracket br(probe, PK_GEN, by, stream); if ((rc = launch_batch(gen, 256, stream, k_plan_gen_coords))) return rc; }
  by = 0;
  for (int i = 0; i < kmaps.count; ++i) by += (double)kmaps.j[i].n_out * (16.0 + kmaps.j[i].K * (4.0 + 12.0));      // coords, table entry written, slot probed
  for (int i = 0; i < smaps.count; ++i)             // strided maps: input coords, <= 8 probes of 12 B per input row, the table filled and hit
    by += (double)smaps.j[i].n_in * (16.0 + 3.4 * 12.0 + 3.4 * 4.0) + (double)smaps.j[i].n_out * smaps.j[i].ks * smaps.j[i].ks * smaps.j[i].ks * 4.0;
  {
    Bracket br(probe, PK_KMAPS, by, stream);
    if ((rc = launch_batch(prefills, 256, stream, k_plan_rows))) return rc;
    if ((rc = launch_batch(smaps, 256, stream, k_plan_strided_maps))) return rc;
    if ((rc = launch_batch(kmaps, 256, stream, k_plan_kernel_maps))) return rc;
  }
  by = 0;
  for (int c = 0; c < nchildren; ++c) by += 27.0 * 4.0 * (children[c].n_parent + 8.0 * children[c].n_parent);
  {
    Bracket br(probe, PK_CHILDREN, by, stream);
    for (int c = 0; c < nchildren; ++c)            // a chain: each generated set's table from the one above it
      if ((rc = fc_kernel_map_children(children[c].pnbr, children[c].n_parent, children[c].nbr, stream))) return rc;
  }
  { Bracket br(probe, PK_FILL, rows_bytes(fills), stream); if ((rc = launch_batch(fills, 256, stream, k_plan_rows))) return rc; }
  { Bracket br(probe, PK_TRANSPOSE, rows_bytes(rows), stream); if ((rc = launch_batch(rows, 256, stream, k_plan_rows))) return rc; }
  double nsort = 0;
  for (int i = 0; i < sorts.count; ++i) nsort += (double)sorts.j[i].n;
  { Bracket br(probe, PK_MASKS, nsort * (27.0 * 4.0 + 4.0), stream); if ((rc = launch_batch(sorts, 256, stream, k_plan_row_masks))) return rc; }
  { Bracket br(probe, PK_RADIX, nsort * 3.0 * (8.0 + 4.0 + 8.0), stream); if ((rc = argsort27_batch(sorts, stream))) return rc; }      // per pass: keys + values read twice-ish, written once
  {
    Batch<SortJob> perm;
    for (int i = 0; i < sorts.count; ++i) perm.add(sorts.j[i], 27 * (int64_t)sorts.j[i].nbx);
    Bracket br(probe, PK_PERMUTE, nsort * (27.0 * 8.0 + 4.0), stream);
    if ((rc = launch_batch(perm, 256, stream, k_plan_permute))) return rc;
  }
  by = 0;
  for (int i = 0; i < pairs.count; ++i) by += 27.0 * (double)pairs.j[i].n * (4.0 + 4.0 + 12.0);      // table read twice, lists + positions written
  {
    Bracket br(probe, PK_PAIRS, by, stream);
    if ((rc = launch_batch(pairs, PBLK, stream, k_plan_pairs_count))) return rc;
    if ((rc = launch_batch(pairs, PBLK, stream, k_plan_pairs_fill))) return rc;
  }
  if (grows.count && grows.blocks()) {
    const int64_t* oc = sets + SETW * cs;
    by = 0;
    for (int i = 0; i < grows.count; ++i) by += (double)grows.j[i].n * (16.0 + 4.0 + 12.0);
    Bracket br(probe, PK_GENROWS, by, stream);
    k_plan_gen_rows<<<(unsigned)grows.blocks(), 256, 0, stream>>>(grows, (const unsigned long long*)oc[S_KEYS], (const int*)oc[S_VALS],
                                                                 (unsigned long long)(oc[S_CAP] - 1));
    FC_CHECK_LAUNCH();
  }
  if (want_head) {
    HeadArgs h;
    memset(&h, 0, sizeof(h));
    h.nl = nl; h.vs = (float)as_double(cfg[C_VS_HEAD]);
    int64_t off = 0;
    for (int l = 0; l < nl; ++l) { h.coords[l] = SC(head_set[l]); h.off[l] = off; off += SN(head_set[l]); }
    h.off[nl] = off;
    if (n_all > 0) {
      Bracket br(probe, PK_HEAD, (double)n_all * (16.0 + 12.0 + 12.0), stream);
      k_plan_head_arrays<<<(unsigned)fc_cdiv(n_all, 256), 256, 0, stream>>>(h, pts, scene, level, order);
      FC_CHECK_LAUNCH();
    }
    // (level, scene) segment starts: rows per scene of a generated set = 8^depth x those of the coarsest level — host arithmetic,
    // staged in the caller's pinned counter block (it stays untouched until the read-back below)
    int* stage = cnt_host + 64 * nm + MAXLV;
    const int* sc_cnt = counts_host + METAW * S0 + (int64_t)cs * B;
    int64_t run_ = 0;
    for (int l = 0; l < nl; ++l)
      for (int b = 0; b < B; ++b) {
        stage[l * B + b] = (int)run_;
        run_ += (int64_t)sc_cnt[b] << (3 * (nl - 1 - l));
      }
    stage[nl * B] = (int)run_;
    FC_HIP(hipMemcpyAsync(seg, stage, sizeof(int) * ((size_t)nl * B + 1), hipMemcpyHostToDevice, stream));
  }
  // ---- ONE read-back: pair-list counts of every map, union hit counters ----
  rc = sync_readback(cnt_host, cnt_dev, sizeof(int) * (64 * nm + MAXLV), stream);
  if (rc) return rc;
  int structured = 1;
  for (int g = S0; g < S; ++g) {
    const int i = nl - 2 - (g - S0);
    if (cnt_host[64 * nm + (g - S0)] != SN(3 + i)) structured = 0;      // a backbone voxel outside the generated set: per-operator path
  }
  out[H_STRUCT] = structured;
  for (int m = 2; m < nm; ++m) {
    int64_t* o = maps + MAPR * m;
    int64_t* df = o + MW_DESC_F;
    int64_t* db = o + MW_DESC_B;
    const MapRec& r = mr[m];
    const bool use_pairs = (o[MW_FLAGS] & 2) != 0;
    int64_t ff = 0, fb = 0;
    if (o[MW_PI] && use_pairs && r.n_out <= pair_rows) {
      o[MW_TILES] = live_tiles(cnt_host + 64 * m, 27);
      for (int w = 0; w < 4; ++w) df[9 + w] = db[9 + w] = o[MW_PI + w];
      df[13] = db[13] = o[MW_TILES];
      ff |= 1; fb |= 1;
    } else {
      df[5] = db[5] = o[MW_SORT]; df[6] = db[6] = o[MW_SORTI];
    }
    if (backward) {
      db[4] = o[MW_NBRT];
      if (use_pairs) {
        for (int w = 0; w < 4; ++w) db[9 + w] = o[MW_PI + w];
        fb |= 4;
      }
      if (use_pairs && r.n_in <= pair_rows) {
        o[MW_TTILES] = live_tiles(cnt_host + 64 * m + 32, 27);
        for (int w = 0; w < 4; ++w) db[14 + w] = o[MW_TPI + w];
        db[18] = o[MW_TTILES];
        fb |= 2;
      } else {
        db[7] = o[MW_SORTT]; db[8] = o[MW_SORTTI];
      }
    }
    df[19] = ff; db[19] = fb;
  }
  return FC_OK;
}

int64_t fc_plan_stage2_bytes(const int64_t* cfg, int64_t* out, const int* counts_host) {
  int64_t need = 0;
  int rc = stage2(cfg, out, counts_host, nullptr, &need, nullptr, nullptr);
  return rc ? -1 : need + 256;
}

int fc_plan_maps(const int64_t* cfg, int64_t* out, const int* counts_host, void* arena2, int64_t arena2_bytes, int* cnt_host,
                 hipStream_t stream) {
  if (!cfg || !out || !counts_host || !arena2 || !cnt_host) return FC_EINVAL;
  int64_t need = 0;
  int rc = stage2(cfg, out, counts_host, nullptr, &need, nullptr, nullptr);
  if (rc) return rc;
  char* base = (char*)fc_align((int64_t)arena2, 256);
  if (arena2_bytes - (base - (char*)arena2) < need) return FC_EWS;
  return stage2(cfg, out, counts_host, base, &need, cnt_host, stream);
}

// brackets of the plans run with cfg[C_PROBE] since the last read-out: ms (float), bytes (double), kind (int) per bracket; -> count
int64_t fc_plan_probe_read(float* ms, double* bytes, int* kind, int64_t cap) {
  int64_t n = 0;
  for (PlanProbe& p : g_pp) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, p.a, p.b) != hipSuccess) t = 0.f;
    if (n < cap) { ms[n] = t; bytes[n] = p.bytes; kind[n] = p.kind; ++n; }
    g_pp_pool.push_back(p.a);
    g_pp_pool.push_back(p.b);
  }
  g_pp.clear();
  return n;
}

// stable argsort of non-negative int32 keys below 2^27 (the occupancy masks of fc_nbr_row_masks) — what torch.argsort did for
// the mask-sorted tables of the per-operator path
int64_t fc_argsort27_ws_bytes(int64_t n) { return argsort27_ws_bytes(n); }
int fc_argsort27(const int* keys, int64_t n, int* order, void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (n < 0) return FC_EINVAL;
  if (ws_bytes < argsort27_ws_bytes(n)) return FC_EWS;
  if (n == 0) return FC_OK;
  Batch<SortJob> bt;
  SortJob j;
  memset(&j, 0, sizeof(j));
  j.n = n; j.masks = const_cast<int*>(keys); j.order = order;
  sort_job_ws(j, ws);
  bt.add(j, j.nbx);
  return argsort27_batch(bt, stream);
}

}  // extern "C"
