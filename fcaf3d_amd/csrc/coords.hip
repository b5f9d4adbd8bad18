// Coordinate management for the sparse-voxel path on gfx950: voxelisation, voxel hash,
// first-occurrence compaction (wave ballot + prefix-sum), kernel maps, generative children,
// union maps, trilinear interpolation.  Integer work, HBM/L2-latency bound.
//
// Replaces what the reference gets from MinkowskiEngine's CoordinateManager at
//   mmdet3d/models/detectors/single_stage_sparse.py:34-37 (collate + SparseTensor),
//   every ME.Minkowski* call in me_resnet.py:19-24,56-62 and fcaf3d_neck_with_head.py:52-71,
//   fcaf3d_neck_with_head.py:101 (union), :115-116 (features_at_coordinates), :125 (pruning).
#include "fc_common.h"

extern "C" {

// ----------------------------------------------------------------------------------------------
// voxelise: coords = [b, floor(x/vs), floor(y/vs), floor(z/vs)], feats = rgb / feat_div
// (true fp32 division, as the reference's `p[:, :3] / voxel_size`, single_stage_sparse.py:35)
__global__ void k_voxelize(const float* __restrict__ pts, int64_t n, int pt_stride, int b, float vs, float feat_div,
                           int nfeat, int* __restrict__ coords, float* __restrict__ feats) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pts + i * pt_stride;
  int4 c;
  c.x = b;
  c.y = (int)floorf(p[0] / vs);
  c.z = (int)floorf(p[1] / vs);
  c.w = (int)floorf(p[2] / vs);
  reinterpret_cast<int4*>(coords)[i] = c;
  for (int j = 0; j < nfeat; ++j) feats[i * nfeat + j] = p[3 + j] / feat_div;
}

int fc_voxelize(const float* points, int64_t n, int pt_stride, int batch_idx, float voxel_size, float feat_div,
                int nfeat, int* coords, float* feats, hipStream_t stream) {
  if (n < 0 || pt_stride < 3 + nfeat || voxel_size <= 0.f) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_voxelize<<<(unsigned)fc_cdiv(n, 256), 256, 0, stream>>>(points, n, pt_stride, batch_idx, voxel_size, feat_div, nfeat,
                                                           coords, feats);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// Input pipeline fused with the voxelisation: one pass over the RAW points of a scene applies, in
// registers and in the reference's order, what its train pipeline does on the CPU before the detector sees the cloud —
//   GlobalAlignment (transforms_3d.py:409-490: p @ R^T + t), IndoorPointSample (:821-895: row gather),
//   RandomFlip3D (:59-170: x or y negated), GlobalRotScaleTrans (:493-645: p @ rot_T, * scale, + trans)
// — and then extract_feat's collate (single_stage_sparse.py:34-36: xyz / voxel_size floored, features / 255).
// The augmented cloud is never written unless the caller asks for it (points_out, tests / visualisation).
// xf (host, 24 floats): [0..8] alignment R (row-major; p' = R p + t), [9..11] t, [12] has_align, [13] flip x, [14] flip y,
// [15] cos(angle), [16] sin(angle), [17] scale, [18..20] translation.  Every step rounds to fp32 as the reference's
// tensor ops do (separate multiply / add: no contraction), so voxel indices agree except within an ulp of a cell face.
struct AugXf { float v[24]; };
__global__ void k_augment_voxelize(const float* __restrict__ pts, int64_t n_src, int pt_stride, const int* __restrict__ sample,
                                   int64_t n_out, AugXf xf, int b, float vs, float feat_div, int nfeat,
                                   int* __restrict__ coords, float* __restrict__ feats, float* __restrict__ points_out) {
#pragma clang fp contract(off)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const int64_t src = sample ? sample[i] : i;
  const float* p = pts + src * pt_stride;
  float x = p[0], y = p[1], z = p[2];
  const float* a = xf.v;
  if (a[12] != 0.f) {                                   // GlobalAlignment: points.rotate(R^T) then translate
    const float nx = (x * a[0] + y * a[1]) + z * a[2];
    const float ny = (x * a[3] + y * a[4]) + z * a[5];
    const float nz = (x * a[6] + y * a[7]) + z * a[8];
    x = nx + a[9]; y = ny + a[10]; z = nz + a[11];
  }
  if (a[13] != 0.f) x = -x;                             // RandomFlip3D 'horizontal'
  if (a[14] != 0.f) y = -y;                             // ... 'vertical'
  {                                                     // GlobalRotScaleTrans: p @ rot_T, rot_T = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    const float c = a[15], sn = a[16];
    const float nx = x * c - y * sn;
    const float ny = x * sn + y * c;
    x = nx * a[17] + a[18];
    y = ny * a[17] + a[19];
    z = z * a[17] + a[20];
  }
  int4 cd;
  cd.x = b;
  cd.y = (int)floorf(x / vs);
  cd.z = (int)floorf(y / vs);
  cd.w = (int)floorf(z / vs);
  reinterpret_cast<int4*>(coords)[i] = cd;
  for (int j = 0; j < nfeat; ++j) feats[i * nfeat + j] = p[3 + j] / feat_div;
  if (points_out) {
    float* o = points_out + i * (3 + nfeat);
    o[0] = x; o[1] = y; o[2] = z;
    for (int j = 0; j < nfeat; ++j) o[3 + j] = p[3 + j];
  }
}

int fc_augment_voxelize(const float* points, int64_t n_src, int pt_stride, const int* sample_idx, int64_t n_out,
                        const float* xform_host, int batch_idx, float voxel_size, float feat_div, int nfeat, int* coords,
                        float* feats, float* points_out, hipStream_t stream) {
  if (n_src < 0 || n_out < 0 || pt_stride < 3 + nfeat || voxel_size <= 0.f || !xform_host) return FC_EINVAL;
  if (!sample_idx && n_out != n_src) return FC_EINVAL;
  if (n_out == 0) return FC_OK;
  AugXf xf;
  for (int j = 0; j < 24; ++j) xf.v[j] = xform_host[j];
  k_augment_voxelize<<<(unsigned)fc_cdiv(n_out, 256), 256, 0, stream>>>(points, n_src, pt_stride, sample_idx, n_out, xf, batch_idx,
                                                                       voxel_size, feat_div, nfeat, coords, feats, points_out);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// Morton (Z-order) keys: batch index in the top bits, then x/y/z bit-interleaved.  Sorting the points by
// this key before the hash insert makes "order of first occurrence" a space-filling-curve order on EVERY
// pyramid level (a Z-order prefix is the Z-order of the parent cell), so the rows a convolution tile gathers
// are neighbours in memory as well as in space (L2 locality of the gather).
__device__ static inline unsigned long long spread3(unsigned int v) {      // 16 bits -> every third bit
  unsigned long long x = v & 0xFFFFull;
  x = (x | (x << 32)) & 0x00FF00000000FFFFull;   // not needed for 16 bits but keeps the classic ladder
  x = (x | (x << 16)) & 0x00FF0000FF0000FFull;
  x = (x | (x << 8)) & 0xF00F00F00F00F00Full;
  x = (x | (x << 4)) & 0x30C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x9249249249249249ull;
  return x;
}

__global__ void k_morton(const int4* __restrict__ coords, int64_t n, long long* __restrict__ keys) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  unsigned long long m = spread3((unsigned)(c.y + 32768)) | (spread3((unsigned)(c.z + 32768)) << 1) |
                         (spread3((unsigned)(c.w + 32768)) << 2);
  keys[i] = (long long)(((unsigned long long)(unsigned)c.x << 48) | (m & 0xFFFFFFFFFFFFull));
}

int fc_morton_keys(const int* coords, int64_t n, long long* keys, hipStream_t stream) {
  if (n < 0) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_morton<<<(unsigned)fc_cdiv(n, 256), 256, 0, stream>>>((const int4*)coords, n, keys);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// device-wide exclusive scan of byte flags:  pos[i] = #flags set before i ; *total = #set.
// 1024 items per 256-thread block, 4 rounds of 256 so that order is preserved.
__device__ static inline int block_excl_scan_flag(int flag, int* wave_sums /*[4]*/, int* block_total) {
  // returns exclusive position of this thread's flag within the 256-thread block
  unsigned long long bal = __ballot(flag);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int within = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_sums[w] = __popcll(bal);
  __syncthreads();
  int base = 0;
  for (int j = 0; j < w; ++j) base += wave_sums[j];
  *block_total = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
  __syncthreads();
  return base + within;
}

__global__ void k_flag_count(const unsigned char* __restrict__ flags, int64_t n, int* __restrict__ blocksums) {
  __shared__ int ws[4];
  int64_t base = (int64_t)blockIdx.x * 1024;
  int cnt = 0;
  for (int r = 0; r < 4; ++r) {
    int64_t i = base + r * 256 + threadIdx.x;
    int f = (i < n) ? (flags[i] != 0) : 0;
    unsigned long long bal = __ballot(f);
    if ((threadIdx.x & 63) == 0) cnt += __popcll(bal);
  }
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) blocksums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// single block, 1024 threads: exclusive scan of blocksums in place, total -> *total
__global__ void k_scan_blocksums(int* __restrict__ blocksums, int64_t nb, int* __restrict__ total) {
  __shared__ int buf[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t start = 0; start < nb; start += 1024) {
    int64_t i = start + threadIdx.x;
    int v = (i < nb) ? blocksums[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
      int t = (threadIdx.x >= off) ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    int incl = buf[threadIdx.x];
    int c = carry;
    if (i < nb) blocksums[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void k_flag_pos(const unsigned char* __restrict__ flags, int64_t n, const int* __restrict__ blocksums,
                           int* __restrict__ pos) {
  __shared__ int ws[4];
  int64_t base = (int64_t)blockIdx.x * 1024;
  int run = blocksums[blockIdx.x];
  for (int r = 0; r < 4; ++r) {
    int64_t i = base + r * 256 + threadIdx.x;
    int f = (i < n) ? (flags[i] != 0) : 0;
    int tot;
    int p = block_excl_scan_flag(f, ws, &tot);
    if (i < n) pos[i] = run + p;
    run += tot;
  }
}

static int scan_flags(const unsigned char* flags, int64_t n, int* pos, int* total_dev, int* blocksums,
                      hipStream_t stream) {
  int64_t nb = fc_cdiv(n, 1024);
  if (n == 0) {
    FC_HIP(hipMemsetAsync(total_dev, 0, sizeof(int), stream));
    return FC_OK;
  }
  k_flag_count<<<(unsigned)nb, 256, 0, stream>>>(flags, n, blocksums);
  FC_CHECK_LAUNCH();
  k_scan_blocksums<<<1, 1024, 0, stream>>>(blocksums, nb, total_dev);
  FC_CHECK_LAUNCH();
  k_flag_pos<<<(unsigned)nb, 256, 0, stream>>>(flags, n, blocksums, pos);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// public: compaction of an arbitrary byte mask. ws: >= 4*ceil(n/1024) bytes.
int fc_scan_flags(const unsigned char* flags, int64_t n, int* pos, int* total_dev, void* ws, int64_t ws_bytes,
                  hipStream_t stream) {
  if (n < 0) return FC_EINVAL;
  if (ws_bytes < (int64_t)sizeof(int) * fc_cdiv(n > 0 ? n : 1, 1024)) return FC_EWS;
  return scan_flags(flags, n, pos, total_dev, (int*)ws, stream);
}

__global__ void k_fill_i32(int* __restrict__ p, int64_t n, int v);
// kept[pos[i]] = i for flagged rows; entries at or past the output's capacity m are dropped (a caller that sizes `kept` from a
// count it knows by construction — sparse.compact_mask(expect_n) — can never be written out of bounds: ADVICE r5)
__global__ void k_scatter_kept(const unsigned char* __restrict__ flags, const int* __restrict__ pos, int64_t n,
                               int* __restrict__ kept, int64_t m) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i] && pos[i] < m) kept[pos[i]] = (int)i;
}

int fc_compact_rows(const unsigned char* flags, const int* pos, int64_t n, int* kept, int64_t m, hipStream_t stream) {
  if (n < 0 || m < 0) return FC_EINVAL;
  if (m > 0) {                                   // deterministic contents where fewer than m flags are set
    k_fill_i32<<<(unsigned)fc_cdiv(m, 256), 256, 0, stream>>>(kept, m, 0);
    FC_CHECK_LAUNCH();
  }
  if (n == 0 || m == 0) return FC_OK;
  k_scatter_kept<<<(unsigned)fc_cdiv(n, 256), 256, 0, stream>>>(flags, pos, n, kept, m);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// voxel hash + unique in order of first occurrence
__global__ void k_table_init(unsigned long long* __restrict__ keys, int* __restrict__ vals, int64_t cap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    keys[i] = FC_EMPTY_KEY;
    vals[i] = 0x7fffffff;
  }
}

__device__ static inline int4 quantize(int4 c, int q) {
  if (q > 1) {
    c.y = fc_floor_div(c.y, q) * q;
    c.z = fc_floor_div(c.z, q) * q;
    c.w = fc_floor_div(c.w, q) * q;
  }
  return c;
}

// `bad` (nullable): set when a coordinate does not fit fc_pack's 16 bits per axis (with room for the largest kernel
// offset) or the batch index its 16 bits — e.g. an inf / far-outlier point: such a key would alias another voxel.
__global__ void k_hash_insert(const int4* __restrict__ coords, int64_t n, int q, unsigned long long* keys, int* vals,
                              unsigned long long mask, int* __restrict__ slot, int* __restrict__ bad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = quantize(coords[i], q);
  if (bad && (c.x < 0 || c.x > 32767 || c.y < -FC_COORD_LIMIT || c.y > FC_COORD_LIMIT || c.z < -FC_COORD_LIMIT ||
              c.z > FC_COORD_LIMIT || c.w < -FC_COORD_LIMIT || c.w > FC_COORD_LIMIT))
    *bad = 1;
  unsigned long long key = fc_pack(c.x, c.y, c.z, c.w);
  unsigned long long h = fc_mix(key) & mask;
  while (true) {
    unsigned long long prev = atomicCAS(&keys[h], FC_EMPTY_KEY, key);
    if (prev == FC_EMPTY_KEY || prev == key) {
      atomicMin(&vals[h], (int)i);   // first occurrence (smallest row) wins — Appendix A.2
      slot[i] = (int)h;
      return;
    }
    h = (h + 1) & mask;
  }
}

__global__ void k_flag_bad_count(const int* __restrict__ bad, int* __restrict__ n_out_dev) {
  if (*bad) *n_out_dev = -1;
}

__global__ void k_winner_flags(const int* __restrict__ slot, const int* __restrict__ vals, int64_t n,
                               unsigned char* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (vals[slot[i]] == (int)i) ? 1 : 0;
}

__global__ void k_unique_finalize(const int4* __restrict__ coords, int64_t n, int q, const unsigned char* __restrict__ flags,
                                  const int* __restrict__ pos, const int* __restrict__ slot, int* vals,
                                  int4* __restrict__ out_coords, int* __restrict__ first_idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flags[i]) return;
  int p = pos[i];
  out_coords[p] = quantize(coords[i], q);
  if (first_idx) first_idx[p] = (int)i;
  vals[slot[i]] = p;   // table now maps key -> row of the new coordinate set
}

__global__ void k_unique_inverse(const int* __restrict__ slot, const int* __restrict__ vals, int64_t n,
                                 int* __restrict__ inverse) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inverse[i] = vals[slot[i]];
}

int64_t fc_hash_unique_ws_bytes(int64_t n) {
  int64_t m = n > 0 ? n : 1;
  return fc_align(4 * m, 256) /*slot*/ + fc_align(m, 256) /*flags*/ + fc_align(4 * m, 256) /*pos*/ + 256 /*range flag*/ +
         fc_align(4 * fc_cdiv(m, 1024), 256) /*blocksums*/;
}

// coords (n,4) int32 -> unique set quantised to multiples of `q` per spatial axis.
// table_keys/table_vals: `cap` entries, cap a power of two >= 2n (initialised here).
// out_coords: room for n rows; first_idx (nullable): n; inverse (nullable): n; n_out_dev: 1 int.
int fc_hash_unique(const int* coords, int64_t n, int q, unsigned long long* table_keys, int* table_vals, int64_t cap,
                   int* out_coords, int* first_idx, int* inverse, int* n_out_dev, void* ws, int64_t ws_bytes,
                   hipStream_t stream) {
  if (n < 0 || q < 1 || cap < 2 || (cap & (cap - 1)) || cap < 2 * n) return FC_EINVAL;
  if (ws_bytes < fc_hash_unique_ws_bytes(n)) return FC_EWS;
  k_table_init<<<(unsigned)fc_cdiv(cap, 256), 256, 0, stream>>>(table_keys, table_vals, cap);
  FC_CHECK_LAUNCH();
  if (n == 0) {
    FC_HIP(hipMemsetAsync(n_out_dev, 0, sizeof(int), stream));
    return FC_OK;
  }
  char* w = (char*)ws;
  int* slot = (int*)w;               w += fc_align(4 * n, 256);
  unsigned char* flags = (unsigned char*)w;  w += fc_align(n, 256);
  int* pos = (int*)w;                w += fc_align(4 * n, 256);
  int* bad = (int*)w;                w += 256;
  int* blocksums = (int*)w;
  unsigned g = (unsigned)fc_cdiv(n, 256);
  FC_HIP(hipMemsetAsync(bad, 0, sizeof(int), stream));
  k_hash_insert<<<g, 256, 0, stream>>>((const int4*)coords, n, q, table_keys, table_vals, (unsigned long long)(cap - 1), slot, bad);
  FC_CHECK_LAUNCH();
  k_winner_flags<<<g, 256, 0, stream>>>(slot, table_vals, n, flags);
  FC_CHECK_LAUNCH();
  int rc = scan_flags(flags, n, pos, n_out_dev, blocksums, stream);
  if (rc) return rc;
  k_unique_finalize<<<g, 256, 0, stream>>>((const int4*)coords, n, q, flags, pos, slot, table_vals, (int4*)out_coords, first_idx);
  FC_CHECK_LAUNCH();
  if (inverse) {
    k_unique_inverse<<<g, 256, 0, stream>>>(slot, table_vals, n, inverse);
    FC_CHECK_LAUNCH();
  }
  k_flag_bad_count<<<1, 1, 0, stream>>>(bad, n_out_dev);      // *n_out_dev = -1: the caller's count read-back raises
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// kernel map: nbr[k][o] = row of (out_coord[o] + offset[k]) in the input set, or -1
__global__ void k_kernel_map(const int4* __restrict__ out_coords, int64_t n_out, const unsigned long long* __restrict__ keys,
                             const int* __restrict__ vals, unsigned long long mask, const int* __restrict__ offsets, int K,
                             int* __restrict__ nbr) {
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (o >= n_out) return;
  int4 c = out_coords[o];
  int dx = offsets[3 * k], dy = offsets[3 * k + 1], dz = offsets[3 * k + 2];
  nbr[(int64_t)k * n_out + o] = fc_lookup(keys, vals, mask, fc_pack(c.x, c.y + dx, c.z + dy, c.w + dz));
}

int fc_kernel_map(const int* out_coords, int64_t n_out, const unsigned long long* table_keys, const int* table_vals,
                  int64_t cap, const int* offsets, int K, int* nbr, hipStream_t stream) {
  if (n_out < 0 || K < 1 || K > 65535 || (cap & (cap - 1))) return FC_EINVAL;
  if (n_out == 0) return FC_OK;
  dim3 grid((unsigned)fc_cdiv(n_out, 256), K);
  k_kernel_map<<<grid, 256, 0, stream>>>((const int4*)out_coords, n_out, table_keys, table_vals,
                                         (unsigned long long)(cap - 1), offsets, K, nbr);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// inverse map for dgrad: nbr_t[k][i] = o  iff  nbr[k][o] == i   (each (k,i) has at most one o)
__global__ void k_fill_i32(int* __restrict__ p, int64_t n, int v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void k_map_transpose(const int* __restrict__ nbr, int64_t n_out, int64_t n_in, int* __restrict__ nbr_t) {
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (o >= n_out) return;
  int i = nbr[(int64_t)k * n_out + o];
  if (i >= 0) nbr_t[(int64_t)k * n_in + i] = (int)o;
}

int fc_kernel_map_transpose(const int* nbr, int64_t n_out, int64_t n_in, int K, int* nbr_t, hipStream_t stream) {
  if (n_out < 0 || n_in < 0 || K < 1 || K > 65535) return FC_EINVAL;
  if (n_in > 0) {
    k_fill_i32<<<(unsigned)fc_cdiv(n_in * K, 256), 256, 0, stream>>>(nbr_t, n_in * K, -1);
    FC_CHECK_LAUNCH();
  }
  if (n_out == 0) return FC_OK;
  dim3 grid((unsigned)fc_cdiv(n_out, 256), K);
  k_map_transpose<<<grid, 256, 0, stream>>>(nbr, n_out, n_in, nbr_t);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// occupancy mask of every output row (bit k = it has a neighbour at offset k) and the neighbour table permuted
// into a given row order: convolution tiles / wgrad chunks made of rows with similar masks can skip the
// offsets none of their rows has (the redundancy of the dense (K,N) table on surface-like data, ~38 %).
__global__ void k_row_masks(const int* __restrict__ nbr, int64_t n_out, int K, int* __restrict__ masks) {
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  unsigned int m = 0;
  for (int k = 0; k < K; ++k)
    if (nbr[(int64_t)k * n_out + o] >= 0) m |= 1u << k;
  masks[o] = (int)m;
}

int fc_nbr_row_masks(const int* nbr, int64_t n_out, int K, int* masks, hipStream_t stream) {
  if (n_out < 0 || K < 1 || K > 31) return FC_EINVAL;
  if (n_out == 0) return FC_OK;
  k_row_masks<<<(unsigned)fc_cdiv(n_out, 256), 256, 0, stream>>>(nbr, n_out, K, masks);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

__global__ void k_permute_nbr(const int* __restrict__ nbr, const int* __restrict__ order, int64_t n_out,
                              int* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (t >= n_out) return;
  out[(int64_t)k * n_out + t] = nbr[(int64_t)k * n_out + order[t]];
}

int fc_permute_nbr(const int* nbr, const int* order, int64_t n_out, int K, int* nbr_sorted, hipStream_t stream) {
  if (n_out < 0 || K < 1 || K > 65535) return FC_EINVAL;
  if (n_out == 0) return FC_OK;
  dim3 grid((unsigned)fc_cdiv(n_out, 256), K);
  k_permute_nbr<<<grid, 256, 0, stream>>>(nbr, order, n_out, nbr_sorted);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// Exact pair lists of a neighbour table, per kernel offset, in ascending output-row order (deterministic):
// pair_out[k][j] = j-th output row o with nbr[k][o] >= 0, pair_in[k][j] = nbr[k][o], cnt[k] = number of pairs,
// pos[k][o] = j (or -1): where output row o sits in list k (nullable).
// The weight-gradient GEMM reduces over these lists, so absent neighbours cost no MFMA work.
#define PAIR_BLK 1024
__global__ __launch_bounds__(PAIR_BLK) void k_pairs_count(const int* __restrict__ nbr, int64_t n, int nblk,
                                                          int* __restrict__ blk_cnt) {
  const int k = blockIdx.y;
  int64_t row = (int64_t)blockIdx.x * PAIR_BLK + threadIdx.x;
  int present = row < n && nbr[(int64_t)k * n + row] >= 0;
  int c = __syncthreads_count(present);
  if (threadIdx.x == 0) blk_cnt[k * nblk + blockIdx.x] = c;
}

__global__ __launch_bounds__(PAIR_BLK) void k_pairs_fill(const int* __restrict__ nbr, int64_t n, int nblk,
                                                         const int* __restrict__ blk_cnt, int* __restrict__ pair_in,
                                                         int* __restrict__ pair_out, int* __restrict__ pos,
                                                         int* __restrict__ cnt) {
  __shared__ int wave_cnt[PAIR_BLK / 64];
  __shared__ int base_s;
  const int k = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {                                   // pairs of offset k in the blocks before this one
    int a = 0;
    for (int b = lane; b < blk; b += 64) a += blk_cnt[k * nblk + b];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) base_s = a;
  }
  int64_t row = (int64_t)blk * PAIR_BLK + tid;
  int v = row < n ? nbr[(int64_t)k * n + row] : -1;
  unsigned long long bal = __ballot(v >= 0);
  if (lane == 0) wave_cnt[wave] = __popcll(bal);
  __syncthreads();
  int pre = base_s;
  for (int w = 0; w < wave; ++w) pre += wave_cnt[w];
  const int j = pre + __popcll(bal & ((1ull << lane) - 1ull));
  if (v >= 0) {
    pair_in[(int64_t)k * n + j] = v;
    pair_out[(int64_t)k * n + j] = (int)row;
  }
  if (pos && row < n) pos[(int64_t)k * n + row] = v >= 0 ? j : -1;
  if (blk == nblk - 1 && tid == PAIR_BLK - 1) cnt[k] = pre + __popcll(bal);
}

int64_t fc_kernel_map_pairs_ws_bytes(int64_t n_out, int K) {
  return (int64_t)K * fc_cdiv(n_out > 0 ? n_out : 1, PAIR_BLK) * (int64_t)sizeof(int);
}

int fc_kernel_map_pairs(const int* nbr, int64_t n_out, int K, int* pair_in, int* pair_out, int* pair_pos, int* cnt,
                        void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (n_out < 0 || K < 1 || K > 65535) return FC_EINVAL;
  if (n_out == 0) {
    FC_HIP(hipMemsetAsync(cnt, 0, (size_t)K * sizeof(int), stream));
    return FC_OK;
  }
  if (ws_bytes < fc_kernel_map_pairs_ws_bytes(n_out, K)) return FC_EWS;
  int nblk = (int)fc_cdiv(n_out, PAIR_BLK);
  dim3 grid((unsigned)nblk, K);
  k_pairs_count<<<grid, PAIR_BLK, 0, stream>>>(nbr, n_out, nblk, (int*)ws);
  FC_CHECK_LAUNCH();
  k_pairs_fill<<<grid, PAIR_BLK, 0, stream>>>(nbr, n_out, nblk, (const int*)ws, pair_in, pair_out, pair_pos, cnt);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// generative transposed conv k2 s2: child row 8*i + k at c_i + {0,half}^3 (x fastest) — Appendix A.4
__global__ void k_gen_coords(const int4* __restrict__ coords, int64_t n, int half, int4* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 8) return;
  int64_t i = t >> 3;
  int k = (int)(t & 7);
  int4 c = coords[i];
  c.y += (k & 1) ? half : 0;
  c.z += (k & 2) ? half : 0;
  c.w += (k & 4) ? half : 0;
  out[t] = c;
}

int fc_gen_coords(const int* coords, int64_t n, int half_stride, int* out_coords, hipStream_t stream) {
  if (n < 0 || half_stride < 1) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_gen_coords<<<(unsigned)fc_cdiv(n * 8, 256), 256, 0, stream>>>((const int4*)coords, n, half_stride, (int4*)out_coords);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// Structure of a GENERATED set (r3): the children set of MinkowskiGenerativeConvolutionTranspose(k2,s2) holds all 8
// children of every parent, child k of parent row i at row 8i + k (k = bx + 2by + 4bz).  Its 3x3x3 kernel map and the
// row of any voxel inside it therefore follow from the PARENT level by index arithmetic — no hash of the 8x larger set
// is ever built or probed (the 441k-row finest neck level was 90 % of all hash probes of a step).
//   neighbour of child (i, k) at offset d in {-1,0,1}^3 (child-stride units): t = bit_k + d in {-1..2} per axis,
//   parent offset D = floor(t / 2) in {-1,0,1}, child bit t - 2D  ->  8 * parent_nbr[D][i] + bits   (or -1)
__global__ void k_kernel_map_children(const int* __restrict__ pnbr, int64_t n_parent, int* __restrict__ nbr) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // child row
  const int d = blockIdx.y;                                                 // child offset index, x fastest
  const int64_t n_child = n_parent * 8;
  if (c >= n_child) return;
  const int64_t i = c >> 3;
  const int k = (int)(c & 7);
  int Didx = 0, bits = 0, mul = 1, dd = d;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const int t = ((k >> ax) & 1) + (dd % 3 - 1);
    dd /= 3;
    const int D = t < 0 ? -1 : (t > 1 ? 1 : 0);
    Didx += (D + 1) * mul;
    mul *= 3;
    bits |= (t - 2 * D) << ax;
  }
  const int j = pnbr[(int64_t)Didx * n_parent + i];
  nbr[(int64_t)d * n_child + c] = j < 0 ? -1 : 8 * j + bits;
}

int fc_kernel_map_children(const int* parent_nbr, int64_t n_parent, int* nbr, hipStream_t stream) {
  if (n_parent < 0 || (!parent_nbr && n_parent > 0) || 8 * n_parent > 0x7fffffffLL) return FC_EINVAL;
  if (n_parent == 0) return FC_OK;
  dim3 grid((unsigned)fc_cdiv(n_parent * 8, 256), 27);
  k_kernel_map_children<<<grid, 256, 0, stream>>>(parent_nbr, n_parent, nbr);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// rows[i] = row of query voxel i (coordinates multiples of the child stride T) in the children set of the parent table's
// set (stride 2T), or -1 when its parent cell is absent; *n_found counts the hits (zeroed here)
__global__ void k_child_rows(const int4* __restrict__ q, int64_t n, const unsigned long long* __restrict__ keys,
                             const int* __restrict__ vals, unsigned long long mask, int T, int* __restrict__ rows,
                             int* __restrict__ n_found) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int hit = 0;
  if (i < n) {
    const int4 c = q[i];
    const int P = 2 * T;
    const int px = fc_floor_div(c.y, P) * P, py = fc_floor_div(c.z, P) * P, pz = fc_floor_div(c.w, P) * P;
    const int r = fc_lookup(keys, vals, mask, fc_pack(c.x, px, py, pz));
    const int bits = ((c.y - px) / T) | (((c.z - py) / T) << 1) | (((c.w - pz) / T) << 2);
    rows[i] = r < 0 ? -1 : 8 * r + bits;
    hit = r >= 0;
  }
  const unsigned long long bal = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_found, __popcll(bal));
}

int fc_child_rows(const int* query_coords, int64_t n, const unsigned long long* parent_keys, const int* parent_vals,
                  int64_t cap, int child_stride, int* rows, int* n_found_dev, hipStream_t stream) {
  if (n < 0 || child_stride < 1 || (cap & (cap - 1)) || !n_found_dev) return FC_EINVAL;
  FC_HIP(hipMemsetAsync(n_found_dev, 0, sizeof(int), stream));
  if (n == 0) return FC_OK;
  k_child_rows<<<(unsigned)fc_cdiv(n, 256), 256, 0, stream>>>((const int4*)query_coords, n, parent_keys, parent_vals,
                                                              (unsigned long long)(cap - 1), child_stride, rows, n_found_dev);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// union map (a + b on different coordinate sets, Appendix A.8): for each row of b either the row of the
// equal coordinate in a, or n_a + (rank among b's new rows).  new_coords receives b's new rows in order.
__global__ void k_union_probe(const int4* __restrict__ coords_b, int64_t n_b, const unsigned long long* __restrict__ keys,
                              const int* __restrict__ vals, unsigned long long mask, int* __restrict__ row_b,
                              unsigned char* __restrict__ is_new) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_b) return;
  int4 c = coords_b[i];
  int r = fc_lookup(keys, vals, mask, fc_pack(c.x, c.y, c.z, c.w));
  row_b[i] = r;
  is_new[i] = r < 0;
}

__global__ void k_union_finalize(const int4* __restrict__ coords_b, int64_t n_b, int n_a, const unsigned char* __restrict__ is_new,
                                 const int* __restrict__ pos, int* __restrict__ row_b, int4* __restrict__ new_coords) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_b || !is_new[i]) return;
  row_b[i] = n_a + pos[i];
  new_coords[pos[i]] = coords_b[i];
}

int64_t fc_union_map_ws_bytes(int64_t n_b) {
  int64_t m = n_b > 0 ? n_b : 1;
  return fc_align(m, 256) + fc_align(4 * m, 256) + fc_align(4 * fc_cdiv(m, 1024), 256);
}

int fc_union_map(const int* coords_b, int64_t n_b, const unsigned long long* table_keys_a, const int* table_vals_a,
                 int64_t cap_a, int64_t n_a, int* row_b, int* new_coords, int* n_new_dev, void* ws, int64_t ws_bytes,
                 hipStream_t stream) {
  if (n_b < 0 || n_a < 0 || (cap_a & (cap_a - 1))) return FC_EINVAL;
  if (ws_bytes < fc_union_map_ws_bytes(n_b)) return FC_EWS;
  if (n_b == 0) {
    FC_HIP(hipMemsetAsync(n_new_dev, 0, sizeof(int), stream));
    return FC_OK;
  }
  char* w = (char*)ws;
  unsigned char* is_new = (unsigned char*)w;  w += fc_align(n_b, 256);
  int* pos = (int*)w;                         w += fc_align(4 * n_b, 256);
  int* blocksums = (int*)w;
  unsigned g = (unsigned)fc_cdiv(n_b, 256);
  k_union_probe<<<g, 256, 0, stream>>>((const int4*)coords_b, n_b, table_keys_a, table_vals_a,
                                       (unsigned long long)(cap_a - 1), row_b, is_new);
  FC_CHECK_LAUNCH();
  int rc = scan_flags(is_new, n_b, pos, n_new_dev, blocksums, stream);
  if (rc) return rc;
  k_union_finalize<<<g, 256, 0, stream>>>((const int4*)coords_b, n_b, (int)n_a, is_new, pos, row_b, (int4*)new_coords);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// trilinear interpolation of a coarse tensor (stride S) at integer query coordinates — the
// features_at_coordinates call of fcaf3d_neck_with_head.py:116 (Appendix A.8).
__global__ void k_interp(const int4* __restrict__ q, int64_t n, const unsigned long long* __restrict__ keys,
                         const int* __restrict__ vals, unsigned long long mask, const float* __restrict__ feats, int C,
                         int S, float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = q[i];
  int bx = fc_floor_div(c.y, S) * S, by = fc_floor_div(c.z, S) * S, bz = fc_floor_div(c.w, S) * S;
  float fx = (float)(c.y - bx) / (float)S, fy = (float)(c.z - by) / (float)S, fz = (float)(c.w - bz) / (float)S;
  for (int ch = 0; ch < C; ++ch) out[i * C + ch] = 0.f;
  for (int k = 0; k < 8; ++k) {
    int ox = k & 1, oy = (k >> 1) & 1, oz = (k >> 2) & 1;
    float w = (ox ? fx : 1.f - fx) * (oy ? fy : 1.f - fy) * (oz ? fz : 1.f - fz);
    if (w == 0.f) continue;
    int r = fc_lookup(keys, vals, mask, fc_pack(c.x, bx + ox * S, by + oy * S, bz + oz * S));
    if (r < 0) continue;
    for (int ch = 0; ch < C; ++ch) out[i * C + ch] += w * feats[(int64_t)r * C + ch];
  }
}

int fc_interp(const int* query_coords, int64_t n, const unsigned long long* table_keys, const int* table_vals, int64_t cap,
              const float* feats, int C, int tensor_stride, float* out, hipStream_t stream) {
  if (n < 0 || C < 1 || tensor_stride < 1 || (cap & (cap - 1))) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_interp<<<(unsigned)fc_cdiv(n, 256), 256, 0, stream>>>((const int4*)query_coords, n, table_keys, table_vals,
                                                         (unsigned long long)(cap - 1), feats, C, tensor_stride, out);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ----------------------------------------------------------------------------------------------
// int32 row gather for coordinates (pruned coordinate set): dst[i] = src[idx[i]]
__global__ void k_gather_int4(const int4* __restrict__ src, const int* __restrict__ idx, int64_t n, int4* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

int fc_gather_coords(const int* src, const int* idx, int64_t n, int* dst, hipStream_t stream) {
  if (n < 0) return FC_EINVAL;
  if (n == 0) return FC_OK;
  k_gather_int4<<<(unsigned)fc_cdiv(n, 256), 256, 0, stream>>>((const int4*)src, idx, n, (int4*)dst);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"
