/* libfcaf3d_hip.so — C ABI of the MI355X-native FCAF3D sparse-voxel hot path.
 *
 * Every entry point takes raw DEVICE pointers, explicit sizes and a hipStream_t, enqueues its
 * kernels on that stream and returns immediately:
 *     0  = ok,  <0 = invalid argument (-1) / workspace too small (-2),  >0 = hipError_t.
 * The library never allocates device memory: data-dependent output sizes come back through a
 * device counter (`*_dev`), and scratch space is passed in (`ws`, size from the matching
 * `*_ws_bytes`).  (It owns a handful of hipEvents: fc_exec, fc_plan_*.)  It never calls exit() (contrast mmdet3d/ops/pcdet_nms/src/iou3d_nms.cpp:14-38).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference
 * repository; "ME" = MinkowskiEngine v0.5.4, the un-vendored dependency pinned at
 * docker/Dockerfile:27-32, named by its Python call site).
 *
 * Layouts: coords int32 (N,4) = [batch, x, y, z] in voxel units (multiples of the tensor stride);
 * features fp32 (N,C) row-major; conv kernels fp32 (K,Cin,Cout) with offsets x-fastest;
 * neighbour tables int32 (K,N_out), -1 = absent; voxel hash = open addressing over 64-bit packed
 * keys (`cap` a power of two >= 2N) with int32 values = row index.
 */
#ifndef FCAF3D_HIP_H
#define FCAF3D_HIP_H
#include <stdint.h>

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this ABI (No reference counterpart): bumped whenever a prototype or the meaning of a flag bit changes — r5 -> 5
 * (flags bit27 went from "input is pre-split planes" to "flat addressing", fc_x6_planes / fc_conv_x6d removed: ADVICE r5),
 * r6 -> 6 (fc_compact_rows takes the output capacity; fc_plan_*, fc_argsort27 added), r6 -> 7 (fc_set_split_mode / fc_get_split_mode / fc_amax /
 * fc_conv_amax_hint / fc_amax_out_hint added; weight images are mode-dependent and fc_x6_weight_images writes their amax word).  A caller built against another
 * version must not go on: tests/test_cabi.py pins the number the Python host was written for. */
#define FC_ABI_VERSION 7
#ifndef FC_AMAX_SLOT_BYTES
#define FC_AMAX_SLOT_BYTES 2048
#endif
int fc_abi_version(void);

/* ---- coordinates -------------------------------------------------------------------------- */

/* ME.utils.batch_sparse_collate quantisation of one scene — single_stage_sparse.py:34-36.
 * coords[i] = [batch_idx, floor(xyz/voxel_size)], feats[i] = points[i,3:3+nfeat] / feat_div. */
int fc_voxelize(const float* points, int64_t n, int pt_stride, int batch_idx, float voxel_size, float feat_div,
                int nfeat, int* coords, float* feats, hipStream_t stream);

/* The reference's indoor TRAIN pipeline fused with the collate above, one pass over the raw points of a scene:
 * GlobalAlignment (mmdet3d/datasets/pipelines/transforms_3d.py:409-490), IndoorPointSample (:821-895, sample_idx: n_out
 * row indices, nullable = every row), RandomFlip3D (:59-170), GlobalRotScaleTrans (:493-645), then
 * ME.utils.batch_sparse_collate's floor(xyz / voxel_size) and features / feat_div (single_stage_sparse.py:34-36).
 * xform_host: 24 floats on the HOST — [0..8] alignment R row-major, [9..11] t, [12] has_align, [13] flip x, [14] flip y,
 * [15] cos(angle), [16] sin(angle), [17] scale, [18..20] translation, [21..23] unused.  points_out (nullable):
 * the augmented cloud (n_out, 3 + nfeat), for inspection — the detector itself needs coords / feats only. */
int fc_augment_voxelize(const float* points, int64_t n_src, int pt_stride, const int* sample_idx, int64_t n_out,
                        const float* xform_host, int batch_idx, float voxel_size, float feat_div, int nfeat, int* coords,
                        float* feats, float* points_out, hipStream_t stream);

/* Z-order key per voxel coordinate [b | x,y,z bit-interleaved]; sorting the collated points by it before
 * fc_hash_unique turns "order of first occurrence" into a space-filling-curve order on every pyramid level.
 * No reference counterpart (optional locality aid of this implementation, off by default). */
int fc_morton_keys(const int* coords, int64_t n, long long* keys, hipStream_t stream);

/* Order-preserving compaction primitive (wave ballot + prefix sum): pos[i] = #set flags before i — the row selection
 * of ME.MinkowskiPruning (fcaf3d_neck_with_head.py:76, called at :124-125) and of the per-scene decomposition.
 * fc_compact_rows: kept has room for m entries (r6: set flags past that capacity are dropped, missing ones read 0). */
int fc_scan_flags(const unsigned char* flags, int64_t n, int* pos, int* total_dev, void* ws, int64_t ws_bytes,
                  hipStream_t stream);
int fc_compact_rows(const unsigned char* flags, const int* pos, int64_t n, int* kept, int64_t m, hipStream_t stream);

/* ME.SparseTensor(coordinates, features) de-duplication (single_stage_sparse.py:37) and the strided
 * output coordinate set of MinkowskiConvolution/MaxPooling(stride=2) (me_resnet.py:19-24,56-62):
 * unique rows of floor(coords/q)*q in order of first occurrence; builds the voxel hash of the
 * result (key -> row).  first_idx / inverse may be NULL.  Coordinates must lie in [-32639, 32639] per axis and the
 * batch index in [0, 32767] (64-bit packed hash keys, 16 bits per field): if any row does not — a stray outlier, an
 * inf coordinate — *n_out_dev is set to -1 and the set must not be used. */
int64_t fc_hash_unique_ws_bytes(int64_t n);
int fc_hash_unique(const int* coords, int64_t n, int q, unsigned long long* table_keys, int* table_vals, int64_t cap,
                   int* out_coords, int* first_idx, int* inverse, int* n_out_dev, void* ws, int64_t ws_bytes,
                   hipStream_t stream);

/* ME CoordinateManager kernel map for one (in set, out set, kernel) triple: nbr[k][o] = input row at
 * out_coords[o] + offsets[k] — built inside every ME.MinkowskiConvolution / MinkowskiMaxPooling call
 * (me_resnet.py:19-24, :56-62; fcaf3d_neck_with_head.py:52, :69). */
int fc_kernel_map(const int* out_coords, int64_t n_out, const unsigned long long* table_keys, const int* table_vals,
                  int64_t cap, const int* offsets, int K, int* nbr, hipStream_t stream);
/* per-row occupancy masks of a neighbour table, and the table permuted into a row order (mask-sorted rows).
 * No reference counterpart: scheduling aid of the output-stationary kernel. */
int fc_nbr_row_masks(const int* nbr, int64_t n_out, int K, int* masks, hipStream_t stream);
int fc_permute_nbr(const int* nbr, const int* order, int64_t n_out, int K, int* nbr_sorted, hipStream_t stream);
/* exact (input row, output row) pair lists per kernel offset, ascending in the output row — what ME's kernel map
 * (in_maps / out_maps per offset; consumed by the convolutions of me_resnet.py:56-62) holds; the weight-gradient pass reduces over them (fc_conv_wgrad_pairs).
 * pair_in / pair_out are (K, n_out) int32 with the first cnt[k] entries of row k valid; pair_pos (nullable,
 * (K, n_out)) is the inverse: pair_pos[k][o] = j with pair_out[k][j] == o, or -1. */
int64_t fc_kernel_map_pairs_ws_bytes(int64_t n_out, int K);
int fc_kernel_map_pairs(const int* nbr, int64_t n_out, int K, int* pair_in, int* pair_out, int* pair_pos, int* cnt,
                        void* ws, int64_t ws_bytes, hipStream_t stream);
/* nbr_t[k][i] = o  iff  nbr[k][o] == i  (the gather table of the backward-data pass: ME's convolution backward walks the
 * same in/out maps with the roles swapped, `gin[i] += gout[o] @ W[k]^T`; reached by autograd of me_resnet.py:56-62). */
int fc_kernel_map_transpose(const int* nbr, int64_t n_out, int64_t n_in, int K, int* nbr_t, hipStream_t stream);

/* ME.MinkowskiGenerativeConvolutionTranspose(k=2,s=2) output coordinates — fcaf3d_neck_with_head.py:60-66:
 * out[8i+k] = coords[i] + {0,half_stride}^3 (x fastest). */
int fc_gen_coords(const int* coords, int64_t n, int half_stride, int* out_coords, hipStream_t stream);

/* Kernel map and membership of a GENERATED children set (the output set of ME.MinkowskiGenerativeConvolutionTranspose,
 * fcaf3d_neck_with_head.py:60-66, on which the neck's k3 convolutions run, :52, :69) from the PARENT level by index
 * arithmetic instead of hash probes: child k of parent row i sits at row 8i + k.
 *   fc_kernel_map_children: nbr (27, 8 n_parent) of the children set onto itself (k3 s1) from the parent set's own
 *     k3 s1 table parent_nbr (27, n_parent) — identical to fc_kernel_map on the children's hash;
 *   fc_child_rows: row of each query voxel (stride T) in the children set of the set behind the parent table (stride 2T),
 *     -1 if its parent cell is absent; *n_found_dev = number of hits — what the sparse `a + b` union (:101) needs when
 *     the backbone level lies inside the generated set. */
int fc_kernel_map_children(const int* parent_nbr, int64_t n_parent, int* nbr, hipStream_t stream);
int fc_child_rows(const int* query_coords, int64_t n, const unsigned long long* parent_keys, const int* parent_vals,
                  int64_t cap, int child_stride, int* rows, int* n_found_dev, hipStream_t stream);

/* SparseTensor `a + b` on different coordinate maps — fcaf3d_neck_with_head.py:101: row of every b
 * voxel in the union (a's rows first, then b's new voxels in order); new_coords gets b's new rows. */
int64_t fc_union_map_ws_bytes(int64_t n_b);
int fc_union_map(const int* coords_b, int64_t n_b, const unsigned long long* table_keys_a, const int* table_vals_a,
                 int64_t cap_a, int64_t n_a, int* row_b, int* new_coords, int* n_new_dev, void* ws, int64_t ws_bytes,
                 hipStream_t stream);

/* SparseTensor.features_at_coordinates (MinkowskiInterpolation) — fcaf3d_neck_with_head.py:115-116. */
int fc_interp(const int* query_coords, int64_t n, const unsigned long long* table_keys, const int* table_vals, int64_t cap,
              const float* feats, int C, int tensor_stride, float* out, hipStream_t stream);

/* coordinate rows of a pruned set (MinkowskiPruning, fcaf3d_neck_with_head.py:125). */
int fc_gather_coords(const int* src, const int* idx, int64_t n, int* dst, hipStream_t stream);

/* ---- the coordinate phase of a step as two native calls (r6; csrc/plan.hip) ------------------------------------------------ */

/* SingleStageSparse3DDetector.extract_feat's collate + ME.SparseTensor (mmdet3d/models/detectors/single_stage_sparse.py:34-37) and
 * the coordinate sets behind every strided ME.MinkowskiConvolution / MinkowskiMaxPooling of the backbone (me_resnet.py:19-24,
 * :56-62) in ONE call: points of B scenes -> the sets [cm0, m1, m2, L1..Lnl] (strides 1, 2, 4, 8 ...), each with its voxel hash,
 * rows in order of first occurrence, level 0 with its feature rows.  The chain runs with device-resident row counts; the call
 * ends with its single host read-back (rows per set, rows per set and scene, range flags) and fills `out`.
 *   cfg (fc_plan_cfg_words() int64): [0] B, [1] nl, [2] nfeat, [3] voxel size (double bits), [4] feature divisor (double bits),
 *     [5] total points, [6] backward tables wanted, [7] mask-sorted tables from this many rows, [8] pair-list convolution up to this
 *     many rows, [9] pts_threshold (-1: none), [10] head location arrays wanted, [11] / [12] pre-voxelised coords / feats (device
 *     pointers, 0 = collate from the points), [13] floats per point, [14] neck sets wanted, [15] the head's voxel size (double bits).
 *   scenes: B x {device pointer, points, floats per point} (int64).  arena1: fc_plan_stage1_bytes bytes of device memory.
 *   counts_host: PINNED host ints, 8 (3 + nl) + (3 + nl) B.  out: fc_plan_out_words int64 (layout: csrc/plan.hip, fcaf3d_amd/plan.py).
 * fc_plan_maps, with those counts: every kernel map of the backbone and of the neck (fcaf3d_neck_with_head.py:52, :60-71, :101:
 * generated children sets and their maps by index arithmetic, union rows of the backbone levels), the derived tables of each
 * convolution route (mask-sorted tables, pair lists, transposed tables) and the head's location / scene / level arrays
 * (fcaf3d_neck_with_head.py:276-277) into arena2 (fc_plan_stage2_bytes), ending with ITS single read-back (pair-list counts, union
 * hits) into cnt_host (PINNED ints, 64 maps + 8 + nl B + 1).  Same sets, tables and row orders as the per-operator entry points above.
 * fc_argsort27: the stable argsort of 27-bit occupancy masks used for the mask-sorted tables (No reference counterpart).
 * cfg[16] != 0: every launch (group) of the two calls is bracketed with a HIP-event pair; fc_plan_probe_read(ms, bytes, kind, cap),
 * called after the device has drained, returns the brackets since the last read-out with their compulsory bytes and kind (0 tables,
 * 1 collate + insert, 2 winner flags + scan, 3 finalize + next insert, 4 generated coordinates, 5 kernel maps, 6 children maps, 7 fills,
 * 8 transposes, 9 row masks, 10 radix argsort, 11 permute, 12 pair lists, 13 union rows, 14 head arrays): bench.py's
 * `roofline.hbm_kernels` (the reference times whole iterations only: tools/analysis_tools/benchmark.py:64-91). */
int fc_plan_cfg_words(void);
int fc_plan_out_words(int B, int nl);
int64_t fc_plan_stage1_bytes(int64_t total_points, int B, int nl, int nfeat);
int fc_plan_levels(const int64_t* cfg, const int64_t* scenes, void* arena1, int64_t arena1_bytes, int64_t* out, int* counts_host,
                   hipStream_t stream);
int64_t fc_plan_stage2_bytes(const int64_t* cfg, int64_t* out, const int* counts_host);
int fc_plan_maps(const int64_t* cfg, int64_t* out, const int* counts_host, void* arena2, int64_t arena2_bytes, int* cnt_host,
                 hipStream_t stream);
int64_t fc_plan_probe_read(float* ms, double* bytes, int* kind, int64_t cap);
int64_t fc_argsort27_ws_bytes(int64_t n);
int fc_argsort27(const int* keys, int64_t n, int* order, void* ws, int64_t ws_bytes, hipStream_t stream);

/* ---- sparse convolution ------------------------------------------------------------------- */

/* flags bit24 (fc_conv_fwd, fc_conv_fwd_pairs, fc_conv_fwd_pairs_tiles): the same fp32 convolution (torch.float32 in and
 * out, as ME.MinkowskiConvolution computes it, me_resnet.py:56-62) on the bf16 matrix pipe by EXACT operand splitting —
 * x = x1 + x2 + x3 with three 8-bit pieces, six bf16 x bf16 products (each exact in the fp32 accumulator) per fp32 product
 * (the three dropped ones: <= 2^-24 of the product with the round-to-nearest split of r4);
 * results sit at fp32 rounding level against fp64, like the fp32 MFMA's (csrc/conv_x6.h, tests/test_gpu_ops.py).  128- and 256-row tiles.
 * flags bit26 (with bit24): `W` is not the fp32 kernel but its pre-split image built by fc_x6_weight_image — for the
 * backward-data pass the image of the transposed operator (then bit23 is not needed).
 * fc_x6_weight_image: image of W (K, R, C) — or, transposed != 0, of the operator W[k]^T where W[k] is stored (C, R) —
 * for a launch with Cin = R, Cout = C; R % 32 == 0, C % 64 == 0; fc_x6_weight_image_bytes(K, R, C) = 6 K R C bytes.
 * flags bit27 (with bit24 | bit26, r5): flat 64-bit addresses for the gathered rows and the image.  Without it a gathering launch
 * whose `in` ends below 2 GB (n_in Cin 4 bytes) reads both through buffer descriptors — a lane's row is a 32-bit byte offset
 * computed once per kernel offset, an absent neighbour an offset past the descriptor's end (the load returns zeros): the same
 * loads, bit-identical results, a third fewer address instructions per stage (csrc/conv_x6.h BUF).  Larger operands and
 * table-free launches (nbr == NULL) take the flat route by themselves.  The descriptor route addresses the weight image with
 * 32-bit offsets: a caller whose image reaches 4 GB (6 K Cin Cout bytes; 42 MB for the largest layer of the reference's networks)
 * sets bit27. */
int64_t fc_x6_weight_image_bytes(int K, int R, int C);
int fc_x6_weight_image(const float* W, void* img, int K, int R, int C, int transposed, hipStream_t stream);
/* The images of many kernels in ONE launch (all convolutions of a model — me_resnet.py:56-62, fcaf3d_neck_with_head.py:52,60-69 —
 * both directions, right after the optimizer step: the reference's optimizer hook, mmcv OptimizerHook.after_train_iter, is where
 * its weights change): `desc` is a DEVICE array
 * of n entries of 8 int64 {W pointer, image pointer, K, R, C, transposed, first block, sibling}; sibling (split mode 2): 0, or
 * the address of another image of the SAME weights listed earlier (the forward image of a backward-data entry) whose amax slot
 * this entry shares instead of making its own pass over W; entry e owns the blocks from its
 * first block up to the next entry's, K (R / 32) (C / 64) of them; total_blocks = the last entry's first block + its blocks. */
int fc_x6_weight_images(const int64_t* desc, int n, int64_t total_blocks, hipStream_t stream);

/* ME.MinkowskiConvolution forward (me_resnet.py:19-21,56-62; BasicBlock; fcaf3d_neck_with_head.py:52,69),
 * its backward-data pass (call with the transposed table and fc_transpose_weight'ed kernel), and with
 * nbr == NULL (K = 1, identity) the dense GEMMs of MinkowskiGenerativeConvolutionTranspose (:60-66)
 * and of the 1x1 head convolutions (:83-85, :257-263).  out[o] = sum_k in[nbr[k][o]] @ W[k].
 * flags bit0: force the generic FMA kernel instead of the MFMA kernel.  flags bit23 (also fc_conv_fwd_pairs /
 * _pairs_tiles; needs nbr != NULL): W[k] is stored TRANSPOSED, (Cout, Cin) row-major — the backward-data pass run on the
 * layer's own (K, Cin, Cout) kernel without a transposed copy.  Layers with too few rows to fill
 * the chip are split over kernel offsets into `ws` and summed in a fixed order (deterministic).
 * out_index (nullable): `nbr` is a table permuted into occupancy-mask order (fc_permute_nbr) and tile row t
 * belongs to output row out_index[t] — tiles of similar rows skip the offsets none of them has. */
int64_t fc_conv_fwd_ws_bytes(int64_t n_out, int K, int Cin, int Cout, int flags);
int fc_conv_fwd(const float* in, const float* W, const int* nbr, const int* out_index, float* out, int64_t n_in,
                int64_t n_out, int K, int Cin, int Cout, int flags, void* ws, int64_t ws_bytes, hipStream_t stream);

/* ME.MinkowskiConvolution (me_resnet.py:56-62, BasicBlock convs; fcaf3d_neck_with_head.py:52) the way ME itself runs it
 * (per offset: gather -> GEMM -> scatter) over the exact pair lists:
 * T_k = in[pair_in[k][:cnt[k]]] @ W[k] into the workspace, then out[o] = sum_k T_k[pair_pos[k][o]] in fixed k order.
 * No MFMA work is issued for absent neighbours; pays off on the small, ~60 % occupied deep levels.  MFMA shapes only
 * (Cin % 32 == 0, Cout % 64 == 0).  For the backward-data pass pass the lists of the transposed table. */
int64_t fc_conv_fwd_pairs_ws_bytes(int64_t n_out, int K, int Cout);
int fc_conv_fwd_pairs(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                      float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int flags, void* ws,
                      int64_t ws_bytes, hipStream_t stream);

/* The same (ME.MinkowskiConvolution per offset over its in/out maps, me_resnet.py:56-62) for a caller that has read the pair
 * counts back: live_tiles = sum_k ceil(pair_cnt[k] / 128) launches exactly the non-empty (offset, 128-row tile) workgroups
 * as one linear list (no workgroup exits on arrival; evenly filled shader engines).  live_tiles <= 0: as fc_conv_fwd_pairs. */
int fc_conv_fwd_pairs_tiles(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                            float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int64_t live_tiles, int flags,
                            void* ws, int64_t ws_bytes, hipStream_t stream);

/* r5 — the convolution -> BatchNorm pairs of the network (BasicBlock conv1 / norm1, conv2 / norm2, me_resnet.py:3, :56-63; the
 * neck's blocks, fcaf3d_neck_with_head.py:52-53, :60-62, :66-68): fc_conv_fwd / fc_conv_fwd_pairs_tiles that ALSO leave, per row
 * block of their result, the column sums of the result and of its square — stats[fc_conv_stats_blocks(...)][2][Cout] — written
 * by whichever kernel produces the final rows (the MFMA tile epilogue, or the fixed-order sum of an offset-split / pair-list
 * launch).  fc_bn_train_fwd(part = stats) turns them into the batch statistics without reading the matrix again.  Split-bf16
 * route only (flags bits 24 | 26); fc_conv_stats_blocks returns 0 where a launch has no statistics epilogue.  stats == NULL:
 * exactly the plain entry points. */
int64_t fc_conv_stats_blocks(int64_t n_out, int K, int Cin, int Cout, int flags, int pairs);
int fc_conv_fwd_stats(const float* in, const float* W, const int* nbr, const int* out_index, float* out, int64_t n_in,
                      int64_t n_out, int K, int Cin, int Cout, int flags, void* ws, int64_t ws_bytes, float* stats,
                      hipStream_t stream);
int fc_conv_fwd_pairs_tiles_stats(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                                  float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int64_t live_tiles, int flags,
                                  void* ws, int64_t ws_bytes, float* stats, hipStream_t stream);
/* ... and the backward half of the same pairing (r5): the backward-data pass of a convolution whose INPUT came out of a BatchNorm
 * (+ ReLU / ELU, no residual: BasicBlock norm1 -> conv2, me_resnet.py:3; the neck's norm -> conv chains and out_block -> head,
 * fcaf3d_neck_with_head.py:52-71, :257-263) also leaves that layer's two backward reductions per row block —
 * stats[blocks][2][Cout] = column sums of g' = g act'(pre) and of g' xhat (pre, xhat from the layer's input bn_x (n_out, Cout) and
 * its batch mean / var / gamma / beta / eps; act 0 none, 1 ReLU, 2 ELU) — which fc_bn_train_bwd(part = stats) consumes instead of
 * reading bn_x and g once more.  add (nullable): a second contribution to that gradient, g = result + add (the layer's output had
 * two consumers: BasicBlock's `out += residual`); bn_y (nullable): the layer's output, from which act'(.) is taken when a residual
 * was added before the activation (norm2 of a BasicBlock; NULL: act' is recomputed from bn_x). */
int fc_conv_fwd_bn_bwd_stats(const float* in, const float* W, const int* nbr, const int* out_index, float* out, int64_t n_in,
                             int64_t n_out, int K, int Cin, int Cout, int flags, void* ws, int64_t ws_bytes, float* stats,
                             const float* bn_x, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                             int act, const float* add, const float* bn_y, hipStream_t stream);
int fc_conv_fwd_pairs_tiles_bn_bwd_stats(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                                         float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int64_t live_tiles, int flags,
                                         void* ws, int64_t ws_bytes, float* stats, const float* bn_x, const float* mean,
                                         const float* var, const float* gamma, const float* beta, float eps, int act,
                                         const float* add, const float* bn_y, hipStream_t stream);

/* backward-weights of ME.MinkowskiConvolution (autograd of me_resnet.py:19-21, :56-62 and fcaf3d_neck_with_head.py:52,
 * :60-69; `gW[k] += in[i]^T (x) gout[o]`): gW[k] = sum_o in[nbr[k][o]]^T (x) gout[o];
 * deterministic two-level reduction.  flags bit24 (both entry points): the split-bf16 kernels (csrc/wgrad_x6.h: fp32 in, fp32
 * accumulate, six exact bf16 x bf16 products per fp32 product) where one exists for the shape — dense tables that qualify for
 * the multi-offset kernel, pair lists with a 128-wide Cin or Cout; the fp32 MFMA kernels otherwise. */
int64_t fc_conv_wgrad_ws_bytes(int64_t n_out, int K, int Cin, int Cout, int flags);
int fc_conv_wgrad(const float* in, const float* gout, const int* nbr, const int* row_index, float* gW, int64_t n_in,
                  int64_t n_out, int K, int Cin, int Cout, int flags, void* ws, int64_t ws_bytes, hipStream_t stream);
/* the same (autograd of me_resnet.py:56-62) over the exact pair lists of fc_kernel_map_pairs (Cin, Cout multiples of 64): the reduction skips absent
 * neighbours, which is ~40 % of a 27-offset table on surface-like scenes.  Workspace as fc_conv_wgrad_ws_bytes. */
int fc_conv_wgrad_pairs(const float* in, const float* gout, const int* pair_in, const int* pair_out, const int* pair_cnt,
                        float* gW, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int flags, void* ws,
                        int64_t ws_bytes, hipStream_t stream);

/* The stem convolution in training (me_resnet.py:19-21: `conv1` = ME.MinkowskiConvolution(3, 64, kernel_size=3, stride=2), K <= 27
 * offsets; its autograd for the weight): the forward additionally saves the gathered inputs of every output row as
 * col (n_out, 84) (27 x 3 floats, zero-padded), and the weight gradient gW (K,3,64) = col^T @ gout streams col and gout
 * instead of repeating 27 scattered 12-byte reads per output row.  col == NULL: plain forward (= fc_conv_fwd). */
int fc_stem_conv_fwd(const float* in, const float* W, const int* nbr, float* out, float* col, int64_t n_in, int64_t n_out,
                     int K, hipStream_t stream);
int64_t fc_stem_conv_wgrad_ws_bytes(int64_t n_out, int K);
int fc_stem_conv_wgrad(const float* col, const float* gout, float* gW, int64_t n_out, int K, void* ws, int64_t ws_bytes,
                       hipStream_t stream);

/* (K,Cin,Cout) -> (K,Cout,Cin): the W[k]^T of ME's backward-data rule `gin[i] += gout[o] @ W[k]^T` (SURVEY.md Appendix A.3;
 * autograd of me_resnet.py:56-62), so that the pass runs through fc_conv_fwd on the transposed table. */
int fc_transpose_weight(const float* W, float* Wt, int K, int Cin, int Cout, hipStream_t stream);

/* ---- normalisation / pooling / rows ------------------------------------------------------- */

/* column statistics per segment: MinkowskiBatchNorm (seg == NULL, one segment: all voxels on this GPU)
 * and MinkowskiInstanceNorm (seg = &coords[0], seg_stride = 4: per scene) — me_resnet.py:22,63. */
int64_t fc_col_stats_ws_bytes(int64_t n, int C, int nseg);
int fc_col_stats(const float* x, const int* seg, int seg_stride, int64_t n, int C, int nseg, float* mean, float* var,
                 float* cnt, void* ws, int64_t ws_bytes, hipStream_t stream);

/* per-segment column sums (deterministic): the per-scene loss normalisers of fcaf3d_neck_with_head.py:178-187. */
int fc_seg_col_sums(const float* x, const int* seg, int seg_stride, int64_t n, int C, int nseg, float* out, void* ws,
                    int64_t ws_bytes, hipStream_t stream);
/* nn.BatchNorm1d training-mode buffer update inside ME.MinkowskiBatchNorm (momentum, unbiased variance) —
 * me_resnet.py:48-50, BasicBlock norm1/norm2, fcaf3d_neck_with_head.py:53, :62, :68. */
int fc_bn_running_update(const float* mean, const float* var, const float* cnt, float momentum, int C, float* running_mean,
                         float* running_var, long long* num_batches_tracked, hipStream_t stream);

/* y = act((x-mean)/sqrt(var+eps)*gamma + beta (+ residual)); act 0 none, 1 ReLU, 2 ELU —
 * ME.MinkowskiInstanceNorm + MinkowskiReLU of the stem (me_resnet.py:22-23), ME.MinkowskiBatchNorm + MinkowskiELU of the
 * neck (fcaf3d_neck_with_head.py:53-54, :62-63, :68-70), BasicBlock's norm / `out += residual` / relu (Appendix A.6, A.7);
 * fc_norm_act_bwd is their autograd. */
int fc_norm_act_fwd(const float* x, const int* seg, int seg_stride, int64_t n, int C, const float* mean, const float* var,
                    float eps, const float* gamma, const float* beta, const float* residual, int act, float* y,
                    hipStream_t stream);
int64_t fc_norm_act_bwd_ws_bytes(int64_t n, int C, int nseg);
int fc_norm_act_bwd(const float* x, const float* y, const float* gy, const int* seg, int seg_stride, int64_t n, int C,
                    int nseg, const float* mean, const float* var, const float* cnt, float eps, const float* gamma,
                    const float* beta, int act, float* gx, float* gres, float* sums, void* ws, int64_t ws_bytes,
                    hipStream_t stream);

/* Training-mode ME.MinkowskiBatchNorm (+ fused ReLU/ELU / residual; the BasicBlock norms of me_resnet.py:3, :56-63 and the
 * neck norms of fcaf3d_neck_with_head.py:53, :62, :68) for small feature matrices in TWO launches per direction: batch statistics, running-buffer update (nn.BatchNorm1d momentum / unbiased variance) and apply. */
int64_t fc_bn_stats_ws_bytes(int64_t n, int C);
int fc_bn_stats_train(const float* x, int64_t n, int C, float momentum, float* mean, float* var, float* cnt,
                      float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                      int64_t ws_bytes, hipStream_t stream);
int64_t fc_bn_small_ws_bytes(int C);
int fc_bn_act_train_fwd(const float* x, int64_t n, int C, float eps, const float* gamma, const float* beta,
                        const float* residual, int act, float momentum, float* y, float* mean, float* var, float* cnt,
                        float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                        int64_t ws_bytes, hipStream_t stream);
int fc_bn_act_train_bwd(const float* x, const float* y, const float* gy, int64_t n, int C, const float* mean,
                        const float* var, float eps, const float* gamma, const float* beta, int act, float* gx, float* gres,
                        float* sums, void* ws, int64_t ws_bytes, hipStream_t stream);

/* r5 — training-mode ME.MinkowskiBatchNorm (BasicBlock norm1 / norm2 / downsample norm: me_resnet.py:3, :56-63; the neck's
 * norms: fcaf3d_neck_with_head.py:53, :62, :68) as ONE entry point per direction for every matrix size, with the two fusions
 * the launch-list executor uses:
 *   forward, `part` != NULL: the batch statistics come from column sums the PRODUCER of x left in its epilogue
 *     (fc_conv_fwd_stats / fc_conv_fwd_pairs_tiles_stats: part[nb_part][2][groups * C] = sums of x and x^2 per row block; groups = 8
 *     for the (n, 8 C) GEMM of a generative transposed convolution viewed as (8 n, C)), combined in fp64 in a fixed order — no read
 *     pass over x for the statistics; nb_part <= 64: one launch (every block of the apply kernel re-reduces the table), else two;
 *   backward, `gy2` != NULL: a second contribution to the incoming gradient, added on the fly (a tensor with two consumers:
 *     BasicBlock's `out += residual`, me_resnet.py:3) instead of a separate add pass.
 * part == NULL / gy2 == NULL: exactly fc_bn_act_train_fwd/bwd (n * C <= small_elems) or fc_bn_stats_train + fc_norm_act_fwd /
 * fc_norm_act_bwd.  sums (2, C) = [d beta, d gamma]. */
int64_t fc_bn_train_ws_bytes(int64_t n, int C);
int fc_bn_train_fwd(const float* x, int64_t n, int C, float eps, const float* gamma, const float* beta, const float* residual,
                    int act, float momentum, float* y, float* mean, float* var, float* cnt, float* running_mean,
                    float* running_var, long long* num_batches_tracked, const float* part, int64_t nb_part, int groups,
                    int64_t small_elems, void* ws, int64_t ws_bytes, hipStream_t stream);
int fc_bn_train_bwd(const float* x, const float* y, const float* gy, const float* gy2, int64_t n, int C, const float* mean,
                    const float* var, const float* cnt, float eps, const float* gamma, const float* beta, int act, float* gx,
                    float* gres, float* sums, const float* part, int64_t nb_part, int64_t small_elems, void* ws, int64_t ws_bytes,
                    hipStream_t stream);

/* ME.MinkowskiMaxPooling(k=2,s=2) — me_resnet.py:24. */
int fc_maxpool_fwd(const float* in, const int* nbr, int64_t n_out, int K, int C, float* out, int* argrow,
                   hipStream_t stream);
int fc_maxpool_bwd(const float* gout, const int* argrow, int64_t n_out, int C, float* gin, hipStream_t stream);

/* feature rows of MinkowskiPruning (fcaf3d_neck_with_head.py:124-125), of the sparse sum `inputs[i] + x` (:101) and of
 * the per-scene decomposition (:266-275); fc_scatter_rows_add is the autograd of the gather. */
int fc_gather_rows(const float* src, const int* idx, int64_t n, int C, float* dst, hipStream_t stream);
int fc_scatter_rows_add(const float* src, const int* idx, int64_t n, int C, float* dst, hipStream_t stream);

/* ---- losses ------------------------------------------------------------------------------- */

/* mmcv sigmoid_focal_loss forward/backward (through mmdet FocalLoss, fcaf3d_neck_with_head.py:29-34,:180):
 * loss_rows[n] = row_weight[n] * sum_c loss[n,c] (row_weight nullable = 1); labels int64 in {-1,0..C-1},
 * -1 = background (all classes negative).  backward: glogits = dloss/dlogits * row_weight[n] * gscale_dev[0]. */
int fc_focal_loss_fwd(const float* logits, const long long* labels, const float* row_weight, int64_t n, int C, float gamma,
                      float alpha, float* loss_rows, hipStream_t stream);
int fc_focal_loss_bwd(const float* logits, const long long* labels, const float* row_weight, int64_t n, int C, float gamma,
                      float alpha, const float* gscale_dev, float* glogits, hipStream_t stream);

/* axis-aligned 3D IoU of (n,6) [cx,cy,cz,w,l,h] boxes and its gradient w.r.t. pred —
 * iou3d_loss.py:21-35 over iou3d_calculator.py:201-330 (is_aligned=True). dpred may be NULL. */
int fc_aiou3d_fwd_bwd(const float* pred, const float* target, int target_stride, int64_t n, float eps, float* iou,
                      float* dpred, hipStream_t stream);

/* Fcaf3DNeckWithHead._loss_single (fcaf3d_neck_with_head.py:160-203) for every location of the batch in one pass, yaw-less
 * heads: sigmoid focal loss (:180), BCE-with-logits centerness loss on the positives (:191-193) and the axis-aligned
 * IoU loss on _bbox_pred_to_bbox (:281-300; iou3d_loss.py:21-35), each row weighted by its scene's 1 / (B * normaliser)
 * (inv_pos, inv_den: (B,) device arrays; scene: (n) int32), summed deterministically and scaled by the loss weights.
 * labels int64 in {-1, 0..C-1}; bbox_t (n,7) gravity-centre targets.  Backward: gradients w.r.t. cls_score (n,C),
 * centerness (n), bbox_pred (n,6) for incoming scalar gradients g_* (device, NULL = loss unused). */
int64_t fc_fcaf3d_loss_ws_bytes(int64_t n);
int fc_fcaf3d_loss_fwd(const float* points, const float* bbox_pred, const float* centerness, const float* cls_score,
                       const float* centerness_t, const float* bbox_t, const long long* labels, const int* scene,
                       const float* inv_pos, const float* inv_den, int64_t n, int n_classes, float gamma, float alpha,
                       float lw_cls, float lw_centerness, float lw_bbox, float* loss_cls, float* loss_centerness,
                       float* loss_bbox, void* ws, int64_t ws_bytes, hipStream_t stream);
int fc_fcaf3d_loss_bwd(const float* points, const float* bbox_pred, const float* centerness, const float* cls_score,
                       const float* centerness_t, const float* bbox_t, const long long* labels, const int* scene,
                       const float* inv_pos, const float* inv_den, int64_t n, int n_classes, float gamma, float alpha,
                       float lw_cls, float lw_centerness, float lw_bbox, const float* g_cls, const float* g_centerness,
                       const float* g_bbox, float* grad_cls_score, float* grad_centerness, float* grad_bbox_pred,
                       hipStream_t stream);

/* rotated 3D IoU of (n,7) [cx,cy,cz,w,l,h,yaw] boxes and its gradient w.r.t. pred — cal_iou_3d,
 * rotated_iou/oriented_iou_loss.py:86-109 + box_intersection_2d.py:13-184 + cuda_op sort_v.
 * weight (nullable): rows with weight <= 0 are skipped (iou = 0, dpred = 0). */
int fc_riou3d_fwd_bwd(const float* pred, const float* target, const float* weight, int64_t n, float* iou, float* dpred,
                      hipStream_t stream);

/* Epilogue of Fcaf3DNeckWithHead.forward_single (fcaf3d_neck_with_head.py:256-279) on the output y (n, ld <= 64) of the
 * fused 1x1 head GEMM, columns [centerness | reg (n_reg = 6 or 8) | cls (n_cls) | padding]:
 *   centerness (n,1) = y[:,0];  bbox_pred (n,n_reg) = [exp(scale * reg[:, :6]) | reg[:, 6:]];
 *   cls_score (n,n_cls) = y[:, 1+n_reg:] + bias;  cls_max (n) = max_c cls_score (the score `_prune` interpolates, :117-121).
 * bwd: gy (n, ld) from the three output gradients (any may be NULL = zero), gscale_row (n) = per-row d/dscale. */
int fc_head_split_fwd(const float* y, int ld, const float* bias, const float* scale_dev, int64_t n, int n_reg, int n_cls,
                      float* centerness, float* bbox_pred, float* cls_score, float* cls_max, hipStream_t stream);
int fc_head_split_bwd(const float* y, int ld, const float* scale_dev, const float* bbox_pred, const float* g_centerness,
                      const float* g_bbox, const float* g_cls, int64_t n, int n_reg, int n_cls, float* gy,
                      float* gscale_row, hipStream_t stream);
/* fc_head_split_bwd plus, in the same pass, the two reductions autograd would run over its outputs: gbias (n_cls) = column sums
 * of g_cls — the gradient of cls_conv's bias (fcaf3d_neck_with_head.py:262, init :91-92) — and gscale (1) = sum of the per-row
 * d/dscale terms — the gradient of Scale.scale (:273-275); either may be NULL.  Two launches (64 rows per workgroup leave
 * 65 partial sums each in ws; one 1024-thread block per column adds them in a fixed order). */
int64_t fc_head_split_bwd_sums_ws_bytes(int64_t n);
int fc_head_split_bwd_sums(const float* y, int ld, const float* scale_dev, const float* bbox_pred, const float* g_centerness,
                           const float* g_bbox, const float* g_cls, int64_t n, int n_reg, int n_cls, float* gy, float* gbias,
                           float* gscale, void* ws, int64_t ws_bytes, hipStream_t stream);

/* `cuda_ext.sort_v(vertices, mask, num_valid)` of the un-vendored Rotated_IoU extension (docker/Dockerfile:35-40),
 * bound by the reference at rotated_iou/box_intersection_2d.py:147: vertices (n_pairs,24,2) f32 centred on the mean
 * of the valid ones, mask (n_pairs,24) bool bytes, num_valid (n_pairs) i32 -> idx (n_pairs,9) i32: valid vertices in
 * angular order, closed (idx[nv] = idx[0]), padded with the first masked intersection slot.
 * fc_riou3d_fwd_bwd has this step fused in; this entry point is the drop-in for the Python-level op. */
int fc_sort_v(const float* vertices, const unsigned char* mask, const int* num_valid, int64_t n_pairs, int* idx,
              hipStream_t stream);

/* Fcaf3DAssigner.assign + compute_centerness (fcaf3d_neck_with_head.py:377-384, :394-466) for ALL scenes of the
 * batch: points (N,3) = locations of every level and scene; scene/level (N) ids; boxes (B,M,7) gravity-centre GT
 * boxes padded to M per scene, box_count (B); order (N) = rows grouped by (level, scene), seg_start (L*B+1).
 * Outputs per location: centerness target (0 for background), box target (7), label (-1 = background). */
int64_t fc_assign_ws_bytes(int B, int M, int L);
int fc_assign_targets(const float* points, const int* scene, const int* level, int64_t N, const float* boxes,
                      const long long* labels, const int* box_count, int B, int M, int L, const int* order,
                      const int* seg_start, int limit, int topk, float* centerness_t, float* bbox_t, long long* labels_out,
                      void* ws, int64_t ws_bytes, hipStream_t stream);

/* ---- NMS ---------------------------------------------------------------------------------- */

/* pcdet_nms nms_gpu (rotated=1) / nms_normal_gpu (rotated=0) — iou3d_nms.cpp:90-186, kernels
 * iou3d_nms_kernel.cu:267-372 — for `nseg` classes at once: boxes (nseg,stride,7) sorted by descending
 * score inside each segment, counts_dev (nseg) valid boxes per segment; keep (nseg,stride) receives the
 * ascending positions of the survivors, keep_count (nseg) their number.  Greedy scan runs on the device. */
int64_t fc_nms_bev_ws_bytes(int nseg, int stride);
int fc_nms_bev(const float* boxes, const int* counts_dev, int nseg, int stride, float thresh, int rotated,
               unsigned long long* mask_ws, int64_t ws_bytes, int* keep, int* keep_count, hipStream_t stream);
/* pcdet_nms boxes_iou_bev_gpu (iou3d_nms_kernel.cu:251-265): (n,m) BEV IoU matrix. */
int fc_boxes_iou_bev(const float* boxes_a, int n, const float* boxes_b, int m, int rotated, float* out,
                     hipStream_t stream);

/* ---- optimizer (the reference's recipe, configs/fcaf3d/fcaf3d.py:30-31) --------------------------------- */

/* mmcv OptimizerHook grad_clip = dict(max_norm=10, norm_type=2) (configs/fcaf3d/fcaf3d.py:31; torch.nn.utils.clip_grad_norm_)
 * over ONE flat gradient buffer of n floats (n % 4 == 0; padding zero): out[0] = the global 2-norm, out[1] = the clip
 * coefficient min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0).  Nothing is scaled here: fc_adamw_step applies
 * the coefficient while it reads the gradient.  Deterministic two-level reduction. */
int64_t fc_grad_norm_ws_bytes(int64_t n);
int fc_grad_norm(const float* g, int64_t n, float max_norm, float* out, void* ws, int64_t ws_bytes, hipStream_t stream);
/* torch.optim.AdamW.step for optimizer = dict(type='AdamW', lr=0.001, weight_decay=0.0001) (configs/fcaf3d/fcaf3d.py:30)
 * over flat parameter / gradient / moment buffers (n % 4 == 0): decoupled weight decay, bias corrections
 * 1 - beta^step computed by the caller, gradient scaled by norm_and_clip[1] (device, nullable: the output of
 * fc_grad_norm).  One pass: 16 B read + 12 B written per parameter. */
int fc_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float bias_correction1, float bias_correction2, const float* norm_and_clip,
                  hipStream_t stream);

/* SURVEY.md 8(f) rank 4 "bf16 fast mode" (No reference counterpart: the reference trains in fp32 throughout, configs/fcaf3d/fcaf3d.py:30-33
 * sets no fp16 hook).  A process-global, flagged NON-PARITY switch: while on, the split-bf16 convolution launches that read a
 * weight image (flags bits 24 | 26, 128-row tiles) and the split-bf16 weight-gradient launches multiply ONLY the leading bf16
 * piece of either operand (the operand rounded to nearest bf16; fp32 accumulate) — one MFMA product instead of six.  Default
 * off; `bench.py` reports it as `config.bf16_fast_mode` with "parity": false, never as `value`. */
int fc_set_bf16_fast(int on);

/* r6: the operand split of the matrix-pipe convolutions (No reference counterpart: MinkowskiEngine multiplies in fp32 FMA, the
 * call sites are mmdet3d/models/backbones/me_resnet.py:56-62 and every ME.MinkowskiConvolution of fcaf3d_neck_with_head.py:49-71).
 * mode 2 (default): every fp32 operand as TWO fp16 pieces, scaled by a power of two taken from the operand tensor's max |x|,
 * three exact products per fp32 product (csrc/conv_x6.h "h3": what is dropped is <= 2^-21 |a b| per product, 2^-25 on average
 * and unbiased — reductions measure closer to fp64 than an fp32 FMA chain); mode 0: three bf16 pieces, six products (r3-r5).  Process-global; weight images are built in the current mode and
 * must be read in it (rebuild them after a switch).  fc_get_split_mode: the mode launches use right now (0 while the bf16 fast
 * mode is on). */
int fc_set_split_mode(int mode);
int fc_get_split_mode(void);
/* max |x| over n floats (x 16-byte aligned) as a bit pattern -> slot[0]; slot = FC_AMAX_SLOT_BYTES (2 048) zero-initialised bytes
 * owned by the caller: 32 sub-words at a 64-byte stride whose MAXIMUM is the operand's amax (producers spread their folds over
 * them, fc_amax_out_hint); this pass writes sub-word 0, slot[1..2] are its scratch and return to zero.  Integer atomicMax: order-independent, bit-reproducible.  The mode-2 launches need
 * this word for their gathered operand(s); No reference counterpart (part of the operand split above, me_resnet.py:56-62). */
int fc_amax(const float* x, int64_t n, unsigned* slot, hipStream_t stream);
/* Hands the amax slots (device addresses; an UPPER bound of max |x| is enough) of the operands of the NEXT
 * fc_conv_fwd* / fc_conv_wgrad* call of the calling thread: amax_in for `in`, amax_gout for `gout` (weight gradients); NULL = the
 * library computes it itself with one fc_amax pass over the operand (a library-owned slot).  Cleared when that call returns.
 * No reference counterpart (see fc_set_split_mode; me_resnet.py:56-62). */
int fc_conv_amax_hint(const unsigned* amax_in, const unsigned* amax_gout);
/* The other end of the same word: the NEXT fc_bn_train_fwd / fc_bn_train_bwd / fc_norm_act_fwd / fc_norm_act_bwd / fc_bn_act_train_bwd /
 * fc_maxpool_fwd / fc_head_split_bwd_sums call of the calling thread also folds max |.| of what it writes (y, gx, out) into the slot at amax_word — one the
 * caller has ZEROED — from inside its apply kernel (one integer atomicMax per block), so that the convolution gathering that tensor
 * needs no fc_amax pass.  Cleared by that call.  No reference counterpart (see fc_set_split_mode; the producers are
 * ME.MinkowskiBatchNorm / MinkowskiReLU / MinkowskiELU / MinkowskiMaxPooling of me_resnet.py:19-24, fcaf3d_neck_with_head.py:49-71). */
int fc_amax_out_hint(unsigned* amax_word);
/* A/B switch (No reference counterpart; tests and tools/nbench): which mode-2 convolution launches on 128-row tiles run the
 * register-operand kernel csrc/conv_h3r.h — 0 none, 1 those on 128-column tiles (default), 2 all.  Results are bit-identical either
 * way (the statistics tables agree to rounding: other summation order). */
int fc_debug_set_h3r(int mode);

/* ---- launch-list executor (the network body in one call per direction) ------------------------------------ */

/* SingleStageSparse3DDetector.extract_feat (mmdet3d/models/detectors/single_stage_sparse.py:43-50: backbone me_resnet.py:43-50 +
 * Fcaf3DNeckWithHead.forward, fcaf3d_neck_with_head.py:94-108) and torch.autograd's backward over it, as a STATIC list of
 * operators walked natively: every operator is one of the entry points above with operands taken from host tables of device
 * addresses (`addr`), row counts (`dims`) and kernel-map descriptors (`maps`, fc_exec_map_words() int64 each) that the caller
 * refreshes per step; `ops` holds fc_exec_op_words() int64 per operator (layouts: fcaf3d_amd/executor.py, csrc/exec.hip).
 * Operators [op_begin, op_end) run on streams[0] (caller's), streams[1] (head branch of the neck), streams[2] (weight
 * gradients), ordered by library-owned events.  ws / ws_bytes: one scratch buffer per stream; a sizing pass runs first: if a
 * buffer is too small NOTHING is launched, ws_need[3] receives the sizes and the call returns -2.
 * cfg[0]: the two-launch BatchNorm is used up to this many elements; cfg[1]: kernel-variant flags (as `flags` above);
 * cfg[2] != 0: bracket every convolution operator with a HIP-event pair on its stream — fc_exec_probe_read(ms, meta, cap), called
 * after the device has drained, returns the number of brackets since the last read-out and writes their durations (ms) and
 * {map index | -1, direction, n_in, n_out, K, Cin, Cout, pair-list route} (8 int64 each): bench.py's live roofline measurement
 * (the reference times whole iterations only: tools/analysis_tools/benchmark.py:64-91).
 * cfg[3] != 0: the sizing pass only (ws_need is filled; 0 or -2; nothing is launched).  Events are kept per device; one caller
 * thread per device at a time. */
int fc_exec_op_words(void);
int64_t fc_exec_probe_read(float* ms, int64_t* meta, int64_t cap);
int fc_exec_map_words(void);
int fc_exec(const int64_t* ops, int64_t op_begin, int64_t op_end, const int64_t* addr, const int64_t* dims, const int64_t* maps,
            const int64_t* streams, const int64_t* ws, const int64_t* ws_bytes, int64_t* ws_need, const int64_t* cfg);

#ifdef __cplusplus
}
#endif
#endif /* FCAF3D_HIP_H */
