// fp32 sparse convolution on the bf16 matrix pipe of gfx950 by EXACT operand splitting (included by conv.hip).
//
// v_mfma_f32_32x32x2_f32 runs at 64 FLOP/clk/SIMD (157 TF), v_mfma_f32_32x32x16_bf16 at 1024 (2.5 PF).  An fp32 number
// has a 24-bit significand = three 8-bit pieces, and a bf16 holds 8 bits with the fp32 exponent range, so
//     x = x1 + x2 + x3          x1 = top 16 bits of x,  x2 = top 16 bits of (x - x1),  x3 = x - x1 - x2      (all exact)
// and a product a*b is the sum of nine bf16 x bf16 products, each EXACT in the fp32 accumulator of the MFMA (8 x 8 = 16
// bits).  The six products down to 2^-16 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1) are kept; the three dropped ones sum to at most
// 2^-24 |a b| with no common sign (r4: the pieces are cut by round-to-nearest-even: |x2| <= 2^-8 |x|, |x3| <= 2^-17 |x|;
// oracle/x6_oracle.py and its test pin these bounds; the truncating split of r3 gave 2^-21, biased towards zero)
// — in a convolution's sum below the rounding of the fp32 accumulation: against fp64 both routes sit at fp32 rounding level
// (rms error 2...5e-8 of the output scale on 27 x 64...128-term sums, which one is closer depends on the pass: tests/test_gpu_ops.py;
// tools/nbench --x6 prints it for every benchmark layer), at 6 x 32 = 192 matrix-pipe cycles per 32x32x16 block instead of
// 8 x 64 = 512.
// Non-finite inputs: +-inf splits into (inf, nan, nan), so an overflowed activation yields NaN where the fp32 pipe
// yields +-inf or NaN.
//
// Same workgroup shape, tiles, grid, pair mode, offset split and output layout as k_conv_mfma; what differs:
//  * LDS holds THREE bf16 planes of the gathered rows (BM x 32 channels x 2 B each) and of the weight slab (BN columns x
//    32 reduction indices, reduction-contiguous — the B operand of the bf16 MFMA wants 8 consecutive reduction indices
//    per lane).  The weights come either as fp32 — the (Cin, Cout) kernel transposed by the access pattern of the staging
//    loads, a thread reads 8 consecutive reduction rows of its column(s); the transposed (backward-data) kernel read
//    straight — or, the product route, as a PRE-SPLIT IMAGE (k_x6_weight_image, once per optimizer step and direction)
//    that a stage copies contiguously: every workgroup of a launch would otherwise split the same weights again
//    (r3 PMC: 6.2 VALU instructions per MFMA where the SIMD issues ~7 per MFMA slot: issue-bound, matrix pipe 44 % busy);
//  * a plane row is 64 B = four 16-byte chunks, chunk c of row r stored at slot c ^ ((r >> 2) & 3): the ds_read_b128 of
//    the MFMA fragments (lane = row r of 32, chunk 2b + lane / 32) is conflict-free in the 16-lane groups the LDS
//    serves (rows {0-3, 12-15, 20-27}: r % 4 picks the 64-byte quarter of the 256-byte bank line, (r >> 2) & 3 = {0, 3, 1, 2}
//    the chunk within it);
//  * the split of the gathered rows costs 4.5 VALU instructions per staged element (r3: 5.5 with and / sub / v_perm_b32).  It is NOT
//    hidden: r3 knock-out builds of the 441k-row 64->128 launch run 905 us compute-only = 540 (MFMA) + 367 (everything else),
//    a wave's staging VALU and another wave's MFMAs do not overlap on a SIMD; what would remove it is activation planes
//    written by the producing kernel (open item 1).
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// two fp32 -> three dwords, each the bf16 pair (piece of x0 in the low half, piece of x1 in the high half).
// Round-to-nearest-even pieces (r4; v_cvt_pk_bf16_f32 converts and packs a pair in one instruction): x1 = RN8(x),
// x2 = RN8(x - x1), x3 = x - x1 - x2 — still EXACT (the second residual has at most 7 significant bits: |x - x1| <= 2^-9 of
// x's binade top, its low end is x's last bit), with |x2| <= 2^-8 |x|, |x3| <= 2^-17 |x| and residuals of either sign: the
// three dropped products sum to <= 2^-24 |a b| and carry no common sign (the truncating split of r3, -DFC_X6_TRUNC: <= 2^-21,
// all with the sign of a b).  9 VALU per pair (3 cvt_pk, 2 shift/and pairs, 2 packed subtractions), r3: 11.
typedef __bf16 x6_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x6_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x6_hi2(unsigned u0, unsigned u1) { return __builtin_amdgcn_perm(u1, u0, 0x07060302u); }
#ifdef FC_X6_TRUNC
__device__ __forceinline__ void x6_split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  p0 = x6_hi2(u0, u1);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  p1 = x6_hi2(v0, v1);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  p2 = x6_hi2(__float_as_uint(s0), __float_as_uint(s1));
}
#else
__device__ __forceinline__ unsigned x6_rn2(float x0, float x1) {
  const x6_f32x2 v = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_bf16x2));
}
__device__ __forceinline__ void x6_split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = x6_rn2(x0, x1);
  const float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
  p1 = x6_rn2(r0, r1);
  const float s0 = r0 - __uint_as_float(p1 << 16), s1 = r1 - __uint_as_float(p1 & 0xffff0000u);
  p2 = x6_rn2(s0, s1);
}
#endif

// ---- r6: TWO fp16 pieces per operand, THREE products ("h3") -------------------------------------------------------------------
// An fp16 carries 11 significant bits.  With round-to-nearest pieces  h = RN11(s x),  l = RN11(s x - h)  the residual s x - h is
// a multiple of x's last bit with at most 12 significant bits, so the pair (h, l) holds s x EXACTLY for three values in four and
// to one fp32 ulp (2^-23 |s x|) otherwise — provided the pieces stay inside fp16's exponent range, which is what the power-of-two
// scale s is for: s = 2^(14 - floor(log2 max|x|)) over the whole operand tensor puts its largest element in [2^14, 2^15) (fp16
// overflows at 65 504), an element 2^-16 of the maximum still has both pieces normal, and below that the low piece goes subnormal
// (kept by the MFMA: scratch/h3_probe.hip) with an ABSOLUTE error <= 2^-25 = 2^-39 of the tensor's maximum.  A product
// a b = (ah + al)(bh + bl) keeps ah bh + ah bl + al bh, each EXACT in the fp32 accumulator (11 x 11 = 22 bits); dropped: al bl
// <= 2^-22 |a b| and the two representation residuals <= 2^-23 |a b| each: worst case 2^-21, on average 2^-25 of a product, no
// common sign (oracle/x6_oracle.py + its test pin these numbers).  Measured on a 1 728-term heavy-tailed reduction: 5.8e-7 of the
// result scale against 1.3e-6 for an fp32 FMA chain (rms 1.0e-7; a fourth product changes nothing), and on every benchmark layer
// closer to fp64 than the six-product route (tools/nbench).  3 x 32 = 96 matrix-pipe cycles per 32x32x16 block instead of 192,
// two planes instead of three through LDS.
// The scale travels with the data: every operand tensor has a word holding max |x| (bit pattern; fc_amax, conv.hip), weight
// images carry theirs in the unused third plane (X6_IMG_AMAX_WORD), and the kernels derive s and 1 / (sa sb) from the exponent
// fields — nothing but integer arithmetic on the host or the device decides a scale, so runs are bit-reproducible.
typedef _Float16 x6_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define X6_IMG_AMAX_WORD 2048          // dword index in a weight image: plane 2, row 0, chunk 0 of the first (k, slab, group) block —
                                       // the first of FC_AMAX_SUB sub-words at FC_AMAX_STRIDE (an amax slot, fc_common.h, inside the unused plane)
__device__ __forceinline__ float h3_scale(unsigned amax_bits) {       // 2^(14 - floor(log2 amax)), clamped to the normal range
  int se = 268 - (int)((amax_bits >> 23) & 0xffu);
  se = se > 254 ? 254 : (se < 1 ? 1 : se);
  return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ float h3_unscale(float sa, float sb) {      // 1 / (sa sb) as a power of two (clamped)
  int e = 381 - (int)(__float_as_uint(sa) >> 23) - (int)(__float_as_uint(sb) >> 23);
  e = e > 254 ? 254 : (e < 1 ? 1 : e);
  return __uint_as_float((unsigned)e << 23);
}
__device__ __forceinline__ unsigned h3_rn2(float x0, float x1) {
  const x6_f32x2 v = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_f16x2));
}
// two fp32 (already scaled) -> two dwords, each the fp16 pair (piece of x0 low half, piece of x1 high half)
__device__ __forceinline__ void h3_split2(float x0, float x1, unsigned& p0, unsigned& p1) {
  p0 = h3_rn2(x0, x1);
  const x6_f16x2 h = __builtin_bit_cast(x6_f16x2, p0);
  const float r0 = __builtin_fmaf((float)h[0], -1.0f, x0), r1 = __builtin_fmaf((float)h[1], -1.0f, x1);      // exact
  p1 = h3_rn2(r0, r1);
}

// (-DFC_H3_SPLIT_MIX, not the default — see below.)  The same two pieces from UNSCALED inputs in five instructions per pair instead of eight: v_fma_mixlo / mixhi_f16 round
// fma(x, s, -0) = s x to fp16 straight into the halves of p0, v_fma_mix_f32 forms the residual fma(x, s, -h) = s x - h exactly in
// fp32 from the fp16 half (no conversion back), one packed conversion makes p1.  Bit for bit h3_split2(s x0, s x1, ...)
// (scratch/mix_probe.hip: 4.2 M heavy-tailed values at three scales, signed zeros included).  On this SIMD a wave's VALU time adds
// to its neighbours' matrix time, and the split is most of the staging's VALU (r6_notes.md section 9).
__device__ __forceinline__ void h3_split2s(float x0, float x1, float s, unsigned& p0, unsigned& p1) {
#ifndef FC_H3_SPLIT_MIX                        // default: the conversion-based sequence.  Measured (ABAB, one box): the five-instruction
  h3_split2(x0 * s, x1 * s, p0, p1);           // mix sequence below runs the step at 476.2 / 476.4 scenes/s against 478.5 / 478.1 — like the
  return;                                      // packed fp32 instructions, VOP3P mix instructions are no bargain beside MFMAs
#endif
  unsigned h = 0u;
  const float nz = -0.0f;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(h) : "v"(x0), "v"(s), "s"(nz));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(h) : "v"(x1), "v"(s), "s"(nz));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(s), "v"(h));
  p0 = h;
  p1 = h3_rn2(r0, r1);
}

// Weight slab in LDS and in a pre-split weight image: per 64-column group three planes of 64 rows x 64 B.  Column c of the
// group lives in row (c % 2) * 32 + c / 2: a wave's sub-tile j reads rows j * 32 + r = the INTERLEAVED columns 2 r + j
// (8-byte output stores).  16-byte unit index of (group-local column c, plane, chunk):
__device__ __forceinline__ int x6_bslot(int c, int plane, int chunk) {
  const int row = (c & 1) * 32 + (c >> 1);
  return (plane * 64 + row) * 4 + (chunk ^ ((row >> 2) & 3));
}
#define X6_GROUP_U16 (3 * 64 * 4)          // 16-byte units per 64-column group (12 KB)

// Pre-split weight image of one layer and direction: [K][R / 32 slabs][C / 64 groups][X6_GROUP_U16 units], R = reduction
// size, C = columns; `transposed`: W[k] is (C, R) row-major (the layer's own kernel read as its backward-data operator),
// else (R, C).  A stage of the convolution copies 12 KB x (BN / 64) CONTIGUOUS bytes of it into LDS: no VALU, no transpose.
__device__ __forceinline__ void x6_weight_image_unit(const float* __restrict__ W, u32x4* __restrict__ img, int64_t t, int R, int C,
                                                     int transposed) {
  const int chunk = (int)(t & 3), c = (int)((t >> 2) & 63);
  const int64_t blk = t >> 8;                                               // (k, slab, group)
  const int g = (int)(blk % (C / 64));
  const int slab = (int)((blk / (C / 64)) % (R / 32));
  const int k = (int)(blk / ((int64_t)(C / 64) * (R / 32)));
  const int col = g * 64 + c, r0 = slab * 32 + chunk * 8;
  const float* Wk = W + (int64_t)k * R * C;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = transposed ? Wk[(int64_t)col * R + r0 + e] : Wk[(int64_t)(r0 + e) * C + col];
  unsigned p[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) x6_split2(x[2 * e], x[2 * e + 1], p[0][e], p[1][e], p[2][e]);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    u32x4 v = {p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
    img[blk * X6_GROUP_U16 + x6_bslot(c, pl, chunk)] = v;
  }
}

// h3 image: planes 0 / 1 hold the fp16 pieces of sw W, plane 2 is unused (its first word holds max |W|: X6_IMG_AMAX_WORD, written
// by the amax pass that runs before this one)
__device__ __forceinline__ void h3_weight_image_unit(const float* __restrict__ W, u32x4* __restrict__ img, int64_t t, int R, int C,
                                                     int transposed, float sw) {
  const int chunk = (int)(t & 3), c = (int)((t >> 2) & 63);
  const int64_t blk = t >> 8;
  const int g = (int)(blk % (C / 64));
  const int slab = (int)((blk / (C / 64)) % (R / 32));
  const int k = (int)(blk / ((int64_t)(C / 64) * (R / 32)));
  const int col = g * 64 + c, r0 = slab * 32 + chunk * 8;
  const float* Wk = W + (int64_t)k * R * C;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = transposed ? Wk[(int64_t)col * R + r0 + e] : Wk[(int64_t)(r0 + e) * C + col];
  unsigned p[2][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) h3_split2s(x[2 * e], x[2 * e + 1], sw, p[0][e], p[1][e]);
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    u32x4 v = {p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
    img[blk * X6_GROUP_U16 + x6_bslot(c, pl, chunk)] = v;
  }
}
// max |W| of the 8 weights of image unit t (the same elements the image pass reads)
__device__ __forceinline__ unsigned h3_weight_unit_amax(const float* __restrict__ W, int64_t t, int R, int C, int transposed) {
  const int chunk = (int)(t & 3), c = (int)((t >> 2) & 63);
  const int64_t blk = t >> 8;
  const int g = (int)(blk % (C / 64));
  const int slab = (int)((blk / (C / 64)) % (R / 32));
  const int k = (int)(blk / ((int64_t)(C / 64) * (R / 32)));
  const int col = g * 64 + c, r0 = slab * 32 + chunk * 8;
  const float* Wk = W + (int64_t)k * R * C;
  unsigned m = 0u;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned u = __float_as_uint(transposed ? Wk[(int64_t)col * R + r0 + e] : Wk[(int64_t)(r0 + e) * C + col]) & 0x7fffffffu;
    m = (u > m && u < 0x7f800000u) ? u : m;          // finite weights only (k_amax)
  }
  return m;
}
// a wave's maximum -> the sub-word of its block in the image's amax slot (striped, and skipped when the sub-word already holds as
// much: with ONE word per image the 4 waves x 69k blocks of the weight pass polled ~100 L2 lines and the pass took 0.62 ms)
__device__ __forceinline__ void h3_block_amax(unsigned m, unsigned* __restrict__ slot) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0) {
    unsigned* w = slot + (blockIdx.x & (FC_AMAX_SUB - 1)) * FC_AMAX_STRIDE;
    if (m > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, m);
  }
}

// MODE: 0 six bf16 products, 2 three fp16 products (amax pass: MODE 3)
template <int MODE>
__global__ void k_x6_weight_image(const float* __restrict__ W, u32x4* __restrict__ img, int K, int R, int C, int transposed) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (k, slab, group, column, chunk)
  const bool live = t < (int64_t)K * (R / 32) * (C / 64) * 256;
  if (MODE == 3) { h3_block_amax(live ? h3_weight_unit_amax(W, t, R, C, transposed) : 0u, reinterpret_cast<unsigned*>(img) + X6_IMG_AMAX_WORD); return; }
  if (MODE == 2) {
    const float sw = h3_scale(fc_amax_read(reinterpret_cast<const unsigned*>(img) + X6_IMG_AMAX_WORD));      // (every lane)
    if (live) h3_weight_image_unit(W, img, t, R, C, transposed, sw);
    return;
  }
  if (live) x6_weight_image_unit(W, img, t, R, C, transposed);
}
#if 0
__global__ void k_x6_weight_image(const float* __restrict__ W, u32x4* __restrict__ img, int K, int R, int C, int transposed) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (k, slab, group, column, chunk)
  if (t < (int64_t)K * (R / 32) * (C / 64) * 256) x6_weight_image_unit(W, img, t, R, C, transposed);
}
#endif

// Images of MANY kernels in one launch (every convolution of a model, both directions, after an optimizer step):
// desc[e] = {W, img, K, R, C, transposed, first block, -}; an entry owns the blocks [first block, first block of e + 1) and
// K (R / 32) (C / 64) of them are live (256 threads = one (k, slab, group) unit each).
// desc word 7 (h3): the address of ANOTHER image of the same weights (the forward image, for a backward-data entry) whose amax slot
// this entry shares — it skips the amax pass, scales by that slot and copies it into its own — or 0.
// r6 (last take): the image pass stages its unit's 32 x 64 weights through LDS — 16-byte loads along the rows of W as they lie
// (256 B per row of a forward kernel, 128 B per row of a transposed one), the transpose in LDS — and the amax pass reads its unit's
// 2 048 weights flat (the maximum does not care which): per-thread strided dword loads had the passes at 3.2 / 2.0 TB/s.
// (Tried and dropped: four units per thread block with the first-block column of desc in LDS, 394 -> 507 us per build — the passes
// live on blocks in flight, not on the eight scalar loads of the search.)
#define X6_IMG_LD 65
__device__ __forceinline__ void h3_weight_image_unit_lds(const float* __restrict__ W, u32x4* __restrict__ img, int64_t blk, int R, int C,
                                                         int transposed, float sw, float* __restrict__ tile /*[32][X6_IMG_LD]*/) {
  const int g = (int)(blk % (C / 64));
  const int slab = (int)((blk / (C / 64)) % (R / 32));
  const int k = (int)(blk / ((int64_t)(C / 64) * (R / 32)));
  const float* Wk = W + (int64_t)k * R * C;
  const int tid = threadIdx.x;
  if (!transposed) {                             // rows slab * 32 + r, columns g * 64 + c: 16 lanes x 16 B per row
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = p * 16 + (tid >> 4), c4 = (tid & 15) * 4;
      const float4 v = *reinterpret_cast<const float4*>(Wk + (int64_t)(slab * 32 + r) * C + g * 64 + c4);
      tile[r * X6_IMG_LD + c4] = v.x; tile[r * X6_IMG_LD + c4 + 1] = v.y; tile[r * X6_IMG_LD + c4 + 2] = v.z; tile[r * X6_IMG_LD + c4 + 3] = v.w;
    }
  } else {                                       // W[k] is (C, R): row g * 64 + c, elements slab * 32 + r: 8 lanes x 16 B per row
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int c = p * 32 + (tid >> 3), r4 = (tid & 7) * 4;
      const float4 v = *reinterpret_cast<const float4*>(Wk + (int64_t)(g * 64 + c) * R + slab * 32 + r4);
      tile[r4 * X6_IMG_LD + c] = v.x; tile[(r4 + 1) * X6_IMG_LD + c] = v.y; tile[(r4 + 2) * X6_IMG_LD + c] = v.z; tile[(r4 + 3) * X6_IMG_LD + c] = v.w;
    }
  }
  __syncthreads();
  const int chunk = tid & 3, c = (tid >> 2) & 63;
  unsigned p[2][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    h3_split2s(tile[(chunk * 8 + 2 * e) * X6_IMG_LD + c], tile[(chunk * 8 + 2 * e + 1) * X6_IMG_LD + c], sw, p[0][e], p[1][e]);
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    u32x4 v = {p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
    img[blk * X6_GROUP_U16 + x6_bslot(c, pl, chunk)] = v;
  }
}

template <int MODE>           // 0 / 2: the image pass of that mode; 3: the amax pass of h3; 4: zero the amax slots (one wave per entry)
__global__ __launch_bounds__(256) void k_x6_weight_images(const long long* __restrict__ desc, int n) {
  if (MODE == 4) {
    const int e = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (e < n && l < FC_AMAX_SUB) reinterpret_cast<unsigned*>(desc[8 * e + 1])[X6_IMG_AMAX_WORD + l * FC_AMAX_STRIDE] = 0u;
    return;
  }
  int lo = 0, hi = n - 1;
  const long long b = blockIdx.x;
  while (lo < hi) {                                                         // last entry whose first block <= b
    const int mid = (lo + hi + 1) >> 1;
    if (desc[8 * mid + 6] <= b) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + 8 * lo;
  const int K = (int)d[2], R = (int)d[3], C = (int)d[4];
  const int64_t t = (b - d[6]) * 256 + threadIdx.x;
  const bool live = t < (int64_t)K * (R / 32) * (C / 64) * 256;            // (block-uniform: a unit is 256 threads)
  const float* W = reinterpret_cast<const float*>(d[0]);
  u32x4* img = reinterpret_cast<u32x4*>(d[1]);
  const bool vec = (reinterpret_cast<uintptr_t>(W) & 15) == 0;
  if (MODE == 3) {
    if (d[7]) return;                           // shares its sibling's slot
    unsigned m = 0u;
    if (live && vec) {                          // the unit's 256 x 8 weights, flat
      const float4* w4 = reinterpret_cast<const float4*>(W + t * 8);
      const float4 x0 = w4[0], x1 = w4[1];
      const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned v = __float_as_uint(xs[e]) & 0x7fffffffu;
        m = (v > m && v < 0x7f800000u) ? v : m;  // finite weights only (k_amax)
      }
    } else if (live) {
      m = h3_weight_unit_amax(W, t, R, C, (int)d[5]);
    }
    h3_block_amax(m, reinterpret_cast<unsigned*>(img) + X6_IMG_AMAX_WORD);
    return;
  }
  if (MODE == 2) {
    __shared__ float tile[32 * X6_IMG_LD];
    const unsigned* slot = d[7] ? reinterpret_cast<const unsigned*>(d[7]) + X6_IMG_AMAX_WORD : reinterpret_cast<const unsigned*>(img) + X6_IMG_AMAX_WORD;
    const unsigned am = fc_amax_read(slot);     // (every lane)
    if (d[7] && b == d[6] && threadIdx.x == 0) reinterpret_cast<unsigned*>(img)[X6_IMG_AMAX_WORD] = am;      // own slot: sub-word 0 (the rest stays zero)
    if (!live) return;
    if (vec) h3_weight_image_unit_lds(W, img, t >> 8, R, C, (int)d[5], h3_scale(am), tile);
    else h3_weight_image_unit(W, img, t, R, C, (int)d[5], h3_scale(am));
    return;
  }
  if (live) x6_weight_image_unit(W, img, t, R, C, (int)d[5]);
}

#define X6_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)
#define H3_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)

// BSRC: where the weight slab comes from — 0: fp32 (Cin, Cout) kernel, split in the staging; 1: the same kernel read
// transposed (backward data on the layer's own weights); 2: a pre-split image of k_x6_weight_image (straight copy).
// r5: what the epilogue of a launch leaves for the BatchNorm next to it, besides the result (conv.hip / norm.hip):
//   stats != NULL, bn_x == NULL: column sums of the result and of its square per row block (forward: the batch statistics of the
//                                BatchNorm that FOLLOWS the convolution);
//   stats != NULL, bn_x != NULL: the launch is a backward-data pass whose result g is the gradient arriving at a BatchNorm (+ ReLU /
//                                ELU) layer with input bn_x: column sums of g' = g act'(pre) and of g' xhat per row block — the two
//                                reductions of that layer's backward pass (norm.hip k_norm_bwd_partial), without its read pass.
//                                add (nullable): a second contribution to that gradient, g = result + add (the layer's output had two
//                                consumers); bn_y (nullable): the layer's OUTPUT, for act'(.) where a residual was added before the
//                                activation (BasicBlock norm2: act' cannot be recomputed from bn_x alone).
struct X6Epi {
  float* stats;
  const float* bn_x;
  const float* mean;
  const float* var;
  const float* gamma;
  const float* beta;
  float eps;
  int act;
  const float* add;
  const float* bn_y;
  const unsigned* amax_in;       // h3 launches: the amax SLOT (fc_common.h fc_amax_read) of the gathered operand
};

// the two terms an element contributes to the statistics table (see X6Epi); mu / is / ga / be: the channel's parameters
// (yv: the layer's output where from_y, else unused; v already includes `add`)
__device__ __forceinline__ void x6_epi_terms(bool bwd, float v, float xv, float mu, float is, float ga, float be, int act, bool from_y,
                                             float yv, float& t1, float& t2) {
  if (!bwd) {
    t1 = v;
    t2 = v * v;
  } else {
    float d = 1.f;
    if (from_y) {                                              // norm.hip act_bwd_from_y
      if (act == 1) d = yv > 0.f ? 1.f : 0.f;
      else if (act == 2) d = yv > 0.f ? 1.f : yv + 1.f;
    } else {
      const float pre = fmaf((xv - mu) * is, ga, be);          // norm.hip bn_pre: the same instruction sequence
      if (act == 1) d = pre > 0.f ? 1.f : 0.f;
      else if (act == 2) d = pre > 0.f ? 1.f : expf(pre);
    }
    const float g = v * d;
    t1 = g;
    t2 = g * ((xv - mu) * is);
  }
}

// FAST (r5, SURVEY.md 8(f) rank 4 "bf16 fast mode" — a flagged NON-PARITY extra, never the benchmark's `value`): only plane 0 of
// either operand — the operand rounded to nearest bf16 — is staged and multiplied: one MFMA per 32x32x16 block instead of six,
// a third of the staged bytes.  Weight-image launches only; a compile-time variant (as a run-time branch it cost the product
// kernel 92 spilled registers).
// BUF (r5; weight-image launches whose gathered operand is smaller than 2 GB): the gathered rows and the image units are read with
// buffer loads — a lane's row is a 32-bit byte offset computed once per kernel offset (an absent neighbour: an offset past the
// descriptor's range, the load returns zeros), the channel slab and the image's stage are SCALAR offsets — instead of 64-bit
// per-lane addresses rebuilt every stage: the stage loop of the flat-address kernel issues ~3.4 VALU and 1.5 scalar instructions
// per MFMA (profiles/r5_conv_pmc.md) of which the address arithmetic is a third.  Same loads, same values: bit-identical.
// MODE (r6; was the bool FAST): 0 six bf16 products, 1 FAST, 2 h3 — three fp16 products on two planes (weight-image launches only)
template <int BM, int BN, bool HAS_NBR, int WM, int BSRC, int MODE = 0, bool BUF = false>
__global__ __launch_bounds__(256, BM == 256 ? 2 : (BN == 64 ? 4 : 3)) void k_conv_x6(
    const float* __restrict__ in, const float* __restrict__ W, const int* __restrict__ nbr,
    const int* __restrict__ out_index, const int* __restrict__ cnt, float* __restrict__ out, int64_t n_out, int K, int Cin,
    int Cout, X6Epi epi) {
  float* __restrict__ stats = epi.stats;
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);      // 32x32 MFMA tiles per wave
  constexpr int RW = BM / WM;                                  // rows of a wave's part of the tile
  constexpr int AR = BM / 32;                                  // float4 gathers per thread per stage (8 threads per row)
  constexpr int CPT = BN / 64;                                 // weight columns per thread per stage (BSRC 0 / 1)
  constexpr int BU = 3 * BN / 64;                              // 16-byte image units per thread per stage (BSRC 2)
  constexpr bool WT = BSRC == 1;
  constexpr bool FAST = MODE == 1;
  constexpr bool H3 = MODE == 2 && BSRC == 2;
  constexpr int NPL = H3 ? 2 : 3;                              // planes staged per operand
  constexpr int GU = NPL * 64 * 4;                             // 16-byte units of a 64-column group in LDS (the IMAGE keeps 3 planes)
  __shared__ u32x4 As[NPL * BM * 4];                           // [plane][row][4 chunks]
  __shared__ u32x4 Bs[(BN / 64) * GU];                         // [64-column group][plane][row][4 chunks]: x6_bslot
  __shared__ unsigned int kmask_s;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = WM == 4 ? wave : wave >> 1, wc = WM == 4 ? 0 : wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // this wave's weight rows: TN == 2: both halves (j) of group wc; TN == 1 (128 x 64 tile, 2 x 2 waves): half wc of group 0
  const int bgrp = TN == 2 ? wc : 0;
  int64_t bx = blockIdx.x;
  const int n0 = blockIdx.y * BN;
  int S = gridDim.z, z = blockIdx.z;
  int kbase = 0;                                 // pair mode: the kernel offset this workgroup works on (image addressing)
  if (cnt) {                                     // pair mode: see k_conv_mfma
    if (gridDim.z == 1 && K > 1) {
      int k = 0;
      for (; k < K - 1; ++k) {
        const int64_t t = ((int64_t)cnt[k] + BM - 1) / BM;
        if (bx < t) break;
        bx -= t;
      }
      z = k;
    }
    const int64_t stride = n_out;
    n_out = cnt[z];
    if (bx * BM >= n_out) return;
    nbr += (int64_t)z * stride;
    if (BSRC == 2) kbase = z; else W += (int64_t)z * Cin * Cout;
    out += (int64_t)z * stride * Cout;
    K = 1; S = 1; z = 0;
  }
  const int64_t m0 = bx * BM;
  const int a_c4 = tid & 7, a_r = tid >> 3;      // A staging: 8 float4 per gathered 32-channel row slab, 32 rows per pass
  float h3_sa = 1.f, h3_inv = 1.f;               // h3: scale of the gathered operand, 1 / (sa sw) for the epilogue
  if (H3) {
    h3_sa = h3_scale(fc_amax_read(epi.amax_in));
    h3_inv = h3_unscale(h3_sa, h3_scale(fc_amax_read(reinterpret_cast<const unsigned*>(W) + X6_IMG_AMAX_WORD)));
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- (1) which of this split's offsets have a neighbour anywhere in the tile ---------------------
  unsigned int kmask;
  {
    if (tid == 0) kmask_s = 0u;
    __syncthreads();
    if (tid < BM) {
      unsigned int mk = 0u;
      int64_t row = m0 + tid;
      if (row < n_out) {
        if (HAS_NBR) {
          for (int k = z; k < K; k += S)
            if (nbr[(int64_t)k * n_out + row] >= 0) mk |= 1u << k;
        } else {
          mk = 1u;
        }
      }
      for (int off = 32; off > 0; off >>= 1) mk |= __shfl_xor(mk, off, 64);
      if (lane == 0 && mk) atomicOr(&kmask_s, mk);
    }
    __syncthreads();
    kmask = BUF ? (unsigned int)__builtin_amdgcn_readfirstlane((int)kmask_s) : kmask_s;
  }
  constexpr bool buf = BUF && BSRC == 2;
  constexpr unsigned X6_DEAD = 0x80000000u;      // BUF: the descriptor of the gathered operand ends here

  // ---- (2) software-pipelined stage loop -------------------------------------------------------------
  if (kmask) {
    const int nst = __popc(kmask) * (Cin / 32);           // stages of this tile
    unsigned int rem = kmask;
    int lk = __ffs(rem) - 1;                              // load cursor: offset / channel slab of the stage requested next
    rem &= rem - 1;
    int lnk = rem ? __ffs(rem) - 1 : lk;                  // ... and the offset after it (its rows are already on their way)
    if (rem) rem &= rem - 1;
    int lc0 = 0;
    bool sw = false;
    constexpr bool fast = FAST && BSRC == 2;
    f32x4 av[AR];
    f32x4 bw[2 * CPT];                                    // BSRC 1: 2 float4 (8 reduction-consecutive floats) per column
    float bf[8][CPT];                                     // BSRC 0: 8 reduction rows x CPT adjacent columns
    u32x4 bi[BU];                                         // BSRC 2: image units
    int vcur[AR], vnxt[AR];
    // tile rows past the end read the table's first row and are masked at the gather (their index is forced to -1)
    const int rows_here = (int)((n_out - m0) < BM ? (n_out - m0) : BM);
    // BUF: the neighbour table through a buffer descriptor too — a lane's row is a constant 32-bit byte offset, the kernel offset a
    // SCALAR one: no 64-bit address per row and offset (hipcc speculated that arithmetic into every stage: ~20 of ~74 VALU)
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, 0x7fffffff, 0x00020000);
    unsigned irow[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) irow[i] = (a_r + 32 * i < rows_here ? (unsigned)(m0 + a_r + 32 * i) : 0u) * 4u;
    auto fetch_idx = [&](int kk, int (&v)[AR]) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        if (BUF && BSRC == 2 && HAS_NBR) {
          v[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rn, (int)irow[i], (int)((unsigned)kk * (unsigned)n_out * 4u), 0);
          continue;
        }
        const int lr = a_r + 32 * i;
        const int64_t row = lr < rows_here ? m0 + lr : 0;
        v[i] = HAS_NBR ? nbr[(int64_t)kk * n_out + row] : (int)row;
      }
    };
    fetch_idx(lk, vcur);
    fetch_idx(lnk, vnxt);
    // BUF: byte offset of this lane's 16 bytes in the rows of the current kernel offset (channel slab 0)
    unsigned ocur[AR];
    auto row_offsets = [&]() {
#pragma unroll
      for (int i = 0; i < AR; ++i)
        ocur[i] = (vcur[i] >= 0 && a_r + 32 * i < rows_here) ? (unsigned)vcur[i] * (unsigned)(Cin * 4) + (unsigned)(a_c4 * 16) : X6_DEAD;
    };
    if (buf) row_offsets();
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)X6_DEAD, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, -1, 0x00020000);
    // weight staging roles.  BSRC 0: thread = (column group g = tid % 64 -> columns g * CPT .. + CPT - 1, chunk tid / 64);
    // BSRC 1: thread = (column tid / 4 + 64 u, chunk tid % 4); BSRC 2: units tid + 256 i of the stage's contiguous image
    const int b_g = WT ? (tid >> 2) : (tid & 63), b_c = WT ? (tid & 3) : (tid >> 6);
    const u32x4* img = reinterpret_cast<const u32x4*>(W);
    const int nslab = Cin / 32, ngrp = Cout / 64;
    auto load_stage = [&]() {
      if (sw) {
        sw = false;
        lk = lnk;
#pragma unroll
        for (int i = 0; i < AR; ++i) vcur[i] = vnxt[i];
        if (rem) {
          lnk = __ffs(rem) - 1;
          rem &= rem - 1;
        }
        fetch_idx(lnk, vnxt);
        if (buf) row_offsets();
      }
      if (buf) {
        const unsigned boff = (((unsigned)(kbase + lk) * (unsigned)nslab + (unsigned)(lc0 / 32)) * (unsigned)ngrp + (unsigned)(n0 / 64)) * (unsigned)(X6_GROUP_U16 * 16);
#pragma unroll
        for (int i = 0; i < BU; ++i)
          if ((!fast || i % 3 == 0) && !(H3 && i % 3 == 2)) bi[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, tid * 16, (int)(boff + 4096u * i), 0);
#pragma unroll
        for (int i = 0; i < AR; ++i)
          av[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)ocur[i], lc0 * 4, 0));
        return;
      }
      if (BSRC == 2) {
        const u32x4* src = img + (((int64_t)(kbase + lk) * nslab + lc0 / 32) * ngrp + n0 / 64) * X6_GROUP_U16 + tid;
#pragma unroll
        for (int i = 0; i < BU; ++i)
          if ((!fast || i % 3 == 0) && !(H3 && i % 3 == 2)) bi[i] = src[256 * i];          // (unit tid + 256 i lies in plane i % 3 of its 64-column group)
      } else if (WT) {
        const float* Wk = W + (int64_t)lk * Cin * Cout;
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          const float* src = Wk + (int64_t)(n0 + b_g + 64 * u) * Cin + lc0 + 8 * b_c;
          bw[2 * u] = *reinterpret_cast<const f32x4*>(src);
          bw[2 * u + 1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
      } else {
        const float* Wk = W + (int64_t)lk * Cin * Cout;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float* src = Wk + (int64_t)(lc0 + 8 * b_c + e) * Cout + n0 + b_g * CPT;
          if (CPT == 2) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(src);
            bf[e][0] = t[0];
            bf[e][CPT - 1] = t[1];
          } else {
            bf[e][0] = *src;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool live = vcur[i] >= 0 && a_r + 32 * i < rows_here;
        const float* src = live ? in + (int64_t)vcur[i] * Cin + lc0 + a_c4 * 4 : g_zero_row + a_c4 * 4;
        av[i] = *reinterpret_cast<const f32x4*>(src);
      }
    };
    load_stage();
    // wave priorities (tools/nbench --prio): unlike the fp32 kernels this one LOSES 15-20 % with s_setprio 1 around the MFMA
    // block (r3: 441k rows 64->128 1259 -> 1083 us without) — the waves that are splitting / storing the next stage need
    // the VALU slots between a multiplying wave's MFMAs.  Default: none; 1: MFMA block (the fp32 kernels' scheme); 2: staging.
    const int pmode = buf ? 0 : g_fc_prio;      // (BUF: no run-time priority switch — five branches per stage)
    // fragment slots of this lane (16-byte units): every row this lane reads is r + a multiple of 32, so the chunk
    // swizzle is (r >> 2) & 3 throughout
    const int swz = (r >> 2) & 3;
    int a_slot[2], b_slot[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      a_slot[b] = (wr * RW + r) * 4 + ((2 * b + h) ^ swz);
      b_slot[b] = bgrp * GU + ((TN == 2 ? 0 : wc * 32) + r) * 4 + ((2 * b + h) ^ swz);
    }
    for (int st = 0; st < nst; ++st) {
      __syncthreads();                           // previous stage fully consumed
      if (pmode == 2) __builtin_amdgcn_s_setprio(1);
      // ---- split + store the gathered rows: 4 channels -> 8 B per plane
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int row = a_r + 32 * i;
        unsigned p[3][2];
        if (H3) {
          h3_split2s(av[i][0], av[i][1], h3_sa, p[0][0], p[1][0]);
          h3_split2s(av[i][2], av[i][3], h3_sa, p[0][1], p[1][1]);
        } else {
          x6_split2(av[i][0], av[i][1], p[0][0], p[1][0], p[2][0]);
          x6_split2(av[i][2], av[i][3], p[0][1], p[1][1], p[2][1]);
        }
        const int slot = row * 4 + ((a_c4 >> 1) ^ ((row >> 2) & 3));
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          if (fast && pl) break;
          u32x2 v = {p[pl][0], p[pl][1]};
          reinterpret_cast<u32x2*>(As + pl * BM * 4 + slot)[a_c4 & 1] = v;
        }
      }
      // ---- the weight slab: image units as they are, or split 8 reduction indices of one column -> 16 B per plane
      if (BSRC == 2) {
#pragma unroll
        for (int i = 0; i < BU; ++i)
          if ((!fast || i % 3 == 0) && !(H3 && i % 3 == 2)) Bs[H3 ? (i / 3) * GU + (i % 3) * 256 + tid : tid + 256 * i] = bi[i];
      } else {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = WT ? bw[2 * u + e / 4][e % 4] : bf[e][u];
          const int cl = WT ? b_g + 64 * u : b_g * CPT + u;          // column of the tile
          unsigned p[3][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x6_split2(x[2 * e], x[2 * e + 1], p[0][e], p[1][e], p[2][e]);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            u32x4 v = {p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
            Bs[(cl >> 6) * X6_GROUP_U16 + x6_bslot(cl & 63, pl, b_c)] = v;
          }
        }
      }
      if (pmode == 2) __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      if (st + 1 < nst) {
        lc0 += 32;
        if (lc0 >= Cin) {
          lc0 = 0;
          sw = true;
        }
      }
      load_stage();                              // (the last iteration re-reads its own stage: see k_conv_mfma_p)
      // BUF: the next stage's loads stay in front of this stage's MFMAs (with fewer address registers in the way hipcc sinks them
      // behind the last MFMA, a step from their use: 378.6 -> 379.9 scenes/s, three interleaved pairs)
      if (buf) __builtin_amdgcn_sched_barrier(0);
      // per 16-channel block: the three planes of the rows, then the weight planes one at a time, smallest products first
      // (a1b3 | a2b2 a1b2 | a3b1 a2b1 a1b1): 32 fragment registers live instead of 48
      if (pmode == 1) __builtin_amdgcn_s_setprio(1);
      if (fast) {                                  // one product per block: a1 b1
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          u32x4 fa1[TM], fb1[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) fa1[i] = As[a_slot[b] + i * 32 * 4];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb1[j] = Bs[b_slot[b] + j * 32 * 4];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = X6_MFMA(fa1[i], fb1[j], acc[i][j]);
        }
      } else if (H3) {                             // three products per block, smallest first: ah bl | al bh, ah bh
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          u32x4 fa[2][TM];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[pl][i] = As[pl * BM * 4 + a_slot[b] + i * 32 * 4];
#pragma unroll
          for (int pb = 1; pb >= 0; --pb) {
            u32x4 fb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = Bs[pb * 64 * 4 + b_slot[b] + j * 32 * 4];
#pragma unroll
            for (int pa = 1 - pb; pa >= 0; --pa)
#pragma unroll
              for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = H3_MFMA(fa[pa][i], fb[j], acc[i][j]);
          }
        }
      } else
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        u32x4 fa[3][TM];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[pl][i] = As[pl * BM * 4 + a_slot[b] + i * 32 * 4];
#pragma unroll
        for (int pb = 2; pb >= 0; --pb) {
          u32x4 fb[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[j] = Bs[pb * 64 * 4 + b_slot[b] + j * 32 * 4];
#pragma unroll
          for (int pa = 2 - pb; pa >= 0; --pa)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = X6_MFMA(fa[pa][i], fb[j], acc[i][j]);
        }
      }
      if (pmode == 1) __builtin_amdgcn_s_setprio(0);
    }
  }
  if (H3) {                                      // back to the operands' own scale (a power of two: exact)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] *= h3_inv;
  }
  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); lane r of sub-tile j holds
  // column 2 r + j of the wave's 64-column group (TN == 1: j = wc)
  float* dst = out + (int64_t)z * n_out * Cout + n0 + bgrp * 64 + 2 * r + (TN == 2 ? 0 : wc);
  int orow[TM][16];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = m0 + wr * RW + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      int o = -1;
      if (row < n_out) o = out_index ? out_index[row] : (int)row;
      orow[i][e] = o;
    }
  if (stats) {
    // r5: BatchNorm statistics out of the epilogue (unsplit neighbour-table / dense launches only: the accumulators ARE the
    // results): per column the two sums of X6Epi over this tile's rows -> stats[tile][2][Cout] (forward: rows past the end and
    // absent neighbours hold exact zeros; backward: they are skipped).  Lane: 16 TM values per column; the other half-wave holds
    // the same columns' other rows; the WM waves along the rows meet in LDS (the stage buffers are free now).  Fixed order:
    // deterministic.
    const bool bwd = epi.bn_x != nullptr;
    const bool from_y = bwd && epi.bn_y != nullptr && epi.act != 0;
    const int col0 = n0 + bgrp * 64 + 2 * r + (TN == 2 ? 0 : wc);          // this lane's first column (TN == 2: and col0 + 1)
    float mu[TN], is[TN], ga[TN], be[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      mu[j] = 0.f; is[j] = 1.f; ga[j] = 1.f; be[j] = 0.f;
      if (bwd) {
        mu[j] = epi.mean[col0 + j];
        is[j] = 1.f / sqrtf(epi.var[col0 + j] + epi.eps);
        ga[j] = epi.gamma ? epi.gamma[col0 + j] : 1.f;
        be[j] = epi.beta ? epi.beta[col0 + j] : 0.f;
      }
    }
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += 4) {
        float xv[4][TN], av[4][TN], yv[4][TN];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                       // four rows of the layer's input (second gradient, output) in flight
          const int o = orow[i][e0 + u];
#pragma unroll
          for (int j = 0; j < TN; ++j) { xv[u][j] = 0.f; av[u][j] = 0.f; yv[u][j] = 0.f; }
          if (bwd && o >= 0) {
            const int64_t at = (int64_t)o * Cout + col0;
            if (TN == 2) {
              const f32x2 t = *reinterpret_cast<const f32x2*>(epi.bn_x + at);
              xv[u][0] = t[0]; xv[u][TN - 1] = t[1];
              if (epi.add) { const f32x2 a2 = *reinterpret_cast<const f32x2*>(epi.add + at); av[u][0] = a2[0]; av[u][TN - 1] = a2[1]; }
              if (from_y) { const f32x2 y2 = *reinterpret_cast<const f32x2*>(epi.bn_y + at); yv[u][0] = y2[0]; yv[u][TN - 1] = y2[1]; }
            } else {
              xv[u][0] = epi.bn_x[at];
              if (epi.add) av[u][0] = epi.add[at];
              if (from_y) yv[u][0] = epi.bn_y[at];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (bwd && orow[i][e0 + u] < 0) continue;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float t1, t2;
            x6_epi_terms(bwd, bwd ? av[u][j] + acc[i][j][e0 + u] : acc[i][j][e0 + u], xv[u][j], mu[j], is[j], ga[j], be[j], epi.act, from_y,
                         yv[u][j], t1, t2);
            s1[j] += t1;
            s2[j] += t2;
          }
        }
      }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(As);                 // [WM][BN][2]
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cl = bgrp * 64 + 2 * r + (TN == 2 ? j : wc);
        red[(wr * BN + cl) * 2 + 0] = s1[j];
        red[(wr * BN + cl) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { a += red[(w * BN + tid) * 2 + 0]; b += red[(w * BN + tid) * 2 + 1]; }
      stats[((int64_t)blockIdx.x * 2 + 0) * Cout + n0 + tid] = a;
      stats[((int64_t)blockIdx.x * 2 + 1) * Cout + n0 + tid] = b;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (orow[i][e] >= 0) {
        if (TN == 2) {
          f32x2 v = {acc[i][0][e], acc[i][TN - 1][e]};
          *reinterpret_cast<f32x2*>(dst + (int64_t)orow[i][e] * Cout) = v;
        } else {
          dst[(int64_t)orow[i][e] * Cout] = acc[i][0][e];
        }
      }
    }
}

// (r4's two staging experiments — the gathered operand as pre-split bf16 planes, k_conv_x6p, and planes + weight image by LDS-DMA,
// k_conv_x6d: bit-identical, +4...30 % per launch, about what writing the planes costs — were removed in r5; numbers and the
// reasoning in profiles/r4_notes.md sections 5 and 10, profiles/r5_notes.md.)
