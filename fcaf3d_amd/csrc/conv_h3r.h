// r6: the h3 convolution (conv_x6.h MODE 2: two fp16 pieces per operand, three products) with the GATHERED operand in registers.
//
// With the matrix work halved, k_conv_x6 MODE 2 is bound by the LDS: per stage of a 128 x 128 tile the workgroup writes 16 KB of
// row pieces and 16 KB of weight pieces and its four waves read 64 KB of fragments — 656 LDS cycles against 768 matrix-pipe
// cycles per SIMD, three workgroups per CU on ONE LDS (128 x 64 tiles: 488 against 384).  Half of that traffic moves rows that only
// one wave needs once the four waves are laid out along the ROWS of the tile (each wave: 32 rows x all BN columns): an MFMA's A
// operand is "row = lane % 32, 8 consecutive reduction indices at 8 (lane / 32)", i.e. a lane can load the 32 bytes of ITS row
// straight from the feature matrix (two 16-byte loads per 16-channel block; lanes l and l + 32 share a row's 64 bytes), split them
// in registers and hand them to the matrix pipe: no LDS write, no LDS read, no barrier for the rows.  Only the weight pieces — shared
// by the four waves — go through LDS, double-buffered: ONE barrier per stage.
// Same pieces, same products in the same order per accumulator as k_conv_x6<..., MODE 2>: results are BIT-IDENTICAL
// (tests/test_gpu_h3.py::test_register_operand_kernel_is_bit_identical); same grid, pair mode, offset split, epilogues.
#pragma once

template <int BN, bool HAS_NBR, bool BUF>
__global__ __launch_bounds__(256, BN == 64 ? 4 : 3) void k_conv_h3r(
    const float* __restrict__ in, const float* __restrict__ W, const int* __restrict__ nbr,
    const int* __restrict__ out_index, const int* __restrict__ cnt, float* __restrict__ out, int64_t n_out, int K, int Cin,
    int Cout, X6Epi epi) {
  constexpr int BM = 128;
  constexpr int TN = BN / 32;                    // 32x32 tiles per wave: 32 rows x BN columns
  constexpr int NG = BN / 64;                    // 64-column groups
  constexpr int GU = 2 * 64 * 4;                 // 16-byte units of a group in LDS (two planes; the image keeps three)
  constexpr int BU = 3 * BN / 64;                // image units per thread per stage (those of plane 2 are skipped)
  __shared__ u32x4 Bs[2][NG * GU];
  __shared__ unsigned int kmask_s;
  float* __restrict__ stats = epi.stats;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int64_t bx = blockIdx.x;
  const int n0 = blockIdx.y * BN;
  int S = gridDim.z, z = blockIdx.z;
  int kbase = 0;
  if (cnt) {                                     // pair mode: see k_conv_x6
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
    kbase = z;
    out += (int64_t)z * stride * Cout;
    K = 1; S = 1; z = 0;
  }
  const int64_t m0 = bx * BM;
  const float h3_sa = h3_scale(fc_amax_read(epi.amax_in));
  const float h3_inv = h3_unscale(h3_sa, h3_scale(fc_amax_read(reinterpret_cast<const unsigned*>(W) + X6_IMG_AMAX_WORD)));

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

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
    kmask = (unsigned int)__builtin_amdgcn_readfirstlane((int)kmask_s);
  }
  constexpr unsigned X6_DEAD = 0x80000000u;

  if (kmask) {
    const int nst = __popc(kmask) * (Cin / 32);
    unsigned int rem = kmask;
    int lk = __ffs(rem) - 1;
    rem &= rem - 1;
    int lnk = rem ? __ffs(rem) - 1 : lk;
    if (rem) rem &= rem - 1;
    int lc0 = 0;
    bool sw = false;
    f32x4 av[4];                                  // this lane's row: channels 8 h + 16 b + 4 q of the stage's slab -> av[2 b + q]
    u32x4 bi[BU];
    int vcur, vnxt;
    const int rows_here = (int)((n_out - m0) < BM ? (n_out - m0) : BM);
    const int lrow = wave * 32 + r;               // this lane's row of the tile
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, 0x7fffffff, 0x00020000);
    const unsigned irow = (lrow < rows_here ? (unsigned)(m0 + lrow) : 0u) * 4u;
    auto fetch_idx = [&](int kk) -> int {
      if (BUF && HAS_NBR) return (int)__builtin_amdgcn_raw_buffer_load_b32(rn, (int)irow, (int)((unsigned)kk * (unsigned)n_out * 4u), 0);
      const int64_t row = lrow < rows_here ? m0 + lrow : 0;
      return HAS_NBR ? nbr[(int64_t)kk * n_out + row] : (int)row;
    };
    vcur = fetch_idx(lk);
    vnxt = fetch_idx(lnk);
    unsigned ocur = 0u;
    auto row_offset = [&]() {
      ocur = (vcur >= 0 && lrow < rows_here) ? (unsigned)vcur * (unsigned)(Cin * 4) + (unsigned)(h * 32) : X6_DEAD;
    };
    if (BUF) row_offset();
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)X6_DEAD, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, -1, 0x00020000);
    const u32x4* img = reinterpret_cast<const u32x4*>(W);
    const int nslab = Cin / 32, ngrp = Cout / 64;
    auto load_stage = [&]() {
      if (sw) {
        sw = false;
        lk = lnk;
        vcur = vnxt;
        if (rem) {
          lnk = __ffs(rem) - 1;
          rem &= rem - 1;
        }
        vnxt = fetch_idx(lnk);
        if (BUF) row_offset();
      }
      if (BUF) {
        const unsigned boff = (((unsigned)(kbase + lk) * (unsigned)nslab + (unsigned)(lc0 / 32)) * (unsigned)ngrp + (unsigned)(n0 / 64)) * (unsigned)(X6_GROUP_U16 * 16);
#pragma unroll
        for (int i = 0; i < BU; ++i)
          if (i % 3 != 2) bi[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, tid * 16, (int)(boff + 4096u * i), 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          av[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(ocur + (unsigned)(64 * (q >> 1) + 16 * (q & 1))), lc0 * 4, 0));
        return;
      }
      const u32x4* src = img + (((int64_t)(kbase + lk) * nslab + lc0 / 32) * ngrp + n0 / 64) * X6_GROUP_U16 + tid;
#pragma unroll
      for (int i = 0; i < BU; ++i)
        if (i % 3 != 2) bi[i] = src[256 * i];
      const bool live = vcur >= 0 && lrow < rows_here;
      const float* rowp = live ? in + (int64_t)vcur * Cin + lc0 + 8 * h : g_zero_row;
#pragma unroll
      for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f32x4*>(rowp + (live ? 16 * (q >> 1) + 4 * (q & 1) : 4 * q));
    };
    // fragments of this lane's row for the two 16-channel blocks of a stage: fa[b][plane]
    u32x4 fa[2][2];
    auto split_rows = [&]() {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        unsigned p0[4], p1[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          h3_split2s(av[2 * b + q][0], av[2 * b + q][1], h3_sa, p0[2 * q], p1[2 * q]);
          h3_split2s(av[2 * b + q][2], av[2 * b + q][3], h3_sa, p0[2 * q + 1], p1[2 * q + 1]);
        }
        fa[b][0] = u32x4{p0[0], p0[1], p0[2], p0[3]};
        fa[b][1] = u32x4{p1[0], p1[1], p1[2], p1[3]};
      }
    };
    auto store_weights = [&](int buf) {
#pragma unroll
      for (int i = 0; i < BU; ++i)
        if (i % 3 != 2) Bs[buf][(i / 3) * GU + (i % 3) * 256 + tid] = bi[i];
    };
    load_stage();
    store_weights(0);
    split_rows();
    const int swz = (r >> 2) & 3;
    int b_slot[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) b_slot[b] = r * 4 + ((2 * b + h) ^ swz);
    for (int st = 0; st < nst; ++st) {
      const int cur = st & 1;
      __syncthreads();                           // this stage's weights are in Bs[cur]; nobody reads Bs[cur ^ 1] any more
      if (st + 1 < nst) {
        lc0 += 32;
        if (lc0 >= Cin) {
          lc0 = 0;
          sw = true;
        }
      }
      load_stage();                              // (the last iteration re-reads its own stage)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int pb = 1; pb >= 0; --pb) {
          u32x4 fb[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[j] = Bs[cur][(j >> 1) * GU + pb * 256 + (j & 1) * 32 * 4 + b_slot[b]];
#pragma unroll
          for (int pa = 1 - pb; pa >= 0; --pa)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = H3_MFMA(fa[b][pa], fb[j], acc[j]);
        }
      }
      store_weights(cur ^ 1);                    // the next stage's weights (a slot nobody reads before the next barrier)
      split_rows();                              // ... and this lane's next fragments
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] *= h3_inv;

  // epilogue: lane r of sub-tile j holds column (j / 2) * 64 + 2 r + (j % 2) of the tile (x6_bslot), rows (e & 3) + 8 (e >> 2) + 4 h
  // of the wave's 32
  float* dst = out + (int64_t)z * n_out * Cout + n0 + 2 * r;
  int orow[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int64_t row = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    int o = -1;
    if (row < n_out) o = out_index ? out_index[row] : (int)row;
    orow[e] = o;
  }
  if (stats) {                                   // see k_conv_x6: the same two sums per column and row block, the same order
    const bool bwd = epi.bn_x != nullptr;
    const bool from_y = bwd && epi.bn_y != nullptr && epi.act != 0;
    float mu[TN], is[TN], ga[TN], be[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (j >> 1) * 64 + 2 * r + (j & 1);
      mu[j] = 0.f; is[j] = 1.f; ga[j] = 1.f; be[j] = 0.f;
      if (bwd) {
        mu[j] = epi.mean[col];
        is[j] = 1.f / sqrtf(epi.var[col] + epi.eps);
        ga[j] = epi.gamma ? epi.gamma[col] : 1.f;
        be[j] = epi.beta ? epi.beta[col] : 0.f;
      }
    }
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int e0 = 0; e0 < 16; e0 += 4) {
      float xv[4][TN], av2[4][TN], yv[4][TN];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = orow[e0 + u];
#pragma unroll
        for (int j = 0; j < TN; ++j) { xv[u][j] = 0.f; av2[u][j] = 0.f; yv[u][j] = 0.f; }
        if (bwd && o >= 0) {
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int64_t at = (int64_t)o * Cout + n0 + g * 64 + 2 * r;
            const f32x2 t = *reinterpret_cast<const f32x2*>(epi.bn_x + at);
            xv[u][2 * g] = t[0]; xv[u][2 * g + 1] = t[1];
            if (epi.add) { const f32x2 a2 = *reinterpret_cast<const f32x2*>(epi.add + at); av2[u][2 * g] = a2[0]; av2[u][2 * g + 1] = a2[1]; }
            if (from_y) { const f32x2 y2 = *reinterpret_cast<const f32x2*>(epi.bn_y + at); yv[u][2 * g] = y2[0]; yv[u][2 * g + 1] = y2[1]; }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (bwd && orow[e0 + u] < 0) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float t1, t2;
          x6_epi_terms(bwd, bwd ? av2[u][j] + acc[j][e0 + u] : acc[j][e0 + u], xv[u][j], mu[j], is[j], ga[j], be[j], epi.act, from_y,
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
    float* red = reinterpret_cast<float*>(&Bs[0][0]);          // [4 waves][BN][2]
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cl = (j >> 1) * 64 + 2 * r + (j & 1);
        red[(wave * BN + cl) * 2 + 0] = s1[j];
        red[(wave * BN + cl) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += red[(w * BN + tid) * 2 + 0]; b += red[(w * BN + tid) * 2 + 1]; }
      stats[((int64_t)blockIdx.x * 2 + 0) * Cout + n0 + tid] = a;
      stats[((int64_t)blockIdx.x * 2 + 1) * Cout + n0 + tid] = b;
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (orow[e] >= 0) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        f32x2 v = {acc[2 * g][e], acc[2 * g + 1][e]};
        *reinterpret_cast<f32x2*>(dst + (int64_t)orow[e] * Cout + g * 64) = v;
      }
    }
  }
}
