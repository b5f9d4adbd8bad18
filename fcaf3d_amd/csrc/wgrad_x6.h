// fp32 weight gradient of the sparse convolution on the bf16 matrix pipe by exact operand splitting (see conv_x6.h for
// the arithmetic; included by conv.hip).
//
//     gW[k][ci][co] = sum_o  in[nbr[k][o]][ci] * gout[o][co]
//
// GEMM with M = Cin, N = Cout and the REDUCTION over rows, so both bf16 MFMA operands want 8 consecutive ROWS of one
// channel per lane, while memory holds rows of channels.  The transposition happens in the staging registers: wave w of the
// workgroup owns rows 8 w .. 8 w + 7 of a 32-row stage, lane l channel l of a 64-channel pass; it loads its channel of those
// 8 rows (8 dword loads, each a fully coalesced 256-byte row segment across the wave — the row indices are wave-uniform and
// travel in scalar registers), splits the 8 values into three bf16 pieces each and stores ONE 16-byte chunk per plane:
// LDS image [plane][channel][4 chunks], chunk c = rows 8 c .. 8 c + 7, slot c ^ ((channel >> 2) & 3) — the image and the
// fragment reads of conv_x6.h with "channel" in the place of "row".
// KO kernel offsets per workgroup share the staged gout rows (dense tables; k_wgrad_multi's idea), PAIRS: the reduction
// runs over the exact pair list of ONE offset (both operands gathered).  Partial sums per row range go to `part` and are
// combined by k_wgrad_reduce in a fixed order, exactly as for the fp32 kernels.
#pragma once

// Knock-out builds (tools/knockout.sh; never in the product library): what the stage loop costs without one of its parts.
//   -DFC_KO_WG_NOMFMA: no matrix instructions (the fragment reads stay: their words are xor-ed into the result)
//   -DFC_KO_WG_NOSPLIT: the three pieces are copies of the fp32 bits' top half (no split arithmetic)
//   -DFC_KO_WG_NOLOAD: no row loads (register constants)      -DFC_KO_WG_NOIDX: row indices by arithmetic, not from the table
#ifdef FC_KO_WG_NOSPLIT
#define WG_SPLIT2(x0, x1, p0, p1, p2) do { p0 = x6_hi2(__float_as_uint(x0), __float_as_uint(x1)); p1 = p0; p2 = p0; } while (0)
#else
#define WG_SPLIT2(x0, x1, p0, p1, p2) x6_split2(x0, x1, p0, p1, p2)
#endif

// (the r3 kernel k_wgrad_x6 — the transposition in the staging registers described above, 40 global_load_dword per thread and
// stage — was removed in r5: k_wgrad_x6t below is bit-identical and 1.3-1.6x faster, profiles/r4_notes.md section 9.)

// ---- r4: rows loaded 16 bytes per lane, transposed by the LDS read ------------------------------------------------------------
// Knock-out builds of k_wgrad_x6 on the 441k-row maps (tools/knockout.sh, tools/nbench): without its MFMAs it runs 5 % faster,
// without the split arithmetic 7-11 %, without its ROW LOADS 2-2.6x — and an XCD-aware block map (the nine offset groups of a
// row range as consecutive residents of ONE XCD, sharing its L2: 1108 / 1841 / 2181 us without, 1129 / 1834 / 2180 with; removed)
// changes nothing: the kernel is bound by the load INSTRUCTIONS of its register transposition
// (lane = channel: 40 global_load_dword per thread and 32-row stage, 256 B per wave instruction), not by bytes or MFMAs.
// Here a thread loads 4 consecutive channels of a row (global_load_dwordx4: 8 lanes = 32 channels of a row, a wave = 8 rows, the
// workgroup = one [32 rows][32 channels] subtile per pass; 10 loads per thread and stage for 64 -> 128 with 3 offsets), splits
// adjacent channels and stores 8 bytes per plane into ROW-MAJOR bf16 subtiles (row stride 64 B, conflict-free 128-byte write
// groups); the MFMA fragments — 8 consecutive ROWS of one channel per lane — come out of ds_read_b64_tr_b16, gfx950's
// transposing LDS read: a 16-lane group supplies the addresses of a [4 rows][16 channels] block (4 lanes per row, 8 B each) and
// lane i receives channel i of the four rows; two of them (rows +0..3, +4..7) make one operand.  The 32 lanes the LDS serves per
// cycle read 4 rows x 64 B = 256 contiguous bytes.  Same pieces, same products in the same order, same k positions: results
// are bit-identical to k_wgrad_x6.
typedef short x6_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 x6_tr_frag(const u32x2* base, int unit) {
  // rows +0..3 and +4..7 of this lane's 8-row group: 4 rows = 256 B = 32 units apart
  const auto p = (__attribute__((address_space(3))) x6_s16x4*)(base + unit);
  const x6_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  const x6_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p + 32);
  const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
  u32x4 v = {a[0], a[1], b[0], b[1]};
  return v;
}

// MODE (r6; was the bool FAST): 0 six bf16 products, 1 FAST (plane 0 only), 2 h3: two fp16 pieces per operand, three products
// (conv_x6.h); amax_a / amax_g: the words holding max |in| / max |gout| (h3 only)
template <int BMc, int BNc, int KO, bool PAIRS, int MODE = 0>
__global__ __launch_bounds__(256, (KO * (BMc / 64) * (BNc / 64) >= 6) ? 2 : 3) void k_wgrad_x6t(
    const float* __restrict__ in, const float* __restrict__ gout, const int* __restrict__ nbr,
    const int* __restrict__ row_index, const int* __restrict__ cnt, float* __restrict__ part, int64_t n_out, int K, int Cin,
    int Cout, int64_t rows_per_split, const unsigned* __restrict__ amax_a, const unsigned* __restrict__ amax_g, int buf) {
  constexpr int TM = BMc / 64, TN = BNc / 64;    // 32x32 tiles per wave (waves 2 x 2 over the BMc x BNc tile)
  constexpr int SA = BMc / 32, SG = BNc / 32;    // [32 rows][32 channels] subtiles per offset / of gout: 256 units of 8 B each
  constexpr bool FAST = MODE == 1, H3 = MODE == 2;
  constexpr int NPL = H3 ? 2 : 3;
  __shared__ u32x2 As[KO * NPL * SA * 256];      // [offset][plane][channel block][row][8 units]
  __shared__ u32x2 Gs[NPL * SG * 256];           // [plane][channel block][row][8 units]
  float h3_sa = 1.f, h3_sg = 1.f, h3_inv = 1.f;
  if (H3) {
    h3_sa = h3_scale(fc_amax_read(amax_a));
    h3_sg = h3_scale(fc_amax_read(amax_g));
    h3_inv = h3_unscale(h3_sa, h3_sg);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_n = Cout / BNc, tiles_m = Cin / BMc;
  int y = blockIdx.y;
  const int tn = y % tiles_n; y /= tiles_n;
  const int tm = y % tiles_m; y /= tiles_m;
  const int k0 = y * KO;
  const int ci0 = tm * BMc, co0 = tn * BNc;
  int64_t total = n_out;
  if (PAIRS) {
    total = cnt[k0];
    rows_per_split = ((total + gridDim.x - 1) / gridDim.x + 31) / 32 * 32;
  }
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_split;
  int64_t r_end = r_begin + rows_per_split;
  if (r_end > total) r_end = total;

  f32x16 acc[KO][TM][TN];
#pragma unroll
  for (int o = 0; o < KO; ++o)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[o][i][j][e] = 0.f;

  if (r_begin < r_end) {
    // staging role: row 8 wave + lane / 8 of the stage, channels 4 (lane % 8) .. + 3 of every 32-channel block
    const int s_row = 8 * wave + (lane >> 3), s_c4 = (lane & 7) * 4;
    const int s_unit = s_row * 8 + (lane & 7);
    int ia[KO], ig;
    // buf: the index tables through buffer descriptors as well (row -> 32-bit byte offset, kernel offset -> scalar offset)
    const __amdgpu_buffer_rsrc_t rnb = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(row_index), 0, 0x7fffffff, 0x00020000);
    auto fetch_idx = [&](int64_t rb) {
      const int64_t row = rb + s_row;
      const int64_t rc = row < r_end ? row : r_end - 1;
#pragma unroll
      for (int o = 0; o < KO; ++o) {
        const int kk = k0 + o < K ? k0 + o : K - 1;
        int t;
        if (buf && nbr) t = (int)__builtin_amdgcn_raw_buffer_load_b32(rnb, (int)((unsigned)rc * 4u), (int)((unsigned)kk * (unsigned)n_out * 4u), 0);
        else t = nbr ? nbr[(int64_t)kk * n_out + rc] : (int)rc;        // no table: the dense GEMM over the rows themselves
        ia[o] = (row < r_end && k0 + o < K) ? t : -1;
      }
      if (PAIRS && buf) ig = row < r_end ? (int)__builtin_amdgcn_raw_buffer_load_b32(rrx, (int)((unsigned)rc * 4u), (int)((unsigned)k0 * (unsigned)n_out * 4u), 0) : -1;
      else ig = row < r_end ? (PAIRS ? row_index[(int64_t)k0 * n_out + rc] : (int)rc) : -1;
    };
    f32x4 av[KO][SA], gv[SG];
    // buf (r6; both operands below 2 GB, fewer than 2^24 rows): rows through buffer descriptors — a 32-bit byte offset per row (24-bit
    // multiply-add), an absent row = an offset past the descriptor's range (the load returns zeros) — instead of a 64-bit address
    // per row and stage (v_mad_u64_u32 + v_lshl_add_u64 per operand: ~30 of the stage's ~200 VALU instructions, and on this SIMD a
    // wave's VALU time ADDS to the other wave's matrix time: 2 x (1 152 + 840) clocks per stage measured as 4 037)
    constexpr unsigned WG_DEAD = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gout), 0, (int)WG_DEAD, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)WG_DEAD, 0x00020000);
    const unsigned g_row_bytes = (unsigned)Cout * 4u, a_row_bytes = (unsigned)Cin * 4u;
    const unsigned g_col_bytes = (unsigned)(co0 + s_c4) * 4u, a_col_bytes = (unsigned)(ci0 + s_c4) * 4u;
    auto load_rows = [&]() {                     // rows of the stage whose indices sit in ia / ig
      if (buf) {
        const unsigned og = ig >= 0 ? __umul24((unsigned)ig, g_row_bytes) + g_col_bytes : WG_DEAD;
#pragma unroll
        for (int p = 0; p < SG; ++p) gv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(og + 128u * p), 0, 0));
#pragma unroll
        for (int o = 0; o < KO; ++o) {
          const unsigned oa = ia[o] >= 0 ? __umul24((unsigned)ia[o], a_row_bytes) + a_col_bytes : WG_DEAD;
#pragma unroll
          for (int p = 0; p < SA; ++p) av[o][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, (int)(oa + 128u * p), 0, 0));
        }
        return;
      }
      const float* gp = ig >= 0 ? gout + (int64_t)ig * Cout + co0 + s_c4 : g_zero_row + s_c4;
#pragma unroll
      for (int p = 0; p < SG; ++p) {
#ifdef FC_KO_WG_NOLOAD
        gv[p] = f32x4{(float)lane, 1.f, 2.f, (float)p};
#else
        gv[p] = *reinterpret_cast<const f32x4*>(gp + (ig >= 0 ? p * 32 : 0));
#endif
      }
#pragma unroll
      for (int o = 0; o < KO; ++o) {
        const float* ap = ia[o] >= 0 ? in + (int64_t)ia[o] * Cin + ci0 + s_c4 : g_zero_row + s_c4;
#pragma unroll
        for (int p = 0; p < SA; ++p) {
#ifdef FC_KO_WG_NOLOAD
          av[o][p] = f32x4{(float)lane, 1.f, (float)o, (float)p};
#else
          av[o][p] = *reinterpret_cast<const f32x4*>(ap + (ia[o] >= 0 ? p * 32 : 0));
#endif
        }
      }
    };
    fetch_idx(r_begin);
    load_rows();
    // fragment role: 16-lane group g = lane / 16 -> channels (g & 1) 16 .. + 15 of a 32-channel tile, rows 8 (g >> 1) .. + 7 of a
    // 16-row block; lane i of the group ADDRESSES row i / 4, channels 4 (i % 4) .. + 3 and RECEIVES channel i
    const int fi = lane & 15, fg = lane >> 4;
    const int f_unit = (8 * (fg >> 1) + (fi >> 2)) * 8 + (fg & 1) * 4 + (fi & 3);
    constexpr bool fast = FAST;                  // bf16 fast mode (flagged non-parity extra, conv_x6.h): plane 0 only
    for (int64_t rb = r_begin; rb < r_end; rb += 32) {
      fetch_idx(rb + 32 < r_end ? rb + 32 : rb);                 // (past the end the last stage is re-read and never used)
      __syncthreads();
#pragma unroll
      for (int p = 0; p < SG; ++p) {
        unsigned q[3][2];
        if (H3) {
          h3_split2s(gv[p][0], gv[p][1], h3_sg, q[0][0], q[1][0]);
          h3_split2s(gv[p][2], gv[p][3], h3_sg, q[0][1], q[1][1]);
        } else {
          WG_SPLIT2(gv[p][0], gv[p][1], q[0][0], q[1][0], q[2][0]);
          WG_SPLIT2(gv[p][2], gv[p][3], q[0][1], q[1][1], q[2][1]);
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          if (fast && pl) break;
          u32x2 v = {q[pl][0], q[pl][1]};
          Gs[(pl * SG + p) * 256 + s_unit] = v;
        }
      }
#pragma unroll
      for (int o = 0; o < KO; ++o)
#pragma unroll
        for (int p = 0; p < SA; ++p) {
          unsigned q[3][2];
          if (H3) {
            h3_split2s(av[o][p][0], av[o][p][1], h3_sa, q[0][0], q[1][0]);
            h3_split2s(av[o][p][2], av[o][p][3], h3_sa, q[0][1], q[1][1]);
          } else {
            WG_SPLIT2(av[o][p][0], av[o][p][1], q[0][0], q[1][0], q[2][0]);
            WG_SPLIT2(av[o][p][2], av[o][p][3], q[0][1], q[1][1], q[2][1]);
          }
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            if (fast && pl) break;
            u32x2 v = {q[pl][0], q[pl][1]};
            As[((o * NPL + pl) * SA + p) * 256 + s_unit] = v;
          }
        }
      __syncthreads();
      load_rows();
      if (fast) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          u32x4 fb1[TN], fa1[KO][TM];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb1[j] = x6_tr_frag(Gs, (wc * TN + j) * 256 + b * 128 + f_unit);
#pragma unroll
          for (int o = 0; o < KO; ++o)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa1[o][i] = x6_tr_frag(As, ((o * NPL) * SA + wr * TM + i) * 256 + b * 128 + f_unit);
#pragma unroll
          for (int o = 0; o < KO; ++o)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[o][i][j] = X6_MFMA(fa1[o][i], fb1[j], acc[o][i][j]);
        }
      } else if (H3) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          u32x4 fb[2][TN];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[pl][j] = x6_tr_frag(Gs, (pl * SG + wc * TN + j) * 256 + b * 128 + f_unit);
          u32x4 fa[KO][2][TM];
#pragma unroll
          for (int o = 0; o < KO; ++o)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
              for (int i = 0; i < TM; ++i) fa[o][pl][i] = x6_tr_frag(As, ((o * 2 + pl) * SA + wr * TM + i) * 256 + b * 128 + f_unit);
#pragma unroll
          for (int pb = 1; pb >= 0; --pb)
#pragma unroll
            for (int pa = 1 - pb; pa >= 0; --pa)
#pragma unroll
              for (int o = 0; o < KO; ++o)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                  for (int j = 0; j < TN; ++j) acc[o][i][j] = H3_MFMA(fa[o][pa][i], fb[pb][j], acc[o][i][j]);
        }
      } else
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        u32x4 fb[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[pl][j] = x6_tr_frag(Gs, (pl * SG + wc * TN + j) * 256 + b * 128 + f_unit);
        u32x4 fa[KO][3][TM];
#pragma unroll
        for (int o = 0; o < KO; ++o)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[o][pl][i] = x6_tr_frag(As, ((o * 3 + pl) * SA + wr * TM + i) * 256 + b * 128 + f_unit);
#pragma unroll
        for (int pb = 2; pb >= 0; --pb)
#pragma unroll
          for (int pa = 2 - pb; pa >= 0; --pa)
#pragma unroll
            for (int o = 0; o < KO; ++o)
#pragma unroll
              for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#ifdef FC_KO_WG_NOMFMA
                  acc[o][i][j][0] += __uint_as_float((fa[o][pa][i][0] ^ fb[pb][j][0]) & 0x3fffffu);
#else
                  acc[o][i][j] = X6_MFMA(fa[o][pa][i], fb[pb][j], acc[o][i][j]);
#endif
                }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < KO; ++o) {
    if (k0 + o >= K) break;
    float* dst = part + ((int64_t)blockIdx.x * K + k0 + o) * Cin * Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = ci0 + wr * (BMc / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int col = co0 + wc * (BNc / 2) + j * 32 + r;
          dst[(int64_t)row * Cout + col] = H3 ? acc[o][i][j][e] * h3_inv : acc[o][i][j][e];
        }
  }
}
