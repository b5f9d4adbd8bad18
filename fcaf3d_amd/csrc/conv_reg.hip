// Register-direct sparse convolution kernels for gfx950 (no LDS staging, no per-stage barriers).
//
// fp32 MFMA runs at the fp32 VECTOR rate (v_mfma_f32_32x32x2_f32: 64 cycles per instruction per SIMD,
// MI355X_MICROARCH.md), i.e. one 64x64x8 step of a wave (16 MFMAs) keeps its SIMD's matrix pipe busy for 1024
// cycles while needing only 4 KB of operands.  At that ratio the operands can come straight from L1/L2 into the
// MFMA fragment registers: every wavefront is an independent stream
//        [prefetch unit u+3 into registers]  ->  [16 MFMAs of unit u]
// with no LDS round trip and no workgroup barrier in the main loop, so waves never wait for each other and a wave
// may skip the MFMAs of a 32-row sub-tile whose rows have no neighbour at the current kernel offset (wave-uniform
// branch) — the granularity the LDS-tiled kernel (conv.hip) cannot have, because its four waves meet at two
// barriers per stage.
//
// Fragment mapping (v_mfma_f32_32x32x2_f32, lane = (r = lane & 31, h = lane >> 5)):
//   A: lane (r,h) loads ONE float4 of gathered row r: channels 8q + 4h + {0..3}; MFMA e of the unit uses element e,
//      i.e. it reduces over the channel pair {8q + e, 8q + 4 + e}.
//   B: lane (c,h) loads, for e = 0..3, TN consecutive floats W[k][8q + 4h + e][n0 + TN*c .. +TN-1]; sub-tile j of the
//      wave therefore owns the INTERLEAVED columns n0 + TN*c + j, which makes the B loads and the output stores
//      8/16-byte vectors (a column permutation of the output tile is free: the epilogue undoes it by address).
// Replaces MinkowskiEngine's ConvolutionForward/Backward called from me_resnet.py:19-21,56-62, BasicBlock,
// fcaf3d_neck_with_head.py:52,60-69,83-85,257-263 (same contract as conv.hip's k_conv_mfma).
#include <type_traits>
#include "fc_common.h"
#include "conv_reg.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) float g_zero_row_reg[64];   // what an absent neighbour gathers from

#define FC_TRACE_DEFINE
#include "fc_trace.h"

template <int N> struct VecT;
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<4> { typedef f32x4 T; };

// WSPLIT = true : the 4 waves of a workgroup share ONE (TM*32 x TN*32) output tile and split its reduction
//                 (the flattened (active offset, 8-channel step) unit list, contiguous quarter each); partial tiles
//                 are summed through LDS in a fixed order.  Smaller per-CU working set, finishes a tile 4x sooner.
// WSPLIT = false: every wave owns its own tile (workgroup = 2x2 tiles); no LDS, no barrier at all.
template <int TM, int TN, bool HAS_NBR, bool WSPLIT>
__global__ __launch_bounds__(256, (TM * TN <= 4) ? 3 : 2) void k_conv_reg(
    const float* __restrict__ in, const float* __restrict__ W, const int* __restrict__ nbr,
    const int* __restrict__ out_index, const int* __restrict__ cnt, float* __restrict__ out, int64_t n_out, int K,
    int Cin, int Cout) {
  static_assert(TM == 1 || TM == 2, "a tile is at most 64 rows: lane <-> row in the prologue");
  typedef typename VecT<TN>::T vecn;
  constexpr int ROWS = TM * 32, COLS = TN * 32, NV = TM * 16;
  // dynamic LDS: [WSPLIT: 2 reduction slabs of NV*64 vecn] [neighbour-index tables: slots x 64 ints, one per workgroup
  // (WSPLIT) or one per wave] [WSPLIT: 4 x TM mask words]
  extern __shared__ __attribute__((aligned(16))) float red[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform values stay in SGPRs
  TR_DECL;
  TR(0);
  const int r = lane & 31, h = lane >> 5;
  int S = gridDim.z, z = blockIdx.z;
  if (cnt) {
    // pair mode (fc_conv_fwd_pairs): grid.z = kernel offset, `nbr` row z = input rows of that offset's cnt[z] pairs;
    // the tile computes T_z[j] = in[pair_in[z][j]] @ W[z] for the compacted rows j into slab z of the workspace.
    const int64_t stride = n_out;
    n_out = cnt[z];
    nbr += (int64_t)z * stride;
    W += (int64_t)z * Cin * Cout;
    out += (int64_t)z * stride * Cout;
    K = 1; S = 1; z = 0;
  }
  int64_t m0;
  int n0;
  if (WSPLIT) {
    m0 = (int64_t)blockIdx.x * ROWS;
    n0 = blockIdx.y * COLS;
    if (m0 >= n_out) return;                    // workgroup-uniform
  } else {
    m0 = ((int64_t)blockIdx.x * 2 + (wave >> 1)) * ROWS;
    n0 = (blockIdx.y * 2 + (wave & 1)) * COLS;
    if (m0 >= n_out || n0 >= Cout) return;      // wave-uniform, no barrier in this mode
  }
  const int slots = (K - z + S - 1) / S;        // offsets of this split: k = z + S * slot
  int* tab = reinterpret_cast<int*>(red) + (WSPLIT ? 2 * NV * 64 * TN : wave * slots * 64);
  unsigned int* msk = reinterpret_cast<unsigned int*>(tab + slots * 64);      // WSPLIT only

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- prologue: neighbour rows of the tile -> LDS table; which offsets have a neighbour in each 32-row sub-tile
  //      (wave-uniform bit masks over SLOTS).  lane <-> tile row; WSPLIT: wave w handles slots w, w+4, ... --------
  unsigned int smask[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) smask[i] = 0u;
  if (HAS_NBR) {
    const int64_t row = m0 + (TM == 1 ? r : lane);
    const bool rok = row < n_out;
    const int64_t rowc = rok ? row : n_out - 1;
    constexpr int BATCH = 8;
    const int sstep = WSPLIT ? 4 : 1;
    for (int s0 = WSPLIT ? wave : 0; s0 < slots; s0 += BATCH * sstep) {
      int t[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int sl = s0 + u * sstep;
        const int slc = sl < slots ? sl : slots - 1;
        t[u] = nbr[(int64_t)(z + S * slc) * n_out + rowc];
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int sl = s0 + u * sstep;
        if (sl < slots) {
          const int v = rok ? t[u] : -1;
          const unsigned long long b = __ballot(v >= 0);
          if (TM == 1) {
            if ((unsigned int)b) smask[0] |= 1u << sl;
            if (lane < 32) tab[sl * 64 + lane] = v;
          } else {
            if ((unsigned int)b) smask[0] |= 1u << sl;
            if ((unsigned int)(b >> 32)) smask[TM - 1] |= 1u << sl;
            tab[sl * 64 + lane] = v;
          }
        }
      }
    }
    if (WSPLIT) {
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) msk[wave * TM + i] = smask[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < TM; ++i) smask[i] = msk[i] | msk[TM + i] | msk[2 * TM + i] | msk[3 * TM + i];
    } else {
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) smask[i] = __builtin_amdgcn_readfirstlane(smask[i]);
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) smask[i] = (m0 + i * 32 < n_out) ? 1u : 0u;
  }
  unsigned int tmask = 0u;
#pragma unroll
  for (int i = 0; i < TM; ++i) tmask |= smask[i];
  TR(1);

  // ---- this wave's units: (active slot a, 8-channel step q), q fastest ---------------------------------
  const int NQ = Cin >> 3;
  const int U = __popc(tmask) * NQ;
  const int u0 = WSPLIT ? (wave * U) >> 2 : 0;
  const int u1 = WSPLIT ? ((wave + 1) * U) >> 2 : U;
  const int n = u1 - u0;

  if (n > 0) {
    auto fetch_idx = [&](int sl, int (&dstv)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (HAS_NBR) dstv[i] = tab[sl * 64 + i * 32 + r];
        else dstv[i] = (m0 + i * 32 + r < n_out) ? (int)(m0 + i * 32 + r) : -1;
      }
    };
    // load cursor (all scalar)
    unsigned int rem = tmask;
    for (int a = u0 / NQ; a > 0; --a) rem &= rem - 1;
    int Ls = __ffs(rem) - 1;                        // current slot
    rem &= rem - 1;
    int Lq = u0 % NQ;
    int Lns = rem ? __ffs(rem) - 1 : Ls;            // next active slot
    rem &= rem - 1;
    int left = n;                                   // units not yet loaded (the cursor saturates on the last one)
    int idx[TM], idxn[TM];
    fetch_idx(Ls, idx);
    fetch_idx(Lns, idxn);

    struct Buf {
      f32x4 a[TM];
      vecn b[4];
      int s;
    };
    auto load = [&](Buf& B) {
      const int cc = Lq * 8 + 4 * h;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float* src = idx[i] < 0 ? g_zero_row_reg + 4 * h : in + (int64_t)idx[i] * Cin + cc;
        B.a[i] = *reinterpret_cast<const f32x4*>(src);
      }
      const float* wp = W + ((int64_t)(z + S * Ls) * Cin + cc) * Cout + n0 + TN * r;
#pragma unroll
      for (int e = 0; e < 4; ++e) B.b[e] = *reinterpret_cast<const vecn*>(wp + (int64_t)e * Cout);
      B.s = Ls;
      if (left > 1) {
        --left;
        if (++Lq == NQ) {
          Lq = 0;
          Ls = Lns;
#pragma unroll
          for (int i = 0; i < TM; ++i) idx[i] = idxn[i];
          if (rem) {
            Lns = __ffs(rem) - 1;
            rem &= rem - 1;
            fetch_idx(Lns, idxn);                  // LDS read (lgkmcnt): does not touch the global-load queue
          }
        }
      }
    };
    auto compute = [&](const Buf& B) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if ((smask[i] >> B.s) & 1u) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(B.a[i][e], B.b[e][j], acc[i][j], 0, 0, 0);
        }
      }
    };
    Buf b0, b1, b2;
    load(b0);
    load(b1);
    load(b2);
#ifdef FC_TRACE
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       // first unit's operands have landed
    TR(2);
#endif
    for (int t = 0; t < n; t += 3) {
      compute(b0);
      load(b0);
      if (t + 1 < n) compute(b1);
      load(b1);
      if (t + 2 < n) compute(b2);
      load(b2);
    }
  }

  TR(3);
  float* dst = out + (int64_t)z * n_out * Cout;
  // output row of accumulator element (i, e) of this lane (C/D layout of the 32x32 MFMA), through out_index when the
  // rows are processed in occupancy-mask order; -1 = beyond the end.  Looked up in one batch ahead of the stores.
  auto out_row = [&](int i, int e) -> int {
    const int64_t row = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (row >= n_out) return -1;
    return out_index ? out_index[row] : (int)row;
  };
  auto store_v = [&](int i, int e, int row) {
    if (row >= 0) {
      vecn v;
#pragma unroll
      for (int j = 0; j < TN; ++j) v[j] = acc[i][j][e];
      *reinterpret_cast<vecn*>(dst + (int64_t)row * Cout + n0 + TN * r) = v;
    }
  };
  int orow[TM][16];
  if (!WSPLIT) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) orow[i][e] = out_row(i, e);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) store_v(i, e, orow[i][e]);
    TR(5);
    TR_FLUSH(n);
    return;
  }
  // ---- fixed-order sum of the four partial tiles: (w0 + w2) + (w1 + w3), each wave finishing half of the rows ----
  vecn* slab = reinterpret_cast<vecn*>(red);
  constexpr int SL = NV * 64;                              // vecn per slab
  auto put = [&](int s, int i, int e) {
    vecn v;
#pragma unroll
    for (int j = 0; j < TN; ++j) v[j] = acc[i][j][e];
    slab[s * SL + (i * 16 + e) * 64 + lane] = v;
  };
  auto add = [&](int s, int i, int e) {
    const vecn v = slab[s * SL + (i * 16 + e) * 64 + lane];
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j][e] += v[j];
  };
  if (wave >= 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) put(wave - 2, i, e);
  }
  __syncthreads();
  if (wave < 2) {
    // wave 0 finishes vectors [0, NV/2), wave 1 finishes [NV/2, NV): look their output rows up now
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bool mine = ((i * 16 + e) < NV / 2) == (wave == 0);
        orow[i][e] = mine ? out_row(i, e) : -1;
      }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) add(wave, i, e);
  }
  __syncthreads();
  if (wave < 2) {                                          // ... and hand the other half over
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bool mine = ((i * 16 + e) < NV / 2) == (wave == 0);
        if (!mine) put(wave, i, e);
      }
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bool mine = ((i * 16 + e) < NV / 2) == (wave == 0);
        if (mine) {
          add(1 - wave, i, e);
          store_v(i, e, orow[i][e]);
        }
      }
  }
  TR(5);
  TR_FLUSH(n);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient, register-direct:  gW[k][ci][co] = sum over the pairs p of offset k of in[pin[p]][ci] * gout[pout[p]][co]
// v_mfma_f32_32x32x2_f32 with M = ci, N = co and the reduction over PAIRS (2 per instruction): lane (m,h) loads
// in[pin[2s+h]][ci0 + 2m .. +1] and gout[pout[2s+h]][co0 + 2m .. +1] (8-byte loads, 32 lanes = one 256-byte row
// segment; sub-tile i/j owns the interleaved channels ci0 + 2m + i / co0 + 2m + j).  A wave owns a 64x64 tile of
// gW[k] and a range of pairs; the four waves of a workgroup share the tile and split the range, their partial tiles
// are summed through LDS in a fixed order, workgroups of different ranges write partials that k_wgrad_reduce
// (conv.hip) sums in a fixed order.
// PAIRS = true : pin/pout are the exact pair lists (K, n_out) with cnt[k] valid entries.
// PAIRS = false: dense table: pair p of offset k = (nbr[k][p], p) for every output row p (absent -> zero row);
//                nbr == NULL: identity (in row p, out row p).
#define WG_CHUNK 16                      // pairs per software-pipeline stage (8 MFMA steps)
template <bool PAIRS, bool HAS_NBR>
__global__ __launch_bounds__(256, 3) void k_wgrad_reg(const float* __restrict__ in, const float* __restrict__ gout,
                                                       const int* __restrict__ pin, const int* __restrict__ pout,
                                                       const int* __restrict__ cnt, float* __restrict__ part,
                                                       int64_t n_out, int K, int Cin, int Cout, int64_t rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // 2 slabs of 32*64 f32x2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int tiles_n = Cout / 64, tiles_m = Cin / 64;
  int y = blockIdx.y;
  const int tn = y % tiles_n; y /= tiles_n;
  const int tm = y % tiles_m; y /= tiles_m;
  const int k = y;
  const int ci0 = tm * 64, co0 = tn * 64;
  int64_t total = n_out;
  if (PAIRS) {
    total = cnt[k];
    rows_per_split = ((total + gridDim.x - 1) / gridDim.x + 4 * WG_CHUNK - 1) / (4 * WG_CHUNK) * (4 * WG_CHUNK);
  }
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_split;
  int64_t r_end = r_begin + rows_per_split;
  if (r_end > total) r_end = total;
  // this wave's quarter of the range, in whole chunks
  const int64_t len = r_end > r_begin ? r_end - r_begin : 0;
  const int64_t chunks = (len + WG_CHUNK - 1) / WG_CHUNK;
  const int64_t c_lo = (chunks * wave) >> 2, c_hi = (chunks * (wave + 1)) >> 2;
  const int64_t w_begin = r_begin + c_lo * WG_CHUNK;
  int64_t w_end = r_begin + c_hi * WG_CHUNK;
  if (w_end > r_end) w_end = r_end;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (w_begin < w_end) {
    const int* pin_k = HAS_NBR ? pin + (int64_t)k * n_out : nullptr;
    const int* pout_k = PAIRS ? pout + (int64_t)k * n_out : nullptr;
    struct Buf {
      f32x2 a[WG_CHUNK / 2], g[WG_CHUNK / 2];
    };
    // lane l < 16 of the wave holds the (input row, output row) of pair (chunk base + l); every MFMA step broadcasts its
    // two pairs with readlane.  The indices of the NEXT chunk are requested at the end of load(), a whole compute phase
    // before they are needed, so the wait in front of the readlanes never stalls on them.
    int vi, vo;
    int64_t nbase = w_begin;
    auto fetch = [&]() {
      const int64_t p = nbase + (lane & (WG_CHUNK - 1));
      const int64_t pc = p < w_end ? p : w_end - 1;
      vi = HAS_NBR ? pin_k[pc] : (int)pc;
      vo = PAIRS ? pout_k[pc] : (int)pc;
      if (p >= w_end) vi = -1;
    };
    fetch();
    auto load = [&](Buf& B) {
#pragma unroll
      for (int s = 0; s < WG_CHUNK / 2; ++s) {
        const int i0 = __builtin_amdgcn_readlane(vi, 2 * s), i1 = __builtin_amdgcn_readlane(vi, 2 * s + 1);
        const int o0 = __builtin_amdgcn_readlane(vo, 2 * s), o1 = __builtin_amdgcn_readlane(vo, 2 * s + 1);
        const int ii = h ? i1 : i0, oo = h ? o1 : o0;
        const float* ap = ii < 0 ? g_zero_row_reg + 2 * m : in + (int64_t)ii * Cin + ci0 + 2 * m;
        const float* gp = ii < 0 ? g_zero_row_reg + 2 * m : gout + (int64_t)oo * Cout + co0 + 2 * m;
        B.a[s] = *reinterpret_cast<const f32x2*>(ap);
        B.g[s] = *reinterpret_cast<const f32x2*>(gp);
      }
      nbase += WG_CHUNK;
      fetch();
    };
    auto compute = [&](const Buf& B) {
#pragma unroll
      for (int s = 0; s < WG_CHUNK / 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(B.a[s][i], B.g[s][j], acc[i][j], 0, 0, 0);
    };
    Buf b0, b1;
    load(b0);
    // loads are unconditional (a chunk past the end reads the zero row): conditionally assigned staging registers end up
    // in scratch (r1 finding)
    for (int64_t base = w_begin; base < w_end; base += 2 * WG_CHUNK) {
      load(b1);
      compute(b0);
      load(b0);
      if (base + WG_CHUNK < w_end) compute(b1);
    }
  }

  // ---- fixed-order sum of the four partial tiles (as k_conv_reg) and store: gW rows ci, interleaved channel map ----
  f32x2* slab = reinterpret_cast<f32x2*>(red);
  constexpr int SL = 32 * 64;
  auto put = [&](int s, int i, int e) {
    f32x2 v = {acc[i][0][e], acc[i][1][e]};
    slab[s * SL + (i * 16 + e) * 64 + lane] = v;
  };
  auto add = [&](int s, int i, int e) {
    const f32x2 v = slab[s * SL + (i * 16 + e) * 64 + lane];
    acc[i][0][e] += v[0];
    acc[i][1][e] += v[1];
  };
  float* dst = part + ((int64_t)blockIdx.x * K + k) * Cin * Cout;
  auto store_v = [&](int i, int e) {
    // C layout: row (= A index m') = (e&3) + 8*(e>>2) + 4h within the 32x32 tile, col (= B index) = lane & 31
    const int mi = (e & 3) + 8 * (e >> 2) + 4 * h;
    const int ci = ci0 + 2 * mi + i;
    f32x2 v = {acc[i][0][e], acc[i][1][e]};
    *reinterpret_cast<f32x2*>(dst + (int64_t)ci * Cout + co0 + 2 * m) = v;
  };
  if (wave >= 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) put(wave - 2, i, e);
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) add(wave, i, e);
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if ((i == 0) != (wave == 0)) put(wave, i, e);
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if ((i == 0) == (wave == 0)) {
          add(1 - wave, i, e);
          store_v(i, e);
        }
  }
}

// ------------------------------------------------------------------------------------------------
// Streaming variant: PERSISTENT waves.  The per-tile fixed costs of the kernels above (neighbour scan, first operand
// round trip, cross-wave reduction: 10-20 us per tile on a loaded chip, tools/nbench_trace) rival the MFMA time of
// a small tile, so here a wave never starts over: it walks a static list of work items
//     item = (offset split z, 32*TM-row tile, 32*TN-column tile),   item i -> wave (i mod #waves)
// as ONE continuous stream of 32-channel "quads" (4 units of 8 channels).  Three cursors walk the same sequence:
//     P (two quads ahead)  requests neighbour rows / output rows / group masks of what comes next,
//     N (one quad ahead)   issues the A/W operand loads into the 4 unit buffers just freed,
//     C                    multiplies, and at the last quad of an item stores the tile and clears the accumulators,
// so operand loads of the next item are in flight while the current one finishes: no prologue, no drain.
// Which offsets a 32-row group needs comes from a precomputed table (fc_nbr_group_masks: one word per group), rows
// are walked in occupancy-mask order (out_index) so that groups share their absent offsets, and every wave owns its
// tile (no LDS, no barrier).  Layers with too few tiles to cover the chip are split over kernel offsets (S > 1,
// partial tiles to the workspace, fixed-order k_sum_parts) — a handful of splits, not one per offset.
template <int TM, int TN>
__global__ __launch_bounds__(256, 3) void k_conv_stream(
    const float* __restrict__ in, const float* __restrict__ W, const int* __restrict__ nbr,
    const unsigned int* __restrict__ gmask, const int* __restrict__ out_index, float* __restrict__ out, int64_t n_out,
    int K, int Cin, int Cout, int S, int tiles_m, int tiles_n, int n_items) {
  static_assert(TM == 1 || TM == 2, "lane <-> row of the tile");
  typedef typename VecT<TN>::T vecn;
  constexpr int ROWS = TM * 32, COLS = TN * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nW = gridDim.x * 4;
  const int item0 = blockIdx.x * 4 + wave;
  if (item0 >= n_items) return;
  TR_DECL;
  TR(0);
  int tr_quads = 0;
  (void)tr_quads;
  const int NQ4 = Cin >> 5;                                   // quads per offset
  const unsigned int row_bytes = 4u * (unsigned int)Cin, wrow_bytes = 4u * (unsigned int)Cout;

  // cursor two quads ahead: the full walk state
  struct PCur {
    int item, k, cq;
    unsigned int sm0, sm1, rem;
    bool newslot, newitem, last, done;
  };
  struct NCur {                               // what the operand loads of a quad need
    int item, k, cq, n0;
    bool v0, v1, last, newslot, newitem, done;
  };
  struct CCur {                               // what the MFMAs / the store of a quad need
    int item;
    bool v0, v1, last;
  };
  auto zmask = [&](int z) -> unsigned int {
    if (S == 1) return (1u << K) - 1u;        // K <= 31
    unsigned int m = 0u;
    for (int k = z; k < K; k += S) m |= 1u << k;
    return m;
  };
  // group-mask words of an item's row tile (scalar loads, requested one item ahead)
  unsigned int gq0 = 0u, gq1 = 0u;
  auto load_gm = [&](int item, unsigned int& g0, unsigned int& g1) {
    const int it = item < n_items ? item : item0;
    const int tr = (it / tiles_n) % tiles_m;
    g0 = gmask[tr * TM];
    g1 = TM == 2 ? gmask[tr * TM + 1] : 0u;
  };
  auto enter_item = [&](PCur& c, int item, unsigned int g0, unsigned int g1) {
    c.item = item;
    const int z = (item / tiles_n) / tiles_m;
    const unsigned int zm = zmask(z);
    c.sm0 = g0 & zm;
    c.sm1 = g1 & zm;
    unsigned int rem = c.sm0 | c.sm1;
    if (rem) {
      c.k = __ffs(rem) - 1;
      rem &= rem - 1;
      c.cq = 0;
    } else {                                  // nothing to multiply: one null quad, so that the (zero) tile is still stored
      c.k = -1;
      c.cq = NQ4 - 1;
    }
    c.rem = rem;
    c.newslot = true;
    c.newitem = true;
    c.last = (c.cq == NQ4 - 1) && rem == 0u;
  };
  auto advance = [&](PCur& c) {               // next quad of the wave's stream
    if (c.done) return;
    c.newslot = false;
    c.newitem = false;
    if (c.cq + 1 < NQ4) {
      ++c.cq;
    } else if (c.rem) {
      c.k = __ffs(c.rem) - 1;
      c.rem &= c.rem - 1;
      c.cq = 0;
      c.newslot = true;
    } else {
      const int nxt = c.item + nW;
      if (nxt >= n_items) {
        c.done = true;
        c.last = false;
        return;
      }
      enter_item(c, nxt, gq0, gq1);
      load_gm(nxt + nW, gq0, gq1);
      return;
    }
    c.last = (c.cq == NQ4 - 1) && c.rem == 0u;
  };
  auto to_n = [&](const PCur& c) -> NCur {
    NCur n;
    n.item = c.item;
    n.k = c.k;
    n.cq = c.cq;
    n.n0 = (c.item % tiles_n) * COLS;
    n.v0 = c.k >= 0 && ((c.sm0 >> c.k) & 1u);
    n.v1 = TM == 2 && c.k >= 0 && ((c.sm1 >> c.k) & 1u);
    n.last = c.last;
    n.newslot = c.newslot;
    n.newitem = c.newitem;
    n.done = c.done;
    return n;
  };

  // ---- per-lane operand addressing ----------------------------------------------------------------------
  // A: 64-bit row pointers (gathered row or the zero row), refreshed when the load cursor enters a new offset.
  // W: (wave-uniform base of the unit) + 32-bit lane offset, one per channel e of the fragment -> SGPR-base loads.
  const char* ap[TM];
  unsigned int amask[TM];                     // 0 for a lane whose row is absent (it re-reads the 32-byte zero row), else ~0
  unsigned int woff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) woff[e] = (4u * (unsigned int)h + (unsigned int)e) * wrow_bytes + 4u * (unsigned int)(TN * r);
  int idxP[TM];
  int oP = 0, oN = 0, oC = 0;                 // output row of tile row `lane` (through out_index), per open item
  auto request_rows = [&](const PCur& c) {
    // neighbour rows of offset c.k for this lane's A-fragment rows; (new item) the output row of tile row `lane`
    const int64_t m0 = (int64_t)((c.item / tiles_n) % tiles_m) * ROWS;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int64_t row = m0 + i * 32 + r;
      if (row >= n_out) row = n_out - 1;
      idxP[i] = nbr[(int64_t)(c.k < 0 ? 0 : c.k) * n_out + row];
    }
    if (c.newitem) {
      int64_t row = m0 + (TM == 1 ? r : lane);
      if (row >= n_out) row = n_out - 1;
      oP = out_index ? out_index[row] : (int)row;
    }
  };
  auto take_rows = [&](const NCur& n) {       // the load cursor enters a new offset: idxP (requested a quad ago) -> pointers
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const bool absent = idxP[i] < 0 || n.k < 0;
      ap[i] = absent ? reinterpret_cast<const char*>(g_zero_row_reg) + 16 * h
                     : reinterpret_cast<const char*>(in) + (size_t)idxP[i] * row_bytes + 16 * h;
      amask[i] = absent ? 0u : ~0u;
    }
    if (n.newitem) oN = oP;
  };

  struct Buf {
    f32x4 a[TM];
    vecn b[4];
  };
  Buf b0, b1, b2, b3;
  auto load = [&](Buf& B, int ncq, int nk, int nn0, const char* const (&rp)[TM], const unsigned int (&rm)[TM], int j) {
    // unit j of the quad (offset nk, 32-channel group ncq, column tile at nn0) whose rows are rp / rm
    const int cq = __builtin_amdgcn_readfirstlane(ncq), kk = __builtin_amdgcn_readfirstlane(nk < 0 ? 0 : nk);
    const int cn0 = __builtin_amdgcn_readfirstlane(nn0);
    const unsigned int aoff = (unsigned int)(cq * 4 + j) * 32u;
#pragma unroll
    for (int i = 0; i < TM; ++i) B.a[i] = *reinterpret_cast<const f32x4*>(rp[i] + (aoff & rm[i]));
    const char* wbase = reinterpret_cast<const char*>(W) +
                        ((size_t)((size_t)kk * Cin + (cq * 4 + j) * 8) * Cout + cn0) * sizeof(float);
#pragma unroll
    for (int e = 0; e < 4; ++e) B.b[e] = *reinterpret_cast<const vecn*>(wbase + woff[e]);
  };
  f32x16 acc[TM][TN];
  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  auto compute = [&](const Buf& B, bool v0, bool v1) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i == 0 ? v0 : v1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(B.a[i][e], B.b[e][j], acc[i][j], 0, 0, 0);
      }
    }
  };
  auto store_tile = [&](int item, int orow_lane) {
    const int it = __builtin_amdgcn_readfirstlane(item);
    const int tc = it % tiles_n, t = it / tiles_n;
    const int cm0 = (t % tiles_m) * ROWS, cz = t / tiles_m;
    float* dst = out + (int64_t)cz * n_out * Cout + tc * COLS + TN * r;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;           // tile row of accumulator element (i, e)
        const int orow = __shfl(orow_lane, tr, 64);
        if ((int64_t)cm0 + tr < n_out) {
          vecn v;
#pragma unroll
          for (int j = 0; j < TN; ++j) v[j] = acc[i][j][e];
          *reinterpret_cast<vecn*>(dst + (int64_t)orow * Cout) = v;
        }
      }
  };

  // ---- fill the pipeline ---------------------------------------------------------------------------------
  PCur P;
  NCur N;
  CCur C;
  {
    unsigned int g0, g1;
    load_gm(item0, g0, g1);
    P.done = false;
    enter_item(P, item0, g0, g1);
    load_gm(item0 + nW, gq0, gq1);
  }
  request_rows(P);
  N = to_n(P);
  take_rows(N);
  advance(P);
  if (!P.done && P.newslot) request_rows(P);
  load(b0, N.cq, N.k, N.n0, ap, amask, 0);
  load(b1, N.cq, N.k, N.n0, ap, amask, 1);
  load(b2, N.cq, N.k, N.n0, ap, amask, 2);
  clear();
  TR(1);
#ifdef FC_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TR(2);
#endif
  // ---- steady state: one quad per iteration; operands are requested 3 units ahead, each load batch issued beside the
  //      MFMAs of an independent buffer (the one just consumed), so that address arithmetic and VMEM issue hide behind
  //      the matrix pipe instead of between its bursts ------------------------------------------------------------
  while (true) {
    C.item = N.item;
    C.v0 = N.v0;
    C.v1 = N.v1;
    C.last = N.last;
    oC = oN;
    // the 4th unit of the quad being multiplied is requested first, while `ap` still points at ITS rows
    load(b3, N.cq, N.k, N.n0, ap, amask, 3);
    N = to_n(P);
    if (N.newslot && !N.done) take_rows(N);
    advance(P);
    if (!P.done && P.newslot) request_rows(P);
    const bool v0 = __builtin_amdgcn_readfirstlane((int)C.v0) != 0, v1 = __builtin_amdgcn_readfirstlane((int)C.v1) != 0;
    compute(b0, v0, v1);
    load(b0, N.cq, N.k, N.n0, ap, amask, 0);
    compute(b1, v0, v1);
    load(b1, N.cq, N.k, N.n0, ap, amask, 1);
    compute(b2, v0, v1);
    load(b2, N.cq, N.k, N.n0, ap, amask, 2);
    compute(b3, v0, v1);
#ifdef FC_TRACE
    ++tr_quads;
#endif
    if (C.last) {
      if (N.done) TR(3);
      store_tile(C.item, oC);
      clear();
      if (N.done) break;
    }
  }
  TR(5);
  TR_FLUSH(tr_quads * 4);
}

// one word per 32-row group of a (mask-sorted) neighbour table: bit k = some row of the group has a neighbour at offset k
__global__ void k_group_masks(const int* __restrict__ nbr, int64_t n_out, int K, unsigned int* __restrict__ gmask) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64 + lane;
  if (row - lane >= n_out) return;
  unsigned int lo = 0u, hi = 0u;
  for (int k = 0; k < K; ++k) {
    int t = -1;
    if (row < n_out) t = nbr[(int64_t)k * n_out + row];
    const unsigned long long b = __ballot(t >= 0);
    if ((unsigned int)b) lo |= 1u << k;
    if ((unsigned int)(b >> 32)) hi |= 1u << k;
  }
  if (lane == 0) {
    gmask[(row >> 5)] = lo;
    if (row + 32 < n_out) gmask[(row >> 5) + 1] = hi;
  }
}

// ---- host-side launchers (called by the C-ABI dispatchers of conv.hip) ------------------------------------------

template <int TM, int TN, bool WSPLIT>
static int launch_conv_reg(const float* in, const float* W, const int* nbr, const int* out_index, const int* cnt,
                           float* dst, int64_t n_rows, int K, int Cin, int Cout, int S, hipStream_t stream) {
  constexpr int ROWS = TM * 32, COLS = TN * 32;
  const int slots = nbr ? (cnt ? 1 : (int)fc_cdiv(K, S)) : 0;
  const size_t lds = WSPLIT ? (size_t)2 * TM * 16 * 64 * TN * sizeof(float) + (size_t)slots * 64 * sizeof(int) + 64
                            : (size_t)4 * slots * 64 * sizeof(int);
  dim3 grid;
  if (WSPLIT) grid = dim3((unsigned)fc_cdiv(n_rows, ROWS), (unsigned)(Cout / COLS), (unsigned)S);
  else grid = dim3((unsigned)fc_cdiv(n_rows, 2 * ROWS), (unsigned)fc_cdiv(Cout / COLS, 2), (unsigned)S);
  if (nbr) k_conv_reg<TM, TN, true, WSPLIT><<<grid, 256, lds, stream>>>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout);
  else k_conv_reg<TM, TN, false, WSPLIT><<<grid, 256, lds, stream>>>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_conv_reg_launch(const float* in, const float* W, const int* nbr, const int* out_index, const int* cnt, float* dst,
                       int64_t n_rows, int K, int Cin, int Cout, int S, int variant, hipStream_t stream) {
  if (Cin % 8 != 0 || Cout % 64 != 0 || K > 32) return FC_EINVAL;
  switch (variant) {
    case FC_REG_64x64_SPLIT: return launch_conv_reg<2, 2, true>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, S, stream);
    case FC_REG_64x64_WAVE: return launch_conv_reg<2, 2, false>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, S, stream);
    case FC_REG_32x128_SPLIT:
      if (Cout % 128) return FC_EINVAL;
      return launch_conv_reg<1, 4, true>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, S, stream);
    case FC_REG_64x128_SPLIT:
      if (Cout % 128) return FC_EINVAL;
      return launch_conv_reg<2, 4, true>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, S, stream);
    case FC_REG_32x64_SPLIT: return launch_conv_reg<1, 2, true>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, S, stream);
    default: return FC_EINVAL;
  }
}

int fc_wgrad_reg_launch(const float* in, const float* gout, const int* pin, const int* pout, const int* cnt, float* part,
                        int64_t n_out, int K, int Cin, int Cout, int S, int64_t rows_per_split, hipStream_t stream) {
  if (Cin % 64 != 0 || Cout % 64 != 0) return FC_EINVAL;
  dim3 grid((unsigned)S, (unsigned)(K * (Cin / 64) * (Cout / 64)));
  const size_t lds = (size_t)2 * 32 * 64 * 2 * sizeof(float);
  if (cnt) k_wgrad_reg<true, true><<<grid, 256, lds, stream>>>(in, gout, pin, pout, cnt, part, n_out, K, Cin, Cout, rows_per_split);
  else if (pin) k_wgrad_reg<false, true><<<grid, 256, lds, stream>>>(in, gout, pin, pout, cnt, part, n_out, K, Cin, Cout, rows_per_split);
  else k_wgrad_reg<false, false><<<grid, 256, lds, stream>>>(in, gout, pin, pout, cnt, part, n_out, K, Cin, Cout, rows_per_split);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

static int g_num_cu = 0;
static int num_cu() {
  if (!g_num_cu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_num_cu = n;
  }
  return g_num_cu;
}

void fc_conv_stream_plan(int64_t n_rows, int K, int Cin, int Cout, int tile, int force_S, int* tm, int* S, int* items) {
  // tile: 0 = auto, 1 = 32-row tiles, 2 = 64-row tiles (columns: always 64)
  const int64_t t64 = fc_cdiv(n_rows, 64) * (Cout / 64);
  int TMsel = tile ? tile : (t64 >= 3072 ? 2 : 1);
  const int64_t tiles = fc_cdiv(n_rows, 32 * TMsel) * (Cout / 64);
  int s = 1;
  if (K > 1 && tiles < 2048) {
    s = (int)fc_cdiv(2048, tiles);
    if (s > 8) s = 8;
    if (s > K) s = K;
  }
  if (force_S) s = force_S > K ? K : force_S;
  *tm = TMsel;
  *S = s;
  *items = (int)(tiles * s);
}

int fc_conv_stream_launch(const float* in, const float* W, const int* nbr, const unsigned int* gmask, const int* out_index,
                          float* dst, int64_t n_rows, int K, int Cin, int Cout, int tm, int S, hipStream_t stream) {
  if (Cin % 32 != 0 || Cout % 64 != 0 || K > 31 || !nbr || !gmask) return FC_EINVAL;
  const int tiles_m = (int)fc_cdiv(n_rows, 32 * tm), tiles_n = Cout / 64;
  const int n_items = tiles_m * tiles_n * S;
  const int occ = 3;
  int blocks = (n_items + 3) / 4;
  const int cap = num_cu() * occ;
  if (blocks > cap) blocks = cap;
  if (tm == 1) k_conv_stream<1, 2><<<blocks, 256, 0, stream>>>(in, W, nbr, gmask, out_index, dst, n_rows, K, Cin, Cout, S, tiles_m, tiles_n, n_items);
  else k_conv_stream<2, 2><<<blocks, 256, 0, stream>>>(in, W, nbr, gmask, out_index, dst, n_rows, K, Cin, Cout, S, tiles_m, tiles_n, n_items);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_group_masks_launch(const int* nbr, int64_t n_out, int K, unsigned int* gmask, hipStream_t stream) {
  if (K < 1 || K > 31) return FC_EINVAL;
  if (n_out == 0) return FC_OK;
  k_group_masks<<<(unsigned)fc_cdiv(n_out, 256), 256, 0, stream>>>(nbr, n_out, K, gmask);
  FC_CHECK_LAUNCH();
  return FC_OK;
}
