// Sparse convolution on gfx950 as an OUTPUT-STATIONARY implicit GEMM over a dense neighbour
// table nbr[k][o] (built by fc_kernel_map):
//     out[o, :] = sum_k  in[nbr[k][o], :] @ W[k]            (rows with nbr < 0 contribute zero)
// Each output row is written exactly once (no atomics, deterministic).  The same kernel computes
// dgrad (gather over the transposed table with W[k]^T) and, with nbr == NULL (identity), the dense
// GEMMs of the generative transposed convolution and the 1x1 head convolutions.
//
// fp32 in / fp32 accumulate on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
// Workgroup = 256 threads = 4 wave64 in a 2x2 grid; gathered input rows and the W[k] slab are
// staged through LDS with 16-byte coalesced loads.
//
// Replaces MinkowskiEngine's ConvolutionForward/Backward (and ConvolutionTranspose, 1x1 mm) called
// from me_resnet.py:19-21,56-62, BasicBlock, fcaf3d_neck_with_head.py:52,60-69,83-85,257-263.
#include "fc_common.h"
#include <cstdlib>
#ifdef FC_TRACE
__device__ unsigned long long* g_trace_buf_lds;
__device__ int g_trace_cap_lds;
extern "C" int fc_debug_trace_lds(unsigned long long* buf, int cap) {
  FC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf_lds), &buf, sizeof(buf)));
  FC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cap_lds), &cap, sizeof(cap)));
  return FC_OK;
}
#define TR_BUF g_trace_buf_lds
#define TR_CAP g_trace_cap_lds
#endif
#include "fc_trace.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: struct float4 copies lower to memcpy and
                                                          // pin the prefetch registers in scratch (r1 finding)

__device__ __attribute__((aligned(16))) float g_zero_row[64];   // what an absent neighbour gathers from (BKT <= 64 floats)

// Wave priority around the MFMA block of a stage (s_setprio 1 ... 0): a wave that is multiplying goes ahead of the waves
// that are staging, so the resident waves of a SIMD drift apart instead of all meeting their barriers together.  r2,
// same box, instruction-identical otherwise: +2.4...3.4 % on the 441k-row launches, +3 % on the 64k-row level, +4 % on
// the 3.5k-row pair mode, +1...4 % on k_wgrad_multi; the one-offset weight-gradient kernels LOSE 2...7 % and the LDS-DMA
// kernel up to 9 %, so those stay at priority 0.  Static per-workgroup priorities (by block index, by hardware wave slot):
// +1 % or -10 %.  g_fc_prio = -1 switches it off (A/B: tools/nbench --prio -1, FC_PRIO_OFF=1; not a C-ABI entry point).
__device__ int g_fc_prio;
extern "C" int fc_debug_set_prio(int mode) {
  FC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fc_prio), &mode, sizeof(mode)));
  return FC_OK;
}

// r5, SURVEY.md 8(f) rank 4 "bf16 fast mode" — a flagged NON-PARITY extra, never the benchmark's `value`: with the switch on,
// the split-bf16 launches that read a weight image (k_conv_x6 BSRC 2, 128-row tiles) and k_wgrad_x6t run their FAST variants:
// ONLY plane 0 of either operand = the operand rounded to nearest bf16 (x6_rn2), fp32 accumulate — one v_mfma_f32_32x32x16_bf16
// per 32x32x16 block instead of six.  Process-global host switch (FC_BF16=1 / fc_set_bf16_fast); default off.
static int g_bf16_fast = 0;
extern "C" int fc_set_bf16_fast(int on) {
  g_bf16_fast = on ? 1 : 0;
  return FC_OK;
}

// r6: how the split kernels cut an fp32 operand (conv_x6.h): 2 = two fp16 pieces, three products ("h3", the default), 0 = three
// bf16 pieces, six products (r3-r5).  Process-global (FC_SPLIT_MODE / fc_set_split_mode): weight images are built in the mode
// that is current when fc_x6_weight_image(s) runs and MUST be read in that mode — whoever switches rebuilds them (functional.py
// set_split_mode does).  The bf16 fast mode reads plane 0 of a SIX-product image: fc_set_bf16_fast(1) implies mode 0 while it is on.
static int g_split_mode = -1;
static inline int split_mode() {
  if (g_split_mode < 0) {
    const char* e = getenv("FC_SPLIT_MODE");
    g_split_mode = (e && atoi(e) == 0) ? 0 : 2;
  }
  return g_bf16_fast ? 0 : g_split_mode;
}
extern "C" int fc_set_split_mode(int mode) {
  if (mode != 0 && mode != 2) return FC_EINVAL;
  g_split_mode = mode;
  return FC_OK;
}
extern "C" int fc_get_split_mode(void) { return split_mode(); }
static int g_h3r = -1;                           // which h3 launches run the register-operand kernel (launch_conv_mfma)
extern "C" int fc_debug_set_h3r(int mode) {
  if (mode < 0 || mode > 2) return FC_EINVAL;
  g_h3r = mode;
  return FC_OK;
}

#include <mutex>
#include "conv_x6.h"
#include "wgrad_x6.h"
#include "conv_h3r.h"

// ---- r6: max |x| of an operand tensor (the scale of the h3 split, conv_x6.h) -----------------------------------------------------
// slot = FC_AMAX_SLOT_BYTES (fc_common.h: 32 sub-words, one per 64-byte line; the kernels read their maximum); this pass uses the
// first line: [0] the result (sub-word 0: bit pattern of max |x|), [1] running maximum, [2] finished blocks.
// Every block folds its elements into [1] with ONE integer atomicMax (order-independent: bit-reproducible); the last block to
// finish publishes [1] to [0] and clears [1], [2] for the slot's next use.  Slots must start zeroed (fc_amax_slots / the ring).
__global__ __launch_bounds__(256) void k_amax(const f32x4* __restrict__ x, int64_t n4, const float* __restrict__ tail, int ntail,
                                              unsigned* __restrict__ slot) {
  __shared__ unsigned wm[4];
  __shared__ int last_s;
  unsigned m = 0u;
  // FINITE elements only: an overflowed activation must poison the rows that gather it (inf s = inf -> NaN, as on the six-product
  // route), not flatten the scale of the whole tensor.  Four loads in flight per thread.
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i + q * stride < n4 ? x[i + q * stride] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const unsigned u = __float_as_uint(v[q][j]) & 0x7fffffffu; m = (u > m && u < 0x7f800000u) ? u : m; }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) { const unsigned u = __float_as_uint(tail[threadIdx.x]) & 0x7fffffffu; m = (u > m && u < 0x7f800000u) ? u : m; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned b = wm[0];
    for (int w = 1; w < 4; ++w) b = wm[w] > b ? wm[w] : b;
    if (b > __hip_atomic_load(&slot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&slot[1], b);
    __threadfence();
    last_s = atomicAdd(&slot[2], 1u) == gridDim.x - 1;
    if (last_s) {
      __threadfence();
      slot[0] = atomicExch(&slot[1], 0u);
      slot[2] = 0u;
    }
  }
}

// library-owned ring of slots per device, for the operands whose caller brings no amax word of its own
constexpr int AMAX_RING = 2048;
static unsigned* g_amax_ring[16] = {};
static unsigned g_amax_next[16] = {};
static std::mutex g_amax_ring_mu;
static int amax_ring_slot(unsigned** slot) {
  int dev = 0;
  FC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return FC_EINVAL;
  std::lock_guard<std::mutex> lock(g_amax_ring_mu);          // (two threads' first calls must not both create the ring)
  if (!g_amax_ring[dev]) {
    FC_HIP(hipMalloc((void**)&g_amax_ring[dev], (size_t)AMAX_RING * FC_AMAX_SLOT_BYTES));
    FC_HIP(hipMemset(g_amax_ring[dev], 0, (size_t)AMAX_RING * FC_AMAX_SLOT_BYTES));
  }
  const unsigned i = __atomic_fetch_add(&g_amax_next[dev], 1u, __ATOMIC_RELAXED) % AMAX_RING;
  *slot = g_amax_ring[dev] + (size_t)i * (FC_AMAX_SLOT_BYTES / 4);
  return FC_OK;
}
static int amax_launch(const float* x, int64_t n, unsigned* slot, hipStream_t stream) {
  if (n <= 0) { FC_HIP(hipMemsetAsync(slot, 0, 4, stream)); return FC_OK; }
  const int64_t n4 = n / 4;
  int64_t blocks = fc_cdiv(n4 > 0 ? n4 : 1, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  k_amax<<<(unsigned)blocks, 256, 0, stream>>>(reinterpret_cast<const f32x4*>(x), n4, x + 4 * n4, (int)(n - 4 * n4), slot);
  FC_CHECK_LAUNCH();
  return FC_OK;
}
// the caller's amax words for the NEXT convolution / weight-gradient call of this thread (fc_conv_amax_hint): a0 = the gathered
// operand (`in`), a1 = `gout` (weight gradients); NULL: computed here, into a ring slot.  Consumed by that call.
static thread_local const unsigned* t_amax_hint[2] = {nullptr, nullptr};
struct AmaxHintScope { ~AmaxHintScope() { t_amax_hint[0] = t_amax_hint[1] = nullptr; } };      // every convolution entry point drops the hints when it returns
static int operand_amax(const float* x, int64_t n, int which, hipStream_t stream, const unsigned** out) {
  const unsigned* h = t_amax_hint[which];
  if (h) { *out = h; return FC_OK; }
  unsigned* slot;
  int rc = amax_ring_slot(&slot);
  if (rc) return rc;
  rc = amax_launch(x, n, slot, stream);
  *out = slot;
  return rc;
}
extern "C" int fc_amax(const float* x, int64_t n, unsigned* slot, hipStream_t stream) {
  if (!slot || n < 0 || (n > 0 && !x) || ((uintptr_t)x & 15)) return FC_EINVAL;
  return amax_launch(x, n, slot, stream);
}
extern "C" int fc_conv_amax_hint(const unsigned* amax_in, const unsigned* amax_gout) {
  t_amax_hint[0] = amax_in;
  t_amax_hint[1] = amax_gout;
  return FC_OK;
}

#define BK 32            // reduction slab (input channels per stage / rows per stage for wgrad)
#define LDA (BK + 4)     // A row stride in floats: 16B-aligned rows, conflict-free ds_read_b128 (9r mod 16)

// ------------------------------------------------------------------------------------------------
// Pipeline per workgroup: (1) which kernel offsets have any neighbour among the tile's rows (offsets without one are
// skipped outright), OR-reduced over the tile's rows; (2) walk the surviving (offset,
// Cin-slab) stages with the NEXT stage's gathers + weight loads issued into registers before the current stage's MFMAs
// run (single LDS buffer, global-load latency hidden behind the matrix pipe).
// gridDim.z > 1 = split over kernel offsets (offset k handled by split k % gridDim.z) for layers whose
// row count cannot fill the chip; partial tiles go to `out` + z*n_out*Cout and are summed by k_sum_parts.
// WM = waves along the rows (2: 2 x 2 waves; 4: 4 x 1 — the 256 x 64 tile for 64-wide outputs: every wave still owns a
// 64 x 64 quadrant, i.e. the operand re-use of the 128 x 128 tile, where a 128 x 64 tile halves it).
// WT: the weights are given TRANSPOSED, W[k] as (Cout, Cin) row-major — the backward-data pass reads the layer's own
// (K, Cin, Cout) kernel as the (Cout -> Cin) operator it needs, no per-step transpose launch.  Only the staging differs:
// a thread loads 4 reduction-consecutive floats of one output column and stores them as 4 ds_write_b32 (conflict-free:
// consecutive lanes = consecutive columns); the LDS image, the fragment reads and the results are the same.
template <int BM, int BN, int BKT, bool HAS_NBR, int WM = 2, bool WT = false>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128 && BKT == 32) ? 4 : (WM == 4 ? 3 : 2)) void k_conv_mfma(const float* __restrict__ in, const float* __restrict__ W,
                                                   const int* __restrict__ nbr,
                                                   const int* __restrict__ out_index, const int* __restrict__ cnt,
                                                   float* __restrict__ out, int64_t n_out, int K, int Cin, int Cout) {
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);      // 32x32 MFMA tiles per wave
  constexpr int RW = BM / WM, CW = BN / WN;      // rows / columns of a wave's part of the tile
  constexpr int LDAT = BKT + 4;                  // (BKT+4)/4 odd -> conflict-free ds_read_b128 of the A fragments
  constexpr int A4 = BKT / 4;                    // float4 per gathered row slab
  constexpr int APASS = 256 / A4;                // rows staged per pass
  constexpr int AR = BM / APASS;                 // float4 gathers per thread per stage
  constexpr int BR = BKT * BN / 1024;            // float4 weight loads per thread per stage
  __shared__ __attribute__((aligned(16))) float As[BM * LDAT];
  __shared__ __attribute__((aligned(16))) float Bs[BKT * BN];
  __shared__ unsigned int kmask_s;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = WM == 4 ? wave : wave >> 1, wc = WM == 4 ? 0 : wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int64_t bx = blockIdx.x;
  const int n0 = blockIdx.y * BN;
  int S = gridDim.z, z = blockIdx.z;
  TR_DECL;
  TR(0);
  int tr_units = 0;
  (void)tr_units;
  if (cnt) {
    // pair mode (fc_conv_fwd_pairs): `nbr` row z lists the input rows of offset z's cnt[z] pairs and the tile computes
    // those compacted rows only: T_z[j] = in[pair_in[z][j]] @ W[z], written to slab z of the workspace.  From here on
    // it is a one-offset convolution over cnt[z] rows.  Two grid shapes: (row tiles, column tiles, K) with the workgroups
    // beyond cnt[z] exiting at once, or — gridDim.z == 1, the caller knows sum_k ceil(cnt[k] / BM) — a LINEAR list of the
    // live tiles only, offset-major (r2: the dead workgroups of the 3-D grid skew the round-robin placement, a few
    // shader engines overflow and their last workgroups start a whole round late: 115 -> 88 us on the 3.5k-row level).
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
    W += (int64_t)z * Cin * Cout;
    out += (int64_t)z * stride * Cout;
    K = 1; S = 1; z = 0;
  }
  const int64_t m0 = bx * BM;
  const int a_c4 = tid % A4, a_r = tid / A4;     // A staging: A4 float4 per row, APASS rows per pass

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
      // wave-level OR, one LDS atomic per wave
      for (int off = 32; off > 0; off >>= 1) mk |= __shfl_xor(mk, off, 64);
      if (lane == 0 && mk) atomicOr(&kmask_s, mk);
    }
    __syncthreads();
    kmask = kmask_s;
  }
  TR(1);

  // ---- (2) software-pipelined stage loop -------------------------------------------------------------
  if (kmask) {
    int k = __ffs(kmask) - 1;
    kmask &= kmask - 1;
    int c0 = 0;
    f32x4 av[AR], bv[BR];
    // Prefetch of one stage, written branch-free so that the compiler issues it as batches (the AR neighbour
    // indices, the weights, then the AR gathers) instead of AR serialised index->wait->gather chains with the data
    // waited for on the spot (r1 ISA reading).  An absent neighbour gathers from a row of zeros (pointer select
    // BEFORE the load), so nothing touches the loaded registers until the next stage's LDS store.
    auto load_stage = [&](int kk, int cc) {
      const float* Wk = W + (int64_t)kk * Cin * Cout;
      int v[AR];
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int64_t row = m0 + a_r + APASS * i;
        int64_t rc = row < n_out ? row : n_out - 1;
        int t = HAS_NBR ? nbr[(int64_t)kk * n_out + rc] : (int)rc;    // compile-time: no null test between the loads
        v[i] = row < n_out ? t : -1;
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        int lin = tid + 256 * i;
        if (WT) {
          bv[i] = *reinterpret_cast<const f32x4*>(Wk + (int64_t)(n0 + lin % BN) * Cin + cc + (lin / BN) * 4);
        } else {
          int kr = lin / (BN / 4), c4 = lin % (BN / 4);
          bv[i] = *reinterpret_cast<const f32x4*>(Wk + (int64_t)(cc + kr) * Cout + n0 + c4 * 4);
        }
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const float* src = v[i] < 0 ? g_zero_row + a_c4 * 4 : in + (int64_t)v[i] * Cin + cc + a_c4 * 4;
        av[i] = *reinterpret_cast<const f32x4*>(src);
      }
    };
    load_stage(k, c0);
    const bool prio = g_fc_prio >= 0;             // see g_fc_prio
    while (true) {
#ifdef FC_TRACE
      if (tr_units == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TR(2); }
      tr_units += BKT / 8;
#endif
      __syncthreads();                           // previous stage fully consumed
#pragma unroll
      for (int i = 0; i < AR; ++i)
        *reinterpret_cast<f32x4*>(&As[(a_r + APASS * i) * LDAT + a_c4 * 4]) = av[i];
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        int lin = tid + 256 * i;
        if (WT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) Bs[((lin / BN) * 4 + e) * BN + lin % BN] = bv[i][e];
        } else {
          int kr = lin / (BN / 4), c4 = lin % (BN / 4);
          *reinterpret_cast<f32x4*>(&Bs[kr * BN + c4 * 4]) = bv[i];
        }
      }
      __syncthreads();
      // issue the next stage's global loads before computing this one
      int nk = k, nc0 = c0 + BKT;
      if (nc0 >= Cin) {
        nc0 = 0;
        nk = kmask ? __ffs(kmask) - 1 : -1;
        kmask &= kmask - 1;
      }
      // unconditional (the last iteration re-reads its own stage): a conditionally assigned float4 array is not
      // promoted to registers by the compiler and lands in scratch
      load_stage(nk >= 0 ? nk : k, nk >= 0 ? nc0 : c0);
      if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int q = 0; q < BKT / 8; ++q) {
        f32x4 a[TM];
        float b[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const f32x4*>(&As[(wr * RW + i * 32 + r) * LDAT + 8 * q + 4 * h]);
        // sub-tile j of the wave owns the INTERLEAVED columns TN*r + j of its half of the tile: one ds_read_b64 per
        // (k, lane) instead of two ds_read_b32 (conflict-free: 32 lanes x 8 B = 64 banks), and 8-byte output stores
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (TN == 2) {
            const f32x2 bb = *reinterpret_cast<const f32x2*>(&Bs[(8 * q + 4 * h + e) * BN + wc * CW + 2 * r]);
            b[0][e] = bb[0];
            b[TN - 1][e] = bb[1];
          } else {
            b[0][e] = Bs[(8 * q + 4 * h + e) * BN + wc * CW + r];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            float ae = a[i][e];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b[j][e], acc[i][j], 0, 0, 0);
          }
        }
      }
      if (prio) __builtin_amdgcn_s_setprio(0);
      if (nk < 0) break;
      k = nk;
      c0 = nc0;
    }
  }
  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  TR(3);
  float* dst = out + (int64_t)z * n_out * Cout + n0 + wc * CW + TN * r;
  int orow[TM][16];                              // looked up in one batch ahead of the stores
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = m0 + wr * RW + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      int o = -1;
      if (row < n_out) o = out_index ? out_index[row] : (int)row;      // rows are processed in mask-sorted order
      orow[i][e] = o;
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
  TR(5);
  TR_FLUSH(tr_units);
}

// Deeper-pipelined variant of k_conv_mfma (same tiles, same LDS layout, same results) at 3 waves per SIMD (168 VGPRs):
//  * the neighbour rows of the NEXT kernel offset are requested one offset ahead (vnxt), so that a stage's gathers are
//    issued at once instead of behind an exposed index round trip (r2 ISA reading of k_conv_mfma: index loads ->
//    s_waitcnt vmcnt(0) -> gathers in front of every stage's MFMA block; the 128-VGPR budget of 4 waves/SIMD had no
//    room for the look-ahead — r1 dead end);
//  * the MFMA fragments of 8-channel step q+1 are read from LDS while step q multiplies (double-buffered fragment
//    registers) instead of read -> s_waitcnt lgkmcnt(0) -> multiply.
template <int BM, int BN, int BKT, bool HAS_NBR, int WM = 2, bool WT = false>
__global__ __launch_bounds__(256, 3) void k_conv_mfma_p(const float* __restrict__ in, const float* __restrict__ W,
                                                   const int* __restrict__ nbr,
                                                   const int* __restrict__ out_index, const int* __restrict__ cnt,
                                                   float* __restrict__ out, int64_t n_out, int K, int Cin, int Cout) {
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);      // 32x32 MFMA tiles per wave
  constexpr int RW = BM / WM, CW = BN / WN;      // rows / columns of a wave's part of the tile
  constexpr int LDAT = BKT + 4;                  // (BKT+4)/4 odd -> conflict-free ds_read_b128 of the A fragments
  constexpr int A4 = BKT / 4;                    // float4 per gathered row slab
  constexpr int APASS = 256 / A4;                // rows staged per pass
  constexpr int AR = BM / APASS;                 // float4 gathers per thread per stage
  constexpr int BR = BKT * BN / 1024;            // float4 weight loads per thread per stage
  __shared__ __attribute__((aligned(16))) float As[BM * LDAT];
  __shared__ __attribute__((aligned(16))) float Bs[BKT * BN];
  __shared__ unsigned int kmask_s;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = WM == 4 ? wave : wave >> 1, wc = WM == 4 ? 0 : wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int64_t bx = blockIdx.x;
  const int n0 = blockIdx.y * BN;
  int S = gridDim.z, z = blockIdx.z;
  TR_DECL;
  TR(0);
  int tr_units = 0;
  (void)tr_units;
  if (cnt) {
    // pair mode (fc_conv_fwd_pairs): `nbr` row z lists the input rows of offset z's cnt[z] pairs and the tile computes
    // those compacted rows only: T_z[j] = in[pair_in[z][j]] @ W[z], written to slab z of the workspace.  From here on
    // it is a one-offset convolution over cnt[z] rows.  Two grid shapes: (row tiles, column tiles, K) with the workgroups
    // beyond cnt[z] exiting at once, or — gridDim.z == 1, the caller knows sum_k ceil(cnt[k] / BM) — a LINEAR list of the
    // live tiles only, offset-major (r2: the dead workgroups of the 3-D grid skew the round-robin placement, a few
    // shader engines overflow and their last workgroups start a whole round late: 115 -> 88 us on the 3.5k-row level).
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
    W += (int64_t)z * Cin * Cout;
    out += (int64_t)z * stride * Cout;
    K = 1; S = 1; z = 0;
  }
  const int64_t m0 = bx * BM;
  const int a_c4 = tid % A4, a_r = tid / A4;     // A staging: A4 float4 per row, APASS rows per pass

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
      // wave-level OR, one LDS atomic per wave
      for (int off = 32; off > 0; off >>= 1) mk |= __shfl_xor(mk, off, 64);
      if (lane == 0 && mk) atomicOr(&kmask_s, mk);
    }
    __syncthreads();
    kmask = kmask_s;
  }
  TR(1);

  // ---- (2) software-pipelined stage loop -------------------------------------------------------------
  if (kmask) {
    const int nst = __popc(kmask) * (Cin / BKT);         // stages of this tile
    unsigned int rem = kmask;
    int lk = __ffs(rem) - 1;                              // load cursor: offset / channel slab of the stage requested next
    rem &= rem - 1;
    int lnk = rem ? __ffs(rem) - 1 : lk;                  // ... and the offset after it (its rows are already on their way)
    if (rem) rem &= rem - 1;
    int lc0 = 0;
    bool sw = false;                                      // the cursor has left lk: switch to lnk at the next request
    f32x4 av[AR], bv[BR];
    int vcur[AR], vnxt[AR];
    int64_t arow[AR];
    bool aok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int64_t row = m0 + a_r + APASS * i;
      aok[i] = row < n_out;
      arow[i] = aok[i] ? row : n_out - 1;
    }
    // raw table entries only: nothing touches a requested index until the stage that uses it
    auto fetch_idx = [&](int kk, int (&v)[AR]) {
#pragma unroll
      for (int i = 0; i < AR; ++i) v[i] = HAS_NBR ? nbr[(int64_t)kk * n_out + arow[i]] : (int)arow[i];
    };
    fetch_idx(lk, vcur);
    fetch_idx(lnk, vnxt);
    auto load_stage = [&]() {
      if (sw) {                                           // first slab of a new offset: its rows were requested an offset ago
        sw = false;
        lk = lnk;
#pragma unroll
        for (int i = 0; i < AR; ++i) vcur[i] = vnxt[i];
        if (rem) {
          lnk = __ffs(rem) - 1;
          rem &= rem - 1;
        }
        fetch_idx(lnk, vnxt);
      }
      const float* Wk = W + (int64_t)lk * Cin * Cout;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        int lin = tid + 256 * i;
        if (WT) {
          bv[i] = *reinterpret_cast<const f32x4*>(Wk + (int64_t)(n0 + lin % BN) * Cin + lc0 + (lin / BN) * 4);
        } else {
          int kr = lin / (BN / 4), c4 = lin % (BN / 4);
          bv[i] = *reinterpret_cast<const f32x4*>(Wk + (int64_t)(lc0 + kr) * Cout + n0 + c4 * 4);
        }
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const float* src = (vcur[i] < 0 || !aok[i]) ? g_zero_row + a_c4 * 4 : in + (int64_t)vcur[i] * Cin + lc0 + a_c4 * 4;
        av[i] = *reinterpret_cast<const f32x4*>(src);
      }
    };
    load_stage();
    const bool prio = g_fc_prio >= 0;
    for (int st = 0; st < nst; ++st) {
#ifdef FC_TRACE
      if (tr_units == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TR(2); }
      tr_units += BKT / 8;
#endif
      __syncthreads();                           // previous stage fully consumed
#pragma unroll
      for (int i = 0; i < AR; ++i)
        *reinterpret_cast<f32x4*>(&As[(a_r + APASS * i) * LDAT + a_c4 * 4]) = av[i];
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        int lin = tid + 256 * i;
        if (WT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) Bs[((lin / BN) * 4 + e) * BN + lin % BN] = bv[i][e];
        } else {
          int kr = lin / (BN / 4), c4 = lin % (BN / 4);
          *reinterpret_cast<f32x4*>(&Bs[kr * BN + c4 * 4]) = bv[i];
        }
      }
      __syncthreads();
      // request the next stage before multiplying this one (the last iteration re-reads its own stage: a conditionally
      // assigned staging array lands in scratch)
      if (st + 1 < nst) {
        lc0 += BKT;
        if (lc0 >= Cin) {
          lc0 = 0;
          sw = true;
        }
      }
      load_stage();
      f32x4 fa[2][TM];
      float fb[2][TN][4];
      auto read_frag = [&](int q, int u) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[u][i] = *reinterpret_cast<const f32x4*>(&As[(wr * RW + i * 32 + r) * LDAT + 8 * q + 4 * h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (TN == 2) {
            const f32x2 bb = *reinterpret_cast<const f32x2*>(&Bs[(8 * q + 4 * h + e) * BN + wc * CW + 2 * r]);
            fb[u][0][e] = bb[0];
            fb[u][TN - 1][e] = bb[1];
          } else {
            fb[u][0][e] = Bs[(8 * q + 4 * h + e) * BN + wc * CW + r];
          }
        }
      };
      read_frag(0, 0);
      if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int q = 0; q < BKT / 8; ++q) {
        if (q + 1 < BKT / 8) read_frag(q + 1, (q + 1) & 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float ae = fa[q & 1][i][e];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, fb[q & 1][j][e], acc[i][j], 0, 0, 0);
          }
        }
#ifdef FC_SGB
        // experiment (-DFC_SGB=4, r2): pin "the LDS reads of step q+1 go out FIRST, interleaved with the head of step q's
        // MFMAs" (the compiler otherwise sinks them behind the 16 MFMAs and waits for them on the spot; with the pin the
        // waits become counted lgkmcnt(2/4)).  Measured: +-1 % on 50 of 58 launch shapes, 441k rows 64->128 +1.3 %,
        // 128->64 -3.2 %, 64->64 (256 x 64 tile, at its VGPR limit) -10.5 %: the fragment reads are not what the loop waits for.
        if (q + 1 < BKT / 8) {
#pragma unroll
          for (int g = 0; g < FC_SGB; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, TM * TN * 4 - 2 * FC_SGB, 0);
        }
#endif
      }
      if (prio) __builtin_amdgcn_s_setprio(0);
    }
  }
  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  TR(3);
  float* dst = out + (int64_t)z * n_out * Cout + n0 + wc * CW + TN * r;
  int orow[TM][16];                              // looked up in one batch ahead of the stores
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = m0 + wr * RW + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      int o = -1;
      if (row < n_out) o = out_index ? out_index[row] : (int)row;      // rows are processed in mask-sorted order
      orow[i][e] = o;
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
  TR(5);
  TR_FLUSH(tr_units);
}

// LDS-DMA variant (r2; VERDICT r1 item 1a — double-buffered LDS, one barrier per stage): the gathered rows and the weight slab go
// from global memory STRAIGHT into LDS (`global_load_lds_dwordx4`, 1 KiB per wave instruction: 8 gathered rows x 128 B, or
// 1 KiB of a weight slab), no staging registers and no ds_write pass, into one of TWO stage buffers; a stage is
//     wait for my own DMAs -> barrier -> issue stage s+1's DMAs into the other buffer -> multiply stage s.
// The DMA's LDS image is lane-linear (wave-uniform base + lane x 16 B: MI355X guide), so the A tile cannot be padded; its
// rows are XOR-swizzled instead ON THE SOURCE SIDE: the lane that fills 16-byte slot p of row j fetches the row's slot
// p ^ ((j >> 1) & 7), and the MFMA fragment read of slot c of row j addresses slot c ^ ((j >> 1) & 7) — conflict-free for
// ds_read_b128's 16-lane groups.  Same tiles, prologue, epilogue, grid shapes and results as k_conv_mfma / _p.
template <int BM, int BN, bool HAS_NBR, int WM = 2>
__global__ __launch_bounds__(256, 2) void k_conv_glds(const float* __restrict__ in, const float* __restrict__ W,
                                                      const int* __restrict__ nbr,
                                                      const int* __restrict__ out_index, const int* __restrict__ cnt,
                                                      float* __restrict__ out, int64_t n_out, int K, int Cin, int Cout) {
  constexpr int BKT = 32;
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int RW = BM / WM, CW = BN / WN;
  constexpr int AR = BM / 32;                    // row DMAs per thread per stage (8 rows per wave instruction)
  constexpr int BR = BN / 32;                    // weight DMAs per thread per stage
  constexpr int BRPI = 256 / BN;                 // weight-slab rows per wave instruction
  constexpr int STAGE = BM * BKT + BKT * BN;     // floats per stage buffer
  __shared__ __attribute__((aligned(1024))) float smem[2 * STAGE];         // ONE array (a second one costs vmcnt(0)s: guide §5)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = WM == 4 ? wave : wave >> 1, wc = WM == 4 ? 0 : wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int64_t bx = blockIdx.x;
  const int n0 = blockIdx.y * BN;
  int S = gridDim.z, z = blockIdx.z;
  if (cnt) {                                     // pair mode: see k_conv_mfma_p
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
    W += (int64_t)z * Cin * Cout;
    out += (int64_t)z * stride * Cout;
    K = 1; S = 1; z = 0;
  }
  const int64_t m0 = bx * BM;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  unsigned int kmask;
  {
    // the tile-mask word lives in stage buffer 1: nothing is written there before every wave has passed stage 0's barrier
    unsigned int* kmask_s = reinterpret_cast<unsigned int*>(&smem[STAGE]);
    if (tid == 0) *kmask_s = 0u;
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
      if (lane == 0 && mk) atomicOr(kmask_s, mk);
    }
    __syncthreads();
    kmask = *kmask_s;
  }

  if (kmask) {
    const int nst = __popc(kmask) * (Cin / BKT);
    unsigned int rem = kmask;
    int lk = __ffs(rem) - 1;
    rem &= rem - 1;
    int lnk = rem ? __ffs(rem) - 1 : lk;
    if (rem) rem &= rem - 1;
    int lc0 = 0;
    bool sw = false;
    // this lane's part of a row DMA: row (i*4 + wave)*8 + lane/8 of the tile, LDS slot lane%8, source slot swizzled
    const int a_j = lane >> 3;
    int vcur[AR], vnxt[AR], a_src4[AR];
    int64_t arow[AR];
    bool aok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int lrow = (i * 4 + wave) * 8 + a_j;
      const int64_t row = m0 + lrow;
      aok[i] = row < n_out;
      arow[i] = aok[i] ? row : n_out - 1;
      a_src4[i] = (((lane & 7) ^ ((lrow >> 1) & 7))) * 4;
    }
    auto fetch_idx = [&](int kk, int (&v)[AR]) {
#pragma unroll
      for (int i = 0; i < AR; ++i) v[i] = HAS_NBR ? nbr[(int64_t)kk * n_out + arow[i]] : (int)arow[i];
    };
    fetch_idx(lk, vcur);
    fetch_idx(lnk, vnxt);
    const int b_kr = lane / (BN / 4), b_c4 = lane % (BN / 4);
    auto issue_stage = [&](int buf) {
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
      }
      float* As = smem + buf * STAGE;
      float* Bs = As + BM * BKT;
      const float* Wk = W + (int64_t)lk * Cin * Cout;
      // every source address first (this consumes the index registers: hipcc waits vmcnt(0) at the first use of an ordinary
      // load's result while a DMA is in flight — here nothing is, the loop-top wait has just drained the queue), then the
      // DMAs back to back
      const float* asrc[AR];
      const float* bsrc[BR];
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool ok = vcur[i] >= 0 && aok[i];
        const int64_t off = ok ? (int64_t)vcur[i] * Cin + lc0 : 0;
        asrc[i] = (ok ? in : g_zero_row) + off + a_src4[i];
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) bsrc[i] = Wk + (int64_t)(lc0 + (i * 4 + wave) * BRPI + b_kr) * Cout + n0 + b_c4 * 4;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < BR; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bsrc[i],
                                         (__attribute__((address_space(3))) void*)(Bs + (i * 4 + wave) * BRPI * BN), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < AR; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)asrc[i],
                                         (__attribute__((address_space(3))) void*)(As + (i * 4 + wave) * 8 * BKT), 16, 0, 0);
    };
    issue_stage(0);
    const int fsw = (r >> 1) & 7;                // the fragment rows' swizzle (tile row = 32-aligned base + r)
    for (int st = 0; st < nst; ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my DMAs of stage st have landed ...
      __syncthreads();                                      // ... and everyone's; stage st-1 is fully consumed
      if (st + 1 < nst) {
        lc0 += BKT;
        if (lc0 >= Cin) {
          lc0 = 0;
          sw = true;
        }
        issue_stage((st + 1) & 1);
      }
      const float* As = smem + (st & 1) * STAGE;
      const float* Bs = As + BM * BKT;
      f32x4 fa[2][TM];
      float fb[2][TN][4];
      auto read_frag = [&](int q, int u) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[u][i] = *reinterpret_cast<const f32x4*>(&As[(wr * RW + i * 32 + r) * BKT + 4 * ((2 * q + h) ^ fsw)]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (TN == 2) {
            const f32x2 bb = *reinterpret_cast<const f32x2*>(&Bs[(8 * q + 4 * h + e) * BN + wc * CW + 2 * r]);
            fb[u][0][e] = bb[0];
            fb[u][TN - 1][e] = bb[1];
          } else {
            fb[u][0][e] = Bs[(8 * q + 4 * h + e) * BN + wc * CW + r];
          }
        }
      };
      read_frag(0, 0);
#pragma unroll
      for (int q = 0; q < BKT / 8; ++q) {
        if (q + 1 < BKT / 8) read_frag(q + 1, (q + 1) & 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float ae = fa[q & 1][i][e];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, fb[q & 1][j][e], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  }
  float* dst = out + (int64_t)z * n_out * Cout + n0 + wc * CW + TN * r;
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

// out[i] = sum_z part[z][i]   (fixed order; elems % 4 == 0)
__global__ void k_sum_parts(const float* __restrict__ part, float* __restrict__ out, int64_t elems4, int S) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems4) return;
  float4 a = reinterpret_cast<const float4*>(part)[i];
  for (int s = 1; s < S; ++s) {
    float4 b = reinterpret_cast<const float4*>(part)[(int64_t)s * elems4 + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  reinterpret_cast<float4*>(out)[i] = a;
}

// out[o] = sum_k T_k[pos[k][o]] over the offsets that have a neighbour at o (fixed k order: deterministic)
__global__ void k_sum_pairs(const float* __restrict__ part, const int* __restrict__ pos, float* __restrict__ out,
                            int64_t n_out, int K, int Cout) {
  const int c4n = Cout / 4;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * c4n) return;
  int64_t o = t / c4n;
  int c4 = (int)(t % c4n);
  // nine offsets per batch: their positions, then their rows, are requested together (an absent offset reads the zero
  // row), and added in offset order — r3: one dependent (position -> row) chain per offset made this launch 19 us on
  // every level, 1 ms per step
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 9) {
    int j[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) j[u] = k0 + u < K ? pos[(int64_t)(k0 + u) * n_out + o] : -1;
    f32x4 v[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const float* src = j[u] >= 0 ? part + ((int64_t)(k0 + u) * n_out + j[u]) * Cout + c4 * 4 : g_zero_row + (c4 & 15) * 4;
      v[u] = *reinterpret_cast<const f32x4*>(src);
    }
#pragma unroll
    for (int u = 0; u < 9; ++u)
      if (j[u] >= 0) a += v[u];
  }
  *reinterpret_cast<f32x4*>(out + o * Cout + c4 * 4) = a;
}

// ---- r5: the same two sums, leaving the BatchNorm statistics of what they write -----------------------------------------------
// A workgroup = 16 float4 lanes (64 channels) x 16 row lanes; grid (row blocks of RB rows, C / 64).  Besides out, every workgroup
// writes the column sums of its rows' results and of their squares: stats[row block][2][C] (norm.hip k_bn2_apply / k_bn2_finalize
// combine them in fp64).  RB = fc_stat_rb(n): row blocks are sized so that small matrices leave <= 64 of them (then the
// BatchNorm that follows is ONE launch) while every thread still has at most a handful of rows.
__host__ __device__ static inline int fc_stat_rb(int64_t n) { return n <= 1024 ? 16 : (n <= 4096 ? 64 : 32); }

__device__ __forceinline__ void stat_block_reduce(f32x4 a1, f32x4 a2, float* __restrict__ stats, int C, f32x4* sm /*[16][2][16]*/) {
  const int lane = threadIdx.x & 15, rl = threadIdx.x >> 4;
  sm[(rl * 2 + 0) * 16 + lane] = a1;
  sm[(rl * 2 + 1) * 16 + lane] = a2;
  __syncthreads();
  if (rl < 2) {                                  // row lane 0 adds the sums, row lane 1 the sums of squares (fixed order)
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm[(q * 2 + rl) * 16 + lane];
    *reinterpret_cast<f32x4*>(stats + ((int64_t)blockIdx.x * 2 + rl) * C + blockIdx.y * 64 + lane * 4) = t;
  }
}

// per-thread channel parameters of the backward form of the statistics (X6Epi, conv_x6.h): 4 channels from c on
struct StatCh { f32x4 mu, is, ga, be; bool bwd, from_y, has_add; int act; };
__device__ __forceinline__ StatCh stat_channels(const X6Epi& epi, int c) {
  StatCh p;
  p.bwd = epi.bn_x != nullptr;
  p.act = epi.act;
  p.from_y = p.bwd && epi.bn_y != nullptr && epi.act != 0;
  p.has_add = p.bwd && epi.add != nullptr;
  p.mu = f32x4{0.f, 0.f, 0.f, 0.f}; p.is = f32x4{1.f, 1.f, 1.f, 1.f}; p.ga = p.is; p.be = p.mu;
  if (p.bwd) {
    p.mu = *reinterpret_cast<const f32x4*>(epi.mean + c);
    const f32x4 va = *reinterpret_cast<const f32x4*>(epi.var + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) p.is[j] = 1.f / sqrtf(va[j] + epi.eps);
    if (epi.gamma) p.ga = *reinterpret_cast<const f32x4*>(epi.gamma + c);
    if (epi.beta) p.be = *reinterpret_cast<const f32x4*>(epi.beta + c);
  }
  return p;
}
// `at`: element offset of (row, first channel) in the (n, C) matrices bn_x / add / bn_y
__device__ __forceinline__ void stat_accumulate(const StatCh& p, const X6Epi& epi, const f32x4& a, int64_t at, f32x4& a1, f32x4& a2) {
  f32x4 xv = {0.f, 0.f, 0.f, 0.f}, av = xv, yv = xv;
  if (p.bwd) xv = *reinterpret_cast<const f32x4*>(epi.bn_x + at);
  if (p.has_add) av = *reinterpret_cast<const f32x4*>(epi.add + at);
  if (p.from_y) yv = *reinterpret_cast<const f32x4*>(epi.bn_y + at);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t1, t2;
    x6_epi_terms(p.bwd, p.bwd ? av[j] + a[j] : a[j], xv[j], p.mu[j], p.is[j], p.ga[j], p.be[j], p.act, p.from_y, yv[j], t1, t2);
    a1[j] += t1;
    a2[j] += t2;
  }
}

__global__ __launch_bounds__(256) void k_sum_parts_stats(const float* __restrict__ part, float* __restrict__ out, int64_t n, int C, int S,
                                                         int RB, X6Epi epi) {
  __shared__ f32x4 sm[16 * 2 * 16];
  const int lane = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  const int c = blockIdx.y * 64 + lane * 4;
  const StatCh p = stat_channels(epi, c);
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
  for (int64_t r = r0 + rl; r < r0 + RB && r < n; r += 16) {
    f32x4 a = *reinterpret_cast<const f32x4*>(part + r * C + c);
    for (int z = 1; z < S; ++z) a += *reinterpret_cast<const f32x4*>(part + ((int64_t)z * n + r) * C + c);
    *reinterpret_cast<f32x4*>(out + r * C + c) = a;
    stat_accumulate(p, epi, a, r * C + c, a1, a2);
  }
  stat_block_reduce(a1, a2, epi.stats, C, sm);
}

__global__ __launch_bounds__(256) void k_sum_pairs_stats(const float* __restrict__ part, const int* __restrict__ pos, float* __restrict__ out,
                                                         int64_t n_out, int K, int Cout, int RB, X6Epi epi) {
  __shared__ f32x4 sm[16 * 2 * 16];
  const int lane = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  const int c = blockIdx.y * 64 + lane * 4;
  const StatCh p = stat_channels(epi, c);
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
  const int64_t rend = (r0 + RB < n_out) ? r0 + RB : n_out;
  // TWO rows of this thread per pass (rows o and o + 16): their positions, then their partial rows, are requested together — with
  // 64-row blocks (1k...4k result rows) a thread owns four rows, and one row per pass made them a chain of 4 x 3 x 2 dependent loads
  for (int64_t o = r0 + rl; o < rend; o += 32) {
    const int64_t o2 = o + 16;
    const bool two = o2 < rend;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;              // as k_sum_pairs: offsets in order, nine in flight
    for (int k0 = 0; k0 < K; k0 += 9) {
      int j[9], j2[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        j[u] = k0 + u < K ? pos[(int64_t)(k0 + u) * n_out + o] : -1;
        j2[u] = (two && k0 + u < K) ? pos[(int64_t)(k0 + u) * n_out + o2] : -1;
      }
      f32x4 v[9], v2[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const float* src = j[u] >= 0 ? part + ((int64_t)(k0 + u) * n_out + j[u]) * Cout + c : g_zero_row + lane * 4;
        const float* src2 = j2[u] >= 0 ? part + ((int64_t)(k0 + u) * n_out + j2[u]) * Cout + c : g_zero_row + lane * 4;
        v[u] = *reinterpret_cast<const f32x4*>(src);
        v2[u] = *reinterpret_cast<const f32x4*>(src2);
      }
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        if (j[u] >= 0) a += v[u];
        if (j2[u] >= 0) b += v2[u];
      }
    }
    *reinterpret_cast<f32x4*>(out + o * Cout + c) = a;
    stat_accumulate(p, epi, a, o * Cout + c, a1, a2);
    if (two) {
      *reinterpret_cast<f32x4*>(out + o2 * Cout + c) = b;
      stat_accumulate(p, epi, b, o2 * Cout + c, a1, a2);
    }
  }
  stat_block_reduce(a1, a2, epi.stats, Cout, sm);
}

// ------------------------------------------------------------------------------------------------
// Stem convolution (Cin = 3 colour channels -> 64, k3 s2; me_resnet.py:19-21): bound by the 148 MB output
// write.  64 output rows per workgroup; the 27x3 gathered inputs of every row and the whole (81,64) kernel sit in
// LDS, zero-padded to a reduction depth of 96, and each wave multiplies one 32x32 tile on the matrix cores.
#define STEM_CIN 3
#define STEM_COUT 64
#define STEM_ROWS 64
#define STEM_FWD_LDA 97
#define STEM_COL_LD 84      // row stride of the saved gathered inputs: 27 x 3 = 81 floats, padded to 21 float4
__global__ __launch_bounds__(256) void k_stem_fwd(const float* __restrict__ in, const float* __restrict__ W,
                                                  const int* __restrict__ nbr, float* __restrict__ out,
                                                  float* __restrict__ col, int64_t n_out, int K) {
  // out[64 rows][64] = A[64][81 -> 96] x W[96][64] on the matrix cores: one 32x32 tile per wave, 48 MFMA steps.
  extern __shared__ float sm[];
  constexpr int SLD = STEM_FWD_LDA;          // odd row stride -> conflict-free column reads of A
  const int KC = K * STEM_CIN;
  float* in_s = sm;                          // [STEM_ROWS][SLD]   (columns >= KC are zero)
  float* W_s = sm + STEM_ROWS * SLD;         // [96][64]           (rows >= KC are zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * STEM_ROWS;
  for (int t = tid; t < STEM_ROWS * (SLD - KC); t += 256) {
    int row = t / (SLD - KC), c = KC + t % (SLD - KC);
    in_s[row * SLD + c] = 0.f;
  }
  for (int t = tid; t < STEM_ROWS * K; t += 256) {
    int k = t / STEM_ROWS, row = t % STEM_ROWS;
    int64_t o = m0 + row;
    int64_t oc = o < n_out ? o : n_out - 1;
    int i = nbr[(int64_t)k * n_out + oc];
    if (o >= n_out) i = -1;
    const float* src = i < 0 ? g_zero_row : in + (int64_t)i * 3;
    float a = src[0], b = src[1], c = src[2];       // (r2: one 12-byte load instead of three dwords is no faster)
    in_s[row * SLD + k * 3] = a; in_s[row * SLD + k * 3 + 1] = b; in_s[row * SLD + k * 3 + 2] = c;
  }
  for (int t = tid; t < 96 * 16; t += 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t < KC * 16) v = reinterpret_cast<const f32x4*>(W)[t];
    reinterpret_cast<f32x4*>(W_s)[t] = v;
  }
  __syncthreads();
  if (col) {
    // training: the gathered inputs of the tile go out as rows of `col` (n_out, 84) — the weight gradient then streams them
    // instead of repeating 27 scattered 12-byte reads per output row (r2: 390 -> ~100 us; the forward pays one 195 MB write)
    for (int t = tid; t < STEM_ROWS * (STEM_COL_LD / 4); t += 256) {
      const int row = t / (STEM_COL_LD / 4), c = (t % (STEM_COL_LD / 4)) * 4;
      const int64_t o = m0 + row;
      if (o < n_out) {
        const float* sp = in_s + row * SLD + c;
        f32x4 v = {sp[0], sp[1], sp[2], sp[3]};
        *reinterpret_cast<f32x4*>(col + o * STEM_COL_LD + c) = v;
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* ap = in_s + (wr * 32 + r) * SLD + h;
  const float* bp = W_s + h * 64 + wc * 32 + r;
#pragma unroll 8
  for (int sidx = 0; sidx < 48; ++sidx)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * sidx], bp[2 * sidx * 64], acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    int64_t o = m0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (o < n_out) out[o * 64 + wc * 32 + r] = acc[e];
  }
}

// stem wgrad on the matrix cores: part[block][j][co] = sum over the block's rows of A[row][j] * gout[row][co], with
// A[row][j = 3k+c] = in[nbr[k][row]][c] staged in LDS (81 real columns padded to 96 = 3 MFMA tiles, gout 64 = 2 tiles).
// Each of the 4 waves owns a quarter of every 64-row chunk as its reduction slice and all 6 tiles; the four partial
// tiles are summed through LDS in wave order (deterministic) at the end.
#define STEM_JP 96
__global__ __launch_bounds__(256) void k_stem_wgrad(const float* __restrict__ in, const float* __restrict__ gout,
                                                    const int* __restrict__ nbr, float* __restrict__ part, int64_t n_out,
                                                    int K, int64_t rows_per_block) {
  extern __shared__ float sm[];
  const int KC = K * STEM_CIN;
  float* in_s = sm;                          // [STEM_ROWS][STEM_JP]   (columns >= KC stay zero)
  float* g_s = sm + STEM_ROWS * STEM_JP;     // [STEM_ROWS][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > n_out) r_end = n_out;
  for (int t = tid; t < STEM_ROWS * STEM_JP; t += 256) in_s[t] = 0.f;
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // register-prefetch pipeline (as k_conv_mfma): the 27 x 3 gathered inputs and the gout rows of chunk t+1 are in flight
  // while chunk t is multiplied (r2: without it the index -> gather chain of every 64-row chunk was exposed: 0.6 TB/s)
  constexpr int NG = (STEM_ROWS * 27 + 255) / 256;          // gathers per thread per chunk (K <= 27)
  float pa[NG][3];
  f32x4 pg[4];
  auto load_chunk = [&](int64_t rb) {
    int idx[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int t = tid + 256 * u;
      const int k = t / STEM_ROWS, row = t % STEM_ROWS;
      const int64_t o = rb + row;
      const int64_t oc = o < r_end ? o : r_end - 1;
      const int kk = k < K ? k : K - 1;
      int i = nbr[(int64_t)kk * n_out + oc];
      idx[u] = (o < r_end && k < K) ? i : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = tid + 256 * u;
      const int64_t o = rb + (t >> 4);
      const float* gp = o < r_end ? gout + o * 64 + (t & 15) * 4 : g_zero_row;
      pg[u] = *reinterpret_cast<const f32x4*>(gp);
    }
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const float* src = idx[u] < 0 ? g_zero_row : in + (int64_t)idx[u] * 3;
      pa[u][0] = src[0]; pa[u][1] = src[1]; pa[u][2] = src[2];
    }
  };
  if (r_begin < r_end) load_chunk(r_begin);
  for (int64_t rb = r_begin; rb < r_end; rb += STEM_ROWS) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int t = tid + 256 * u;
      const int k = t / STEM_ROWS, row = t % STEM_ROWS;
      if (k < K) {
        in_s[row * STEM_JP + k * 3] = pa[u][0]; in_s[row * STEM_JP + k * 3 + 1] = pa[u][1]; in_s[row * STEM_JP + k * 3 + 2] = pa[u][2];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) reinterpret_cast<f32x4*>(g_s)[tid + 256 * u] = pg[u];
    __syncthreads();
    load_chunk(rb + STEM_ROWS < r_end ? rb + STEM_ROWS : rb);          // unconditional (see k_conv_mfma)
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
      const int row = wave * 16 + 2 * sidx + h;
      float a[3], b[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = in_s[row * STEM_JP + i * 32 + r];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = g_s[row * 64 + j * 32 + r];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // cross-wave sum in wave order through LDS ([96][64] floats = 24 KB, reusing the staging area)
  __syncthreads();
  float* red = sm;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            int jj = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, co = j * 32 + r;
            float v = acc[i][j][e];
            if (w > 0) v += red[jj * 64 + co];
            red[jj * 64 + co] = v;
          }
    }
    __syncthreads();
  }
  float* dst = part + (int64_t)blockIdx.x * KC * 64;
  for (int t = tid; t < KC * 64; t += 256) dst[t] = red[t];
}

// The same reduction with the gathered inputs read back from `col` (k_stem_fwd) — no neighbour table, no gathers.
__global__ __launch_bounds__(256) void k_stem_wgrad_col(const float* __restrict__ col, const float* __restrict__ gout,
                                                        float* __restrict__ part, int64_t n_out, int K, int64_t rows_per_block) {
  extern __shared__ float sm[];
  const int KC = K * STEM_CIN;
  float* in_s = sm;                          // [STEM_ROWS][STEM_JP]   (columns >= KC stay zero)
  float* g_s = sm + STEM_ROWS * STEM_JP;     // [STEM_ROWS][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > n_out) r_end = n_out;
  for (int t = tid; t < STEM_ROWS * STEM_JP; t += 256) in_s[t] = 0.f;
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // the gathered inputs come as rows of `col` (saved by k_stem_fwd): two coalesced streams, prefetched one chunk ahead
  constexpr int NC = (STEM_ROWS * (STEM_COL_LD / 4) + 255) / 256;      // float4 per thread per chunk
  f32x4 pc[NC], pg[4];
  auto load_chunk = [&](int64_t rb) {
#pragma unroll
    for (int u = 0; u < NC; ++u) {
      const int t = tid + 256 * u;
      const int row = t / (STEM_COL_LD / 4), c = (t % (STEM_COL_LD / 4)) * 4;
      const int64_t o = rb + row;
      const float* cp = (o < r_end && row < STEM_ROWS) ? col + o * STEM_COL_LD + c : g_zero_row;
      pc[u] = *reinterpret_cast<const f32x4*>(cp);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = tid + 256 * u;
      const int64_t o = rb + (t >> 4);
      const float* gp = o < r_end ? gout + o * 64 + (t & 15) * 4 : g_zero_row;
      pg[u] = *reinterpret_cast<const f32x4*>(gp);
    }
  };
  if (r_begin < r_end) load_chunk(r_begin);
  for (int64_t rb = r_begin; rb < r_end; rb += STEM_ROWS) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NC; ++u) {
      const int t = tid + 256 * u;
      const int row = t / (STEM_COL_LD / 4), c = (t % (STEM_COL_LD / 4)) * 4;
      if (row < STEM_ROWS) *reinterpret_cast<f32x4*>(&in_s[row * STEM_JP + c]) = pc[u];      // columns 84..95 stay zero
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) reinterpret_cast<f32x4*>(g_s)[tid + 256 * u] = pg[u];
    __syncthreads();
    load_chunk(rb + STEM_ROWS < r_end ? rb + STEM_ROWS : rb);          // unconditional (see k_conv_mfma)
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
      const int row = wave * 16 + 2 * sidx + h;
      float a[3], b[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = in_s[row * STEM_JP + i * 32 + r];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = g_s[row * 64 + j * 32 + r];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // cross-wave sum in wave order through LDS ([96][64] floats = 24 KB, reusing the staging area)
  __syncthreads();
  float* red = sm;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            int jj = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, co = j * 32 + r;
            float v = acc[i][j][e];
            if (w > 0) v += red[jj * 64 + co];
            red[jj * 64 + co] = v;
          }
    }
    __syncthreads();
  }
  float* dst = part + (int64_t)blockIdx.x * KC * 64;
  for (int t = tid; t < KC * 64; t += 256) dst[t] = red[t];
}

// generic fallback (any Cin/Cout): one thread per (row, cout).  Used for the Cin=3 stem and as the
// cross-check path of the parity tests (flags & 1).
__global__ void k_conv_fma(const float* __restrict__ in, const float* __restrict__ W, const int* __restrict__ nbr,
                           float* __restrict__ out, int64_t n_out, int K, int Cin, int Cout, int wt) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * Cout) return;
  int64_t o = t / Cout;
  int co = (int)(t % Cout);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    int i = nbr ? nbr[(int64_t)k * n_out + o] : (int)o;
    if (i < 0) continue;
    const float* x = in + (int64_t)i * Cin;
    const float* w = W + (int64_t)k * Cin * Cout + (wt ? (int64_t)co * Cin : co);
    const int64_t ws = wt ? 1 : Cout;             // wt: W[k] stored (Cout, Cin)
    for (int ci = 0; ci < Cin; ++ci) acc = fmaf(x[ci], w[ci * ws], acc);
  }
  out[t] = acc;
}

extern "C" {

#define FC_CONV_WT (1 << 23)   // fc_conv_fwd / fc_conv_fwd_pairs(_tiles): W[k] is stored transposed, (Cout, Cin) row-major
// The deeper-pipelined LDS kernel (k_conv_mfma_p) holds 3 workgroups per CU (768 slots) where k_conv_mfma holds 4 (1024): it
// wins on launches of many rounds and on launches that fit 768 slots anyway, and loses a round in between (r2: +5.5 / +7 %
// on the 441k / 55k-row levels, +5 % on the 862-row pair mode, -12 % on the 3.5k-row pair mode with its 972 workgroups).
// flags bit18 forces it on, bit17 off.
// flags bit21: the LDS-DMA kernel (k_conv_glds, 2 workgroups per CU) instead.  Returns 0 / 1 / 2 = k_conv_mfma / _p / k_conv_glds.
static inline int conv_pipe(int flags, dim3 grid) {
  if (flags & (1 << 24)) return (flags & (1 << 26)) ? 4 : 3;     // split-bf16 kernel (conv_x6.h); bit26: W is a pre-split image
  if (flags & (1 << 21)) return 2;
  if (flags & (1 << 18)) return 1;
  if (flags & (1 << 17)) return 0;
  const int64_t wgs = (int64_t)grid.x * grid.y * grid.z;
  if (wgs >= 1536 || wgs <= 768) return 1;
  // in between: offset-split launches of a dense table go to the LDS-DMA kernel (r2 nbench, same box: 6.9k rows 256->256
  // 251 -> 232 us, 256->128 139 -> 128 us, 14.9k rows 128->128 163 -> 146 us; unsplit and pair-list launches: neutral)
  return (grid.z > 1 && !(flags & (1 << 22))) ? 2 : 0;
}

static void conv_plan(int64_t n_out, int K, int Cin, int Cout, int flags, bool* mfma, int* bm, int* bn, int* S) {
  *mfma = !(flags & 1) && (Cin % BK == 0) && (Cout % 64 == 0) && K <= 32;
  *bn = (Cout % 128 == 0) ? 128 : 64;
  const int64_t wg128 = fc_cdiv(n_out, 128) * (Cout / *bn);
  // measured on the benchmark's layers (tools/convbench.py): 128-row tiles win at every size once the grid
  // is topped up to ~1024 workgroups by splitting over kernel offsets
  *bm = (n_out > 64 && wg128 >= 4) ? 128 : 64;
  // 64-wide outputs on big maps: 256 x 64 tiles, 4 waves along the rows (r2: +6 % on the 441k-row level, 88 / 95 TF)
  if (*bn == 64 && Cout == 64 && fc_cdiv(n_out, 256) >= 1024 && !(flags & (1 << 24))) *bm = 256;      // (split-bf16: 128 x 64 at 4 waves / SIMD is ahead, 613 vs 628 us)
  const int64_t tiles = fc_cdiv(n_out, *bm) * (Cout / *bn);
  int s = 1;
  // split over kernel offsets: the LARGEST split that still fits one resident round (1024 workgroup slots) — one workgroup
  // over (r1 rounded up) starts a second, nearly empty round (r2 sweep: 256->256 on 6.9k rows 275 us at S = 10 -> 238 at
  // S = 9); from ~400 tiles on the unsplit launch wins (64->64 on 64k rows: 159 us at S = 2 -> 144 at S = 1, and no
  // partial tiles to write and sum)
  if (*mfma && K > 1 && tiles < 384) {          // (r5 sweep with the split-bf16 kernels: 256 / 512 / 768 change nothing, 368.1-368.5 scenes/s)
    s = (int)(1024 / tiles);
    if (s > K) s = K;
    if (s < 1) s = 1;
  }
  // tuning overrides: flags[4:5] BM (1=64, 2=128), flags[6:7] BN (1=64, 2=128), flags[8:15] S
  int fbm = (flags >> 4) & 3, fbn = (flags >> 6) & 3, fs = (flags >> 8) & 255;
  if (fbm) *bm = fbm == 1 ? 64 : (fbm == 2 ? 128 : 256);
  if (fbn && (Cout % (fbn == 1 ? 64 : 128) == 0)) *bn = fbn == 1 ? 64 : 128;
  if (*bm == 256) *bn = 64;                      // the 4 x 1 wave arrangement: 256 x 64 tiles
  if (fbm || fbn) {
    const int64_t t2 = fc_cdiv(n_out, *bm) * (Cout / *bn);
    s = 1;
    if (*mfma && K > 1 && t2 < 384) { s = (int)(1024 / t2); if (s > K) s = K; if (s < 1) s = 1; }
  }
  if (fs) s = fs > K ? K : fs;
  if ((flags & (1 << 24)) && *mfma && *bm == 64) {     // the split-bf16 kernel has 128- and 256-row tiles only
    *bm = 128;
    const int64_t t3 = fc_cdiv(n_out, 128) * (Cout / *bn);
    if (!fs) { s = 1; if (K > 1 && t3 < 384) { s = (int)(1024 / t3); if (s > K) s = K; } }
  }
  *S = s;
}

// one launch of the LDS-tiled MFMA kernel (32-deep slabs; measured r1: 64-deep slabs, 256-row tiles, an LDS index table
// and LDS-padding occupancy caps all lose or are neutral — profiles/r1_conv_pmc.md — and were removed in r2)
static int launch_conv_mfma(int pipe, int bm, int bn, dim3 grid, const float* in, const float* W, const int* nbr,
                            const int* out_index, const int* cnt, float* dst, int64_t n_rows, int K, int Cin, int Cout,
                            hipStream_t stream, bool wt = false, const X6Epi* epi = nullptr, int64_t n_in = -1,
                            int64_t n_in_rows = -1) {
  if (epi && (pipe < 3 || bm < 128 || cnt || grid.z != 1)) return FC_EINVAL;      // the statistics epilogue lives in k_conv_x6
  // buffer addressing (k_conv_x6 BUF; gathering launches on a weight image): the gathered operand must end below the 2 GB its
  // descriptor spans; n_in < 0: the caller asks for (flags bit27) or only knows flat addresses
  const bool bufok = nbr && n_in >= 0 && (uint64_t)n_in * (uint64_t)Cin * 4u < (1ull << 31) - 4096u &&
                     (uint64_t)K * (uint64_t)n_rows * 4u < (1ull << 31) - 4096u;      // (r6: the neighbour table goes through a descriptor too)
  X6Epi e6 = {};
  if (epi) e6 = *epi;
  const bool h3 = pipe == 4 && bm >= 128 && split_mode() == 2;
  if (h3) {                                      // the gathered operand's amax word: the caller's hint or a pass of our own
    if (n_in_rows < 0) return FC_EINVAL;
    int rc = operand_amax(in, n_in_rows * (int64_t)Cin, 0, stream, &e6.amax_in);
    if (rc != FC_OK) return rc;
  }
#define FC_ARGS <<<grid, 256, 0, stream>>>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout)
#define FC_LAUNCH_MFMA(KERNEL, BM_, BN_, WM_)                                    \
  do {                                                                           \
    if (wt) KERNEL<BM_, BN_, 32, true, WM_, true> FC_ARGS;                        \
    else if (nbr) KERNEL<BM_, BN_, 32, true, WM_> FC_ARGS;                        \
    else KERNEL<BM_, BN_, 32, false, WM_> FC_ARGS;                                \
  } while (0)
#define FC_LAUNCH_GLDS(BM_, BN_, WM_)                                            \
  do {                                                                           \
    if (nbr) k_conv_glds<BM_, BN_, true, WM_> FC_ARGS;                            \
    else k_conv_glds<BM_, BN_, false, WM_> FC_ARGS;                               \
  } while (0)
  if (pipe == 4) wt = false;                     // a weight image already is the operator of its direction
  if (wt && !nbr) return FC_EINVAL;              // transposed weights: neighbour-table / pair-list launches only
  if (wt && pipe == 2) pipe = 0;                 // the LDS-DMA image cannot be transposed in flight
  // r6: h3 launches on 128 x 128 tiles take the register-operand kernel (conv_h3r.h): +1...11 % per launch there (tools/nbench, same
  // box), while the 64-column tiles LOSE 7-14 % on the 441k-row maps — a lane-per-row load touches 32 cache lines per instruction
  // where the LDS staging touches 8, and those launches are bound by the gather.  FC_H3R / fc_debug_set_h3r: 0 never, 1 (default)
  // 128-column tiles, 2 every 128-row tile.
  if (g_h3r < 0) g_h3r = getenv("FC_H3R") ? atoi(getenv("FC_H3R")) : 1;
  if (h3 && bm == 128 && !g_bf16_fast && (g_h3r == 2 || (g_h3r == 1 && bn == 128))) {
#define FC_LAUNCH_H3R(BN_)                                                                                               \
  do {                                                                                                                  \
    if (nbr && bufok) k_conv_h3r<BN_, true, true><<<grid, 256, 0, stream>>>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, e6);       \
    else if (nbr) k_conv_h3r<BN_, true, false><<<grid, 256, 0, stream>>>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, e6);          \
    else k_conv_h3r<BN_, false, false><<<grid, 256, 0, stream>>>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, e6);                  \
  } while (0)
    if (bn == 128) FC_LAUNCH_H3R(128); else FC_LAUNCH_H3R(64);
#undef FC_LAUNCH_H3R
    FC_CHECK_LAUNCH();
    return FC_OK;
  }
  if (pipe >= 3 && bm >= 128) {                  // split-bf16 kernel; pipe 4: the weights are a pre-split image
#define FC_ARGS6 <<<grid, 256, 0, stream>>>(in, W, nbr, out_index, cnt, dst, n_rows, K, Cin, Cout, e6)
#define FC_LAUNCH_X6(BM_, BN_, WM_)                                              \
  do {                                                                           \
    if (pipe == 4 && g_bf16_fast && BM_ == 128 && nbr) k_conv_x6<128, BN_, true, 2, 2, 1> FC_ARGS6;   \
    else if (pipe == 4 && g_bf16_fast && BM_ == 128) k_conv_x6<128, BN_, false, 2, 2, 1> FC_ARGS6;  \
    else if (pipe == 4 && h3 && bufok && BM_ == 128) k_conv_x6<128, BN_, true, 2, 2, 2, true> FC_ARGS6;  \
    else if (pipe == 4 && h3 && nbr) k_conv_x6<BM_, BN_, true, WM_, 2, 2> FC_ARGS6;  \
    else if (pipe == 4 && h3) k_conv_x6<BM_, BN_, false, WM_, 2, 2> FC_ARGS6;  \
    else if (pipe == 4 && bufok && BM_ == 128) k_conv_x6<128, BN_, true, 2, 2, 0, true> FC_ARGS6;  \
    else if (pipe == 4 && nbr) k_conv_x6<BM_, BN_, true, WM_, 2> FC_ARGS6;       \
    else if (pipe == 4) k_conv_x6<BM_, BN_, false, WM_, 2> FC_ARGS6;        \
    else if (wt) k_conv_x6<BM_, BN_, true, WM_, 1> FC_ARGS6;                \
    else if (nbr) k_conv_x6<BM_, BN_, true, WM_, 0> FC_ARGS6;               \
    else k_conv_x6<BM_, BN_, false, WM_, 0> FC_ARGS6;                       \
  } while (0)
    if (bm == 256) FC_LAUNCH_X6(256, 64, 4);
    else if (bn == 128) FC_LAUNCH_X6(128, 128, 2);
    else FC_LAUNCH_X6(128, 64, 2);
#undef FC_LAUNCH_X6
#undef FC_ARGS6
    FC_CHECK_LAUNCH();
    return FC_OK;
  }
  if (pipe >= 3) pipe = 1;                       // 64-row tiles: fp32 pipe
  if (pipe == 2 && bm >= 128) {
    if (bm == 256) FC_LAUNCH_GLDS(256, 64, 4);
    else if (bn == 128) FC_LAUNCH_GLDS(128, 128, 2);
    else FC_LAUNCH_GLDS(128, 64, 2);
  } else if (pipe && bm == 256) FC_LAUNCH_MFMA(k_conv_mfma_p, 256, 64, 4);
  else if (pipe && bm == 128 && bn == 128) FC_LAUNCH_MFMA(k_conv_mfma_p, 128, 128, 2);
  else if (pipe && bm == 128 && bn == 64) FC_LAUNCH_MFMA(k_conv_mfma_p, 128, 64, 2);
  else if (bm == 256) FC_LAUNCH_MFMA(k_conv_mfma, 256, 64, 4);
  else if (bm == 128 && bn == 128) FC_LAUNCH_MFMA(k_conv_mfma, 128, 128, 2);
  else if (bm == 128) FC_LAUNCH_MFMA(k_conv_mfma, 128, 64, 2);
  else if (bn == 128) FC_LAUNCH_MFMA(k_conv_mfma, 64, 128, 2);
  else FC_LAUNCH_MFMA(k_conv_mfma, 64, 64, 2);
#undef FC_LAUNCH_GLDS
#undef FC_LAUNCH_MFMA
#undef FC_ARGS
  FC_CHECK_LAUNCH();
  return FC_OK;
}

static inline bool is_stem(const int* nbr, int K, int Cin, int Cout, int flags) {
  return !(flags & 1) && !(flags & FC_CONV_WT) && nbr && Cin == STEM_CIN && Cout == STEM_COUT && K <= 27;
}

int64_t fc_conv_fwd_ws_bytes(int64_t n_out, int K, int Cin, int Cout, int flags) {
  bool mfma; int bm, bn, S;
  conv_plan(n_out > 0 ? n_out : 1, K, Cin, Cout, flags, &mfma, &bm, &bn, &S);
  return S > 1 ? (int64_t)S * n_out * Cout * (int64_t)sizeof(float) : 0;
}

static int sum_parts(const float* part, float* out, int64_t n_out, int Cout, int S, hipStream_t stream) {
  int64_t e4 = n_out * Cout / 4;
  k_sum_parts<<<(unsigned)fc_cdiv(e4, 256), 256, 0, stream>>>(part, out, e4, S);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

static int sum_parts_stats(const float* part, float* out, int64_t n_out, int Cout, int S, const X6Epi& epi, hipStream_t stream) {
  const int rb = fc_stat_rb(n_out);
  k_sum_parts_stats<<<dim3((unsigned)fc_cdiv(n_out, rb), Cout / 64), 256, 0, stream>>>(part, out, n_out, Cout, S, rb, epi);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

static int conv_fwd_impl(const float* in, const float* W, const int* nbr, const int* out_index,
                         float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int flags, void* ws,
                         int64_t ws_bytes, hipStream_t stream, const X6Epi* epi = nullptr) {
  AmaxHintScope hint_scope;
  const bool stats = epi != nullptr;
  if (n_in < 0 || n_out < 0 || K < 1 || Cin < 1 || Cout < 1) return FC_EINVAL;
  if (n_out == 0) return FC_OK;                 // nothing to write (an empty table may well be a NULL pointer)
  if (!nbr && (K != 1 || n_in != n_out)) return FC_EINVAL;
  if (out_index && !nbr) return FC_EINVAL;
  if (is_stem(nbr, K, Cin, Cout, flags) && !out_index) {
    size_t smem = (size_t)(STEM_ROWS * STEM_FWD_LDA + 96 * 64) * sizeof(float);
    k_stem_fwd<<<(unsigned)fc_cdiv(n_out, STEM_ROWS), 256, smem, stream>>>(in, W, nbr, out, nullptr, n_out, K);
    FC_CHECK_LAUNCH();
    return FC_OK;
  }
  const bool wt = (flags & FC_CONV_WT) != 0;    // W[k] given as (Cout, Cin): the backward-data pass on the layer's own kernel
  if (wt && !nbr) return FC_EINVAL;
  bool mfma_ok; int bm, bn, S;
  conv_plan(n_out, K, Cin, Cout, flags, &mfma_ok, &bm, &bn, &S);
  if (stats && (!mfma_ok || !(flags & (1 << 24)))) return FC_EINVAL;      // see fc_conv_stats_blocks
  if (!mfma_ok) {
    if (out_index || (flags & (1 << 26))) return FC_EINVAL;      // sorted-row tables and weight images are MFMA-path features
    k_conv_fma<<<(unsigned)fc_cdiv(n_out * Cout, 256), 256, 0, stream>>>(in, W, nbr, out, n_out, K, Cin, Cout, wt ? 1 : 0);
    FC_CHECK_LAUNCH();
    return FC_OK;
  }
  if (S > 1 && ws_bytes < (int64_t)S * n_out * Cout * (int64_t)sizeof(float)) return FC_EWS;
  float* dst = S > 1 ? (float*)ws : out;
  dim3 grid((unsigned)fc_cdiv(n_out, bm), Cout / bn, S);
  int rc = launch_conv_mfma(conv_pipe(flags, grid), bm, bn, grid, in, W, nbr, out_index, nullptr, dst, n_out, K, Cin, Cout, stream, wt,
                            S > 1 ? nullptr : epi, (flags & (1 << 27)) ? -1 : n_in, n_in);
  if (rc != FC_OK) return rc;
  if (S > 1) return stats ? sum_parts_stats(dst, out, n_out, Cout, S, *epi, stream) : sum_parts(dst, out, n_out, Cout, S, stream);
  return FC_OK;
}

// Row blocks of the statistics table a convolution launch leaves for the BatchNorm behind it (fc_conv_fwd_stats /
// fc_conv_fwd_pairs_tiles_stats: stats[blocks][2][Cout], column sums of the result and of its square per row block); 0: this launch
// has no statistics epilogue (not the split-bf16 MFMA route).  pairs != 0: the per-offset pair-list route.
int64_t fc_conv_stats_blocks(int64_t n_out, int K, int Cin, int Cout, int flags, int pairs) {
  if (n_out < 1 || !(flags & (1 << 24))) return 0;
  if (pairs) return (Cin % 32 == 0 && Cout % 64 == 0) ? fc_cdiv(n_out, fc_stat_rb(n_out)) : 0;
  bool mfma_ok; int bm, bn, S;
  conv_plan(n_out, K, Cin, Cout, flags, &mfma_ok, &bm, &bn, &S);
  if (!mfma_ok || bm < 128) return 0;
  return S > 1 ? fc_cdiv(n_out, fc_stat_rb(n_out)) : fc_cdiv(n_out, bm);
}

int fc_conv_fwd_stats(const float* in, const float* W, const int* nbr, const int* out_index, float* out, int64_t n_in,
                      int64_t n_out, int K, int Cin, int Cout, int flags, void* ws, int64_t ws_bytes, float* stats,
                      hipStream_t stream) {
  if (stats && fc_conv_stats_blocks(n_out, K, Cin, Cout, flags, 0) == 0) return FC_EINVAL;
  X6Epi e = {};
  e.stats = stats;
  return conv_fwd_impl(in, W, nbr, out_index, out, n_in, n_out, K, Cin, Cout, flags, ws, ws_bytes, stream, stats ? &e : nullptr);
}

// The backward-data pass of a convolution whose INPUT came out of a BatchNorm (+ ReLU / ELU) layer: besides the gradient g it
// writes, the launch leaves that layer's two backward reductions per row block — stats[blocks][2][Cout] = column sums of
// g' = g act'(.) and of g' xhat, xhat from the layer's input bn_x (n_out, Cout) and its batch statistics — for
// fc_bn_train_bwd(part = stats).  act: 0 none, 1 ReLU, 2 ELU; add (nullable): a second contribution to the gradient, g = result + add;
// bn_y (nullable): the layer's output, which act'(.) is taken from (a layer WITH a residual; NULL: recomputed from bn_x).
int fc_conv_fwd_bn_bwd_stats(const float* in, const float* W, const int* nbr, const int* out_index, float* out, int64_t n_in,
                             int64_t n_out, int K, int Cin, int Cout, int flags, void* ws, int64_t ws_bytes, float* stats,
                             const float* bn_x, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                             int act, const float* add, const float* bn_y, hipStream_t stream) {
  if (!stats || !bn_x || !mean || !var || act < 0 || act > 2) return FC_EINVAL;
  if (fc_conv_stats_blocks(n_out, K, Cin, Cout, flags, 0) == 0) return FC_EINVAL;
  X6Epi e = {stats, bn_x, mean, var, gamma, beta, eps, act, add, bn_y};
  return conv_fwd_impl(in, W, nbr, out_index, out, n_in, n_out, K, Cin, Cout, flags, ws, ws_bytes, stream, &e);
}

// flags: bit0 = force the generic FMA kernel.
int fc_conv_fwd(const float* in, const float* W, const int* nbr, const int* out_index, float* out, int64_t n_in,
                int64_t n_out, int K, int Cin, int Cout, int flags, void* ws, int64_t ws_bytes, hipStream_t stream) {
  return conv_fwd_impl(in, W, nbr, out_index, out, n_in, n_out, K, Cin, Cout, flags, ws, ws_bytes, stream);
}

// Pre-split weight image for the split-bf16 kernel (flags bit24 | bit26 of fc_conv_fwd / fc_conv_fwd_pairs*): R = reduction
// size (Cin of the launch), C = its columns (Cout of the launch); transposed != 0: W[k] is stored (C, R) row-major.
int64_t fc_x6_weight_image_bytes(int K, int R, int C) { return (int64_t)K * R * C * 6; }

int fc_x6_weight_image(const float* W, void* img, int K, int R, int C, int transposed, hipStream_t stream) {
  if (K < 1 || R < 32 || C < 64 || R % 32 != 0 || C % 64 != 0) return FC_EINVAL;
  const int64_t total = (int64_t)K * (R / 32) * (C / 64) * 256;
  const unsigned nb = (unsigned)fc_cdiv(total, 256);
  if (split_mode() == 2) {                       // h3: max |W| into the image's amax word, then the fp16 pieces
    FC_HIP(hipMemsetAsync((char*)img + 4 * X6_IMG_AMAX_WORD, 0, FC_AMAX_SLOT_BYTES, stream));
    k_x6_weight_image<3><<<nb, 256, 0, stream>>>(W, (u32x4*)img, K, R, C, transposed);
    k_x6_weight_image<2><<<nb, 256, 0, stream>>>(W, (u32x4*)img, K, R, C, transposed);
  } else {
    k_x6_weight_image<0><<<nb, 256, 0, stream>>>(W, (u32x4*)img, K, R, C, transposed);
  }
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_x6_weight_images(const int64_t* desc, int n, int64_t total_blocks, hipStream_t stream) {
  if (!desc || n < 1 || total_blocks < 1 || total_blocks > 0x7fffffffll) return FC_EINVAL;
  const long long* d = reinterpret_cast<const long long*>(desc);
  if (split_mode() == 2) {
    k_x6_weight_images<4><<<(unsigned)fc_cdiv(n, 4), 256, 0, stream>>>(d, n);
    k_x6_weight_images<3><<<(unsigned)total_blocks, 256, 0, stream>>>(d, n);
    k_x6_weight_images<2><<<(unsigned)total_blocks, 256, 0, stream>>>(d, n);
  } else {
    k_x6_weight_images<0><<<(unsigned)total_blocks, 256, 0, stream>>>(d, n);
  }
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int64_t fc_conv_fwd_pairs_ws_bytes(int64_t n_out, int K, int Cout) {
  return (int64_t)K * n_out * Cout * (int64_t)sizeof(float);
}

// Convolution over the exact pair lists: per offset a compacted gather-GEMM into the workspace, then a gather-sum.
static int conv_fwd_pairs_impl(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                            float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int64_t live_tiles, int flags,
                            void* ws, int64_t ws_bytes, hipStream_t stream, const X6Epi* epi) {
  AmaxHintScope hint_scope;
  if (n_in < 0 || n_out < 0 || K < 1 || K > 65535 || Cin < 1 || Cout < 1) return FC_EINVAL;
  if (!pair_in || !pair_cnt || !pair_pos) return FC_EINVAL;
  if (Cin % 32 != 0 || Cout % 64 != 0) return FC_EINVAL;       // MFMA shapes only; callers use fc_conv_fwd otherwise
  if (n_out == 0) return FC_OK;
  if (ws_bytes < fc_conv_fwd_pairs_ws_bytes(n_out, K, Cout)) return FC_EWS;
  float* part = (float*)ws;
  const bool wt = (flags & FC_CONV_WT) != 0;
  // (r5, measured null: 64-column tiles for the few-thousand-row pair-list launches — 4 workgroups per CU, finer rounds — 373.6 /
  // 372.5 / 372.2 scenes/s at <= 1k / 4k / 16k rows against 375.4: profiles/r5_notes.md)
  const bool wide = (Cout % 128 == 0) && !(((flags >> 6) & 3) == 1);
  const int bn = wide ? 128 : 64;
  dim3 grid((unsigned)fc_cdiv(n_out, 128), Cout / bn, K);
  if (live_tiles > 0) grid = dim3((unsigned)live_tiles, Cout / bn, 1);       // linear list of the live (offset, tile) pairs
  {
    int rc = launch_conv_mfma((live_tiles > 0 || (flags & (1 << 24))) ? conv_pipe(flags, grid) : ((flags & (1 << 21)) ? 2 : ((flags & (1 << 18)) ? 1 : 0)), 128, bn, grid, in, W, pair_in, nullptr, pair_cnt, part, n_out, K, Cin, Cout, stream, wt, nullptr, (flags & (1 << 27)) ? -1 : n_in, n_in);
    if (rc != FC_OK) return rc;
  }
  if (epi) {
    const int rb = fc_stat_rb(n_out);
    k_sum_pairs_stats<<<dim3((unsigned)fc_cdiv(n_out, rb), Cout / 64), 256, 0, stream>>>(part, pair_pos, out, n_out, K, Cout, rb, *epi);
  } else {
    k_sum_pairs<<<(unsigned)fc_cdiv(n_out * (Cout / 4), 256), 256, 0, stream>>>(part, pair_pos, out, n_out, K, Cout);
  }
  FC_CHECK_LAUNCH();
  return FC_OK;
}

int fc_conv_fwd_pairs_tiles(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                            float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int64_t live_tiles, int flags,
                            void* ws, int64_t ws_bytes, hipStream_t stream) {
  return conv_fwd_pairs_impl(in, W, pair_in, pair_cnt, pair_pos, out, n_in, n_out, K, Cin, Cout, live_tiles, flags, ws, ws_bytes, stream,
                             nullptr);
}

int fc_conv_fwd_pairs_tiles_stats(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                                  float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int64_t live_tiles, int flags,
                                  void* ws, int64_t ws_bytes, float* stats, hipStream_t stream) {
  if (stats && !(flags & (1 << 24))) return FC_EINVAL;
  X6Epi e = {};
  e.stats = stats;
  return conv_fwd_pairs_impl(in, W, pair_in, pair_cnt, pair_pos, out, n_in, n_out, K, Cin, Cout, live_tiles, flags, ws, ws_bytes, stream,
                             stats ? &e : nullptr);
}

int fc_conv_fwd_pairs_tiles_bn_bwd_stats(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                                         float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int64_t live_tiles, int flags,
                                         void* ws, int64_t ws_bytes, float* stats, const float* bn_x, const float* mean,
                                         const float* var, const float* gamma, const float* beta, float eps, int act,
                                         const float* add, const float* bn_y, hipStream_t stream) {
  if (!stats || !bn_x || !mean || !var || act < 0 || act > 2 || !(flags & (1 << 24))) return FC_EINVAL;
  X6Epi e = {stats, bn_x, mean, var, gamma, beta, eps, act, add, bn_y};
  return conv_fwd_pairs_impl(in, W, pair_in, pair_cnt, pair_pos, out, n_in, n_out, K, Cin, Cout, live_tiles, flags, ws, ws_bytes, stream,
                             &e);
}

int fc_conv_fwd_pairs(const float* in, const float* W, const int* pair_in, const int* pair_cnt, const int* pair_pos,
                      float* out, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int flags, void* ws,
                      int64_t ws_bytes, hipStream_t stream) {
  return fc_conv_fwd_pairs_tiles(in, W, pair_in, pair_cnt, pair_pos, out, n_in, n_out, K, Cin, Cout, 0, flags, ws, ws_bytes, stream);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// wgrad:  gW[k][ci][co] = sum_o in[nbr[k][o]][ci] * gout[o][co]
// GEMM with M = Cin, N = Cout and the reduction over output rows, split over `S` row ranges whose
// partial products are written to the workspace and summed by k_wgrad_reduce in a fixed order
// (deterministic, no atomics).
// PAIRS: `nbr` / `row_index` are the exact pair lists of fc_kernel_map_pairs (input row, output row; `cnt[k]` valid
// entries per offset) and the reduction runs over the pairs only, split evenly over gridDim.x on the device.
template <int BMc, int BNc, bool HAS_NBR, int BKR, bool PAIRS>
__global__ __launch_bounds__(256, 2) void k_wgrad_mfma(const float* __restrict__ in, const float* __restrict__ gout,
                                                    const int* __restrict__ nbr, const int* __restrict__ row_index,
                                                    const int* __restrict__ cnt,
                                                    float* __restrict__ part, int64_t n_out, int K, int Cin, int Cout,
                                                    int64_t rows_per_split) {
  constexpr int TM = BMc / 64, TN = BNc / 64;
  constexpr int AR = BMc * BKR / 1024, GR = BNc * BKR / 1024;    // float4 loads per thread per stage (BKR rows)
  __shared__ __attribute__((aligned(16))) float As[BKR * BMc];
  __shared__ __attribute__((aligned(16))) float Gs[BKR * BNc];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_n = Cout / BNc, tiles_m = Cin / BMc;
  int y = blockIdx.y;
  const int tn = y % tiles_n; y /= tiles_n;
  const int tm = y % tiles_m; y /= tiles_m;
  const int k = y;
  const int ci0 = tm * BMc, co0 = tn * BNc;
  int64_t total = n_out;
  if (PAIRS) {
    total = cnt[k];
    rows_per_split = ((total + gridDim.x - 1) / gridDim.x + BKR - 1) / BKR * BKR;
  }
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_split;
  int64_t r_end = r_begin + rows_per_split;
  if (r_end > total) r_end = total;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // register-prefetch pipeline: the gathers + gout rows of chunk t+1 are in flight while chunk t is multiplied.
  // (Measured r1: processing wgrad rows in occupancy-mask order with per-chunk skipping LOSES 15-35 % — gout rows stop
  //  being sequential and the skip test costs a barrier per chunk — so wgrad always walks rows in natural order
  //  and row_index must be NULL.)
  f32x4 av[AR], gv[GR];
  auto load_chunk = [&](int64_t rb) {        // branch-free, batched (see k_conv_mfma): indices, gout rows, gathers
    int src[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int lin = tid + 256 * i;
      int rr = lin / (BMc / 4);
      int64_t row = rb + rr;
      int64_t rc = row < r_end ? row : r_end - 1;
      int t = HAS_NBR ? nbr[(int64_t)k * n_out + rc] : (int)rc;      // compile-time: no null test between the loads
      src[i] = row < r_end ? t : -1;
    }
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      int lin = tid + 256 * i;
      int rr = lin / (BNc / 4), c4 = lin % (BNc / 4);
      int64_t row = rb + rr;
      int64_t rc = row < r_end ? row : r_end - 1;
      if (PAIRS) rc = row_index[(int64_t)k * n_out + rc];
      const float* gp = row < r_end ? gout + rc * Cout + co0 + c4 * 4 : g_zero_row + (c4 & 15) * 4;
      gv[i] = *reinterpret_cast<const f32x4*>(gp);
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int lin = tid + 256 * i;
      int c4 = lin % (BMc / 4);
      const float* ap = src[i] < 0 ? g_zero_row + (c4 & 15) * 4 : in + (int64_t)src[i] * Cin + ci0 + c4 * 4;
      av[i] = *reinterpret_cast<const f32x4*>(ap);
    }
  };
  if (r_begin < r_end) load_chunk(r_begin);
  for (int64_t rb = r_begin; rb < r_end; rb += BKR) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int lin = tid + 256 * i;
      int rr = lin / (BMc / 4), c4 = lin % (BMc / 4);
      *reinterpret_cast<f32x4*>(&As[rr * BMc + c4 * 4]) = av[i];
    }
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      int lin = tid + 256 * i;
      int rr = lin / (BNc / 4), c4 = lin % (BNc / 4);
      *reinterpret_cast<f32x4*>(&Gs[rr * BNc + c4 * 4]) = gv[i];
    }
    __syncthreads();
    if (rb + BKR < r_end) load_chunk(rb + BKR);
#pragma unroll
    for (int q = 0; q < BKR / 8; ++q) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][e] = As[(8 * q + 4 * h + e) * BMc + wr * (BMc / 2) + i * 32 + r];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j][e] = Gs[(8 * q + 4 * h + e) * BNc + wc * (BNc / 2) + j * 32 + r];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
  }
  float* dst = part + ((int64_t)blockIdx.x * K + k) * Cin * Cout;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int row = ci0 + wr * (BMc / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        int col = co0 + wc * (BNc / 2) + j * 32 + r;
        dst[(int64_t)row * Cout + col] = acc[i][j][e];
      }
}

// Deeper-pipelined weight gradient (64-channel Cin tile x BNc, 32-row chunks; same partial layout and results as
// k_wgrad_mfma).  r2 ISA reading of the pair-list path of k_wgrad_mfma: every gout row of a chunk went through its own
// `row index load -> s_waitcnt vmcnt(0) -> row load` chain (a bounds branch per load) — five exposed round trips in front of
// 2 048 MFMA cycles.  Here the (input row, output row) indices of chunk t+2 are requested while chunk t+1's rows are
// (branch-free: out-of-range rows read the zero row), and the MFMA fragments of 8-row step q+1 are read from LDS while step
// q multiplies.  3 waves per SIMD.
template <int BNc, bool HAS_NBR, bool PAIRS>
__global__ __launch_bounds__(256, 3) void k_wgrad_mfma_p(const float* __restrict__ in, const float* __restrict__ gout,
                                                          const int* __restrict__ nbr, const int* __restrict__ row_index,
                                                          const int* __restrict__ cnt, float* __restrict__ part,
                                                          int64_t n_out, int K, int Cin, int Cout, int64_t rows_per_split) {
  constexpr int BMc = 64, BKR = 32;
  constexpr int TN = BNc / 64;
  constexpr int AR = 2, GR = BNc * BKR / 1024;   // float4 loads per thread per chunk
  __shared__ __attribute__((aligned(16))) float As[BKR * BMc];
  __shared__ __attribute__((aligned(16))) float Gs[BKR * BNc];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_n = Cout / BNc, tiles_m = Cin / BMc;
  int y = blockIdx.y;
  const int tn = y % tiles_n; y /= tiles_n;
  const int tm = y % tiles_m; y /= tiles_m;
  const int k = y;
  const int ci0 = tm * BMc, co0 = tn * BNc;
  int64_t total = n_out;
  if (PAIRS) {
    total = cnt[k];
    rows_per_split = ((total + gridDim.x - 1) / gridDim.x + BKR - 1) / BKR * BKR;
  }
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_split;
  int64_t r_end = r_begin + rows_per_split;
  if (r_end > total) r_end = total;

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  if (r_begin < r_end) {
    const int* tab_in = HAS_NBR ? nbr + (int64_t)k * n_out : nullptr;
    const int* tab_out = PAIRS ? row_index + (int64_t)k * n_out : nullptr;
    const int a_rr[AR] = {tid / 16, (tid + 256) / 16};
    const int a_c4 = tid % 16;
    int g_rr[GR];
#pragma unroll
    for (int i = 0; i < GR; ++i) g_rr[i] = (tid + 256 * i) / (BNc / 4);
    const int g_c4 = tid % (BNc / 4);
    // indices of a chunk (raw table entries; rows beyond the range are clamped here and zeroed at the row load)
    int ia[AR], ig[GR], ian[AR], ign[GR];
    auto fetch_idx = [&](int64_t rb, int (&va)[AR], int (&vg)[GR]) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int64_t row = rb + a_rr[i];
        if (row >= r_end) row = r_end - 1;
        va[i] = HAS_NBR ? tab_in[row] : (int)row;
      }
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        int64_t row = rb + g_rr[i];
        if (row >= r_end) row = r_end - 1;
        vg[i] = PAIRS ? tab_out[row] : (int)row;
      }
    };
    f32x4 av[AR], gv[GR];
    auto load_rows = [&](int64_t rb) {          // rows of chunk rb from the indices in ia / ig
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        const bool ok = rb + g_rr[i] < r_end;
        const float* gp = ok ? gout + (int64_t)ig[i] * Cout + co0 + g_c4 * 4 : g_zero_row + (g_c4 & 15) * 4;
        gv[i] = *reinterpret_cast<const f32x4*>(gp);
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool ok = rb + a_rr[i] < r_end && ia[i] >= 0;
        const float* ap = ok ? in + (int64_t)ia[i] * Cin + ci0 + a_c4 * 4 : g_zero_row + a_c4 * 4;
        av[i] = *reinterpret_cast<const f32x4*>(ap);
      }
    };
    fetch_idx(r_begin, ia, ig);
    fetch_idx(r_begin + BKR < r_end ? r_begin + BKR : r_begin, ian, ign);
    load_rows(r_begin);
    for (int64_t rb = r_begin; rb < r_end; rb += BKR) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(&As[a_rr[i] * BMc + a_c4 * 4]) = av[i];
#pragma unroll
      for (int i = 0; i < GR; ++i) *reinterpret_cast<f32x4*>(&Gs[g_rr[i] * BNc + g_c4 * 4]) = gv[i];
      __syncthreads();
      // chunk t+1: its indices arrived a chunk ago; request chunk t+2's indices, then t+1's rows (unconditional: past the
      // end the last chunk is re-read and never used)
      const int64_t nb = rb + BKR < r_end ? rb + BKR : rb;
      const int64_t nnb = nb + BKR < r_end ? nb + BKR : nb;
#pragma unroll
      for (int i = 0; i < AR; ++i) ia[i] = ian[i];
#pragma unroll
      for (int i = 0; i < GR; ++i) ig[i] = ign[i];
      fetch_idx(nnb, ian, ign);
      load_rows(nb);
      float fa[2][4], fb[2][TN][4];
      auto read_frag = [&](int q, int u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          fa[u][e] = As[(8 * q + 4 * h + e) * BMc + wr * 32 + r];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[u][j][e] = Gs[(8 * q + 4 * h + e) * BNc + wc * (BNc / 2) + j * 32 + r];
        }
      };
      read_frag(0, 0);
#pragma unroll
      for (int q = 0; q < BKR / 8; ++q) {
        if (q + 1 < BKR / 8) read_frag(q + 1, (q + 1) & 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][e], fb[q & 1][j][e], acc[j], 0, 0, 0);
      }
    }
  }
  float* dst = part + ((int64_t)blockIdx.x * K + k) * Cin * Cout;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = ci0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int col = co0 + wc * (BNc / 2) + j * 32 + r;
      dst[(int64_t)row * Cout + col] = acc[j][e];
    }
}

// Dense-table weight gradient with KO kernel offsets per workgroup sharing ONE gout chunk.  With 64 input channels the
// one-offset kernel above moves 24 KB (8 KB gathered rows + 16 KB of gout) per 0.5 MFLOP chunk = 22 FLOP/B and sits on
// the fabric at ~4.3 TB/s (r2 measurement: 88 TF on the 437k-row level); re-using the staged gout rows for KO = 3 offsets
// lifts that to 39 FLOP/B.  64 x BNc tile of gW[k] per offset, 4 waves as 2 x 2, reduction over rows in natural order.
template <int BNc, int KO>
__global__ __launch_bounds__(256, (BNc == 128) ? 2 : 3) void k_wgrad_multi(const float* __restrict__ in, const float* __restrict__ gout,
                                                         const int* __restrict__ nbr, float* __restrict__ part, int64_t n_out,
                                                         int K, int Cin, int Cout, int64_t rows_per_split) {
  constexpr int TN = BNc / 64;
  constexpr int GR = BNc * 32 / 1024;            // float4 gout loads per thread per 32-row chunk
  __shared__ __attribute__((aligned(16))) float As[KO][32 * 64];
  __shared__ __attribute__((aligned(16))) float Gs[32 * BNc];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_n = Cout / BNc, tiles_m = Cin / 64;
  int y = blockIdx.y;
  const int tn = y % tiles_n; y /= tiles_n;
  const int tm = y % tiles_m; y /= tiles_m;
  const int k0 = y * KO;
  const int ci0 = tm * 64, co0 = tn * BNc;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_split;
  int64_t r_end = r_begin + rows_per_split;
  if (r_end > n_out) r_end = n_out;

  f32x16 acc[KO][TN];
#pragma unroll
  for (int o = 0; o < KO; ++o)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[o][j][e] = 0.f;

  f32x4 av[KO][2], gv[GR];
  auto load_chunk = [&](int64_t rb) {        // branch-free, batched: indices, gout rows, gathers
    int src[KO][2];
#pragma unroll
    for (int o = 0; o < KO; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rr = (tid + 256 * i) / 16;
        const int64_t row = rb + rr;
        const int64_t rc = row < r_end ? row : r_end - 1;
        const int kk = k0 + o < K ? k0 + o : K - 1;
        const int t = nbr[(int64_t)kk * n_out + rc];
        src[o][i] = (row < r_end && k0 + o < K) ? t : -1;
      }
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      const int lin = tid + 256 * i;
      const int rr = lin / (BNc / 4), c4 = lin % (BNc / 4);
      const int64_t row = rb + rr;
      const float* gp = row < r_end ? gout + row * Cout + co0 + c4 * 4 : g_zero_row + (c4 & 15) * 4;
      gv[i] = *reinterpret_cast<const f32x4*>(gp);
    }
#pragma unroll
    for (int o = 0; o < KO; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = (tid + 256 * i) % 16;
        const float* ap = src[o][i] < 0 ? g_zero_row + c4 * 4 : in + (int64_t)src[o][i] * Cin + ci0 + c4 * 4;
        av[o][i] = *reinterpret_cast<const f32x4*>(ap);
      }
  };
  if (r_begin < r_end) load_chunk(r_begin);
  const bool prio = g_fc_prio == 0;               // see g_fc_prio (mode 1: forward / backward-data kernels only)
  for (int64_t rb = r_begin; rb < r_end; rb += 32) {
    __syncthreads();
#pragma unroll
    for (int o = 0; o < KO; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int lin = tid + 256 * i;
        *reinterpret_cast<f32x4*>(&As[o][(lin / 16) * 64 + (lin % 16) * 4]) = av[o][i];
      }
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      const int lin = tid + 256 * i;
      *reinterpret_cast<f32x4*>(&Gs[(lin / (BNc / 4)) * BNc + (lin % (BNc / 4)) * 4]) = gv[i];
    }
    __syncthreads();
    load_chunk(rb + 32 < r_end ? rb + 32 : rb);            // unconditional (see k_conv_mfma)
    if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float b[TN][4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j][e] = Gs[(8 * q + 4 * h + e) * BNc + wc * (BNc / 2) + j * 32 + r];
#pragma unroll
      for (int o = 0; o < KO; ++o) {
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = As[o][(8 * q + 4 * h + e) * 64 + wr * 32 + r];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[o][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[j][e], acc[o][j], 0, 0, 0);
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
  }
#pragma unroll
  for (int o = 0; o < KO; ++o) {
    if (k0 + o >= K) break;
    float* dst = part + ((int64_t)blockIdx.x * K + k0 + o) * Cin * Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = ci0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = co0 + wc * (BNc / 2) + j * 32 + r;
        dst[(int64_t)row * Cout + col] = acc[o][j][e];
      }
  }
}

// generic wgrad: block = (row range, k); each thread owns (ci,co) pairs strided by blockDim.
__global__ void k_wgrad_fma(const float* __restrict__ in, const float* __restrict__ gout, const int* __restrict__ nbr,
                            float* __restrict__ part, int64_t n_out, int K, int Cin, int Cout, int64_t rows_per_split) {
  const int k = blockIdx.y;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_split;
  int64_t r_end = r_begin + rows_per_split;
  if (r_end > n_out) r_end = n_out;
  float* dst = part + ((int64_t)blockIdx.x * K + k) * Cin * Cout;
  for (int p = threadIdx.x; p < Cin * Cout; p += blockDim.x) {
    int ci = p / Cout, co = p % Cout;
    float acc = 0.f;
    for (int64_t o = r_begin; o < r_end; ++o) {
      int i = nbr ? nbr[(int64_t)k * n_out + o] : (int)o;
      if (i >= 0) acc = fmaf(in[(int64_t)i * Cin + ci], gout[o * Cout + co], acc);
    }
    dst[p] = acc;
  }
}

__global__ void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ gW, int64_t elems, int S) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += part[(int64_t)s * elems + i];
  gW[i] = acc;
}

// The same for elems % 4 == 0 with 16-byte accesses and G slab groups per output (256 threads = 256 / G float4 columns x G
// groups): group g sums slabs [g * ceil(S / G), ...) in slab order with four loads in flight, the G group sums are added in
// group order through LDS — a fixed order, whatever the grid (r3: the stem's 192 x 20 KB slabs went through 5 184 threads
// with one dependent 4-byte load each).  Measured and dropped (r3, profiles/r3_notes.md): combining the partial tiles INSIDE
// the weight-gradient launch (last-arriving workgroup per tile, agent-scope release / ticket / acquire) — the release's
// buffer_wbl2 in each of the ~1 500 workgroups of a launch writes back everything the concurrently running main-stream
// kernels have dirtied in that L2: 32.4 -> 34.3 ms per step.
template <int G>
__global__ __launch_bounds__(256) void k_wgrad_reduce4(const float* __restrict__ part, float* __restrict__ gW, int64_t elems4, int S) {
  __shared__ f32x4 red[256];
  constexpr int COLS = 256 / G;
  const int col = threadIdx.x % COLS, g = threadIdx.x / COLS;
  const int64_t i = (int64_t)blockIdx.x * COLS + col;
  const int per = (S + G - 1) / G;
  const int s0 = g * per, s1 = (s0 + per < S) ? s0 + per : S;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < elems4) {
    const f32x4* src = reinterpret_cast<const f32x4*>(part) + i;
    int sl = s0;
    for (; sl + 4 <= s1; sl += 4) {
      const f32x4 v0 = src[(int64_t)sl * elems4], v1 = src[(int64_t)(sl + 1) * elems4];
      const f32x4 v2 = src[(int64_t)(sl + 2) * elems4], v3 = src[(int64_t)(sl + 3) * elems4];
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; sl < s1; ++sl) a += src[(int64_t)sl * elems4];
  }
  if (G == 1) {
    if (i < elems4) reinterpret_cast<f32x4*>(gW)[i] = a;
    return;
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (g == 0 && i < elems4) {
    for (int q = 1; q < G; ++q) a += red[q * COLS + col];
    reinterpret_cast<f32x4*>(gW)[i] = a;
  }
}

// (K,Cin,Cout) -> (K,Cout,Cin)
__global__ void k_transpose_w(const float* __restrict__ W, float* __restrict__ Wt, int K, int Cin, int Cout) {
  __shared__ float tile[32][33];
  int k = blockIdx.z;
  int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int ci = ci0 + j, co = co0 + tx;
    tile[j][tx] = (ci < Cin && co < Cout) ? W[((int64_t)k * Cin + ci) * Cout + co] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int co = co0 + j, ci = ci0 + tx;
    if (ci < Cin && co < Cout) Wt[((int64_t)k * Cout + co) * Cin + ci] = tile[tx][j];
  }
}

extern "C" {

static void wgrad_tiles(int Cin, int Cout, int flags, int* bm, int* bn) {
  *bm = 64;                                  // measured: 64-channel Cin tiles beat 128 on every benchmark layer
  *bn = (Cout % 128 == 0) ? 128 : 64;
  int fbm = (flags >> 4) & 3, fbn = (flags >> 6) & 3;      // tuning overrides
  if (fbm == 1) *bm = 64;
  if (fbm == 2 && Cin % 128 == 0) *bm = 128;
  if (fbn == 1) *bn = 64;
  if (fbn == 2 && Cout % 128 == 0) *bn = 128;
}

// several offsets per workgroup (k_wgrad_multi): every dense table with >= 4096 rows (r2 nbench, one-offset kernel -> multi:
// 55k rows 128->128 504 -> 446 us, 6.9k rows 256->256 278 -> 247, 256->128 148 -> 135, 441k rows 128->64 2121 -> 2003, 64->64
// 1086 -> 1051); flags bit29 disables it, bit30 restricts it to its first rule (Cin = 64, >= 32768 rows)
#define WGRAD_KO 3
// split-bf16 weight gradients: rows loaded 16 B per lane and transposed by ds_read_b64_tr_b16 (k_wgrad_x6t, r4) or the register
// transposition of r3 (k_wgrad_x6; FC_WGRAD_TR=0) — bit-identical results
// (r5: the r3 register-transposing kernel k_wgrad_x6 and its FC_WGRAD_TR switch are gone; FC_WGRAD_TR64=0 keeps 64 x 64-channel
// pair lists and the table-free dense GEMMs on the fp32 MFMA kernel)
static inline bool wgrad_tr64() {
  static const bool on = !(getenv("FC_WGRAD_TR64") && atoi(getenv("FC_WGRAD_TR64")) == 0);
  return on;
}
static inline bool wgrad_multi_ok(int64_t n_out, int K, int Cin, int Cout, int flags, bool dense_table) {
  return dense_table && !(flags & 1) && !(flags & (1 << 29)) && K % WGRAD_KO == 0 &&
         Cin % 64 == 0 && Cout % 64 == 0 && n_out >= 4096 && (!(flags & (1 << 30)) || (Cin == 64 && n_out >= 32768));
}

static void wgrad_plan(int64_t n_out, int K, int Cin, int Cout, int flags, bool dense_table, int* S, int64_t* rows_per_split) {
  if (!(flags & 1) && Cin == STEM_CIN && Cout == STEM_COUT && K <= 27) {     // stem: 4096 rows per block
    int64_t m = n_out > 0 ? n_out : 1;
    *rows_per_split = 1024;                      // (r2: 512 rows per block is slower — more partial tiles to write and reduce)
    *S = (int)fc_cdiv(m, 1024);
    return;
  }
  bool mfma_ok = !(flags & 1) && (Cin % 64 == 0) && (Cout % 64 == 0);
  int tbm, tbn;
  wgrad_tiles(Cin, Cout, flags, &tbm, &tbn);
  if ((flags & (1 << 24)) && !dense_table && Cin % 128 == 0) tbm = 128;      // split-bf16 pair-list kernel: 128-channel tiles
  int64_t tiles = mfma_ok ? (int64_t)K * (Cin / tbm) * (Cout / tbn) : (int64_t)K;
  // aim for ~1728 workgroups (r2 sweep: 2048 rounded UP left a nearly empty last round on most layers — 128->128 on 55k
  // rows 559 us at 38 splits, 448 at 32), at least 512 rows per split, at most 256 splits
  int64_t s = 1728 / tiles;                  // (same-box A/B in the full step: neutral, 230.3 vs 230.9 scenes/s; kept: fewer partial tiles)
  if (wgrad_multi_ok(n_out, K, Cin, Cout, flags, dense_table)) {
    // uniform long workgroups: exactly one resident round (3 per CU), fewer partial gradients to write and re-read
    const bool wide = Cout % 128 == 0;
    tiles = (int64_t)(K / WGRAD_KO) * (Cin / 64) * (Cout / (wide ? 128 : 64));
    // r4 A/B in the full step (weight gradients beside the dependent chain): 3/4 of a resident round 22.91 ms, a full round
    // (r3) 23.34, half 23.70, a quarter 28.63, two rounds 23.25 — the main stream's kernels find a slot sooner.  With the
    // transposing-read kernel (k_wgrad_x6t, 1.4x faster per launch) half a round of the wide variant is ahead: 256 / 512
    // 22.26-22.35 ms, 192 / 512 22.41, 384 / 512 22.65, 128 / 512 23.36 (same box)
    static const int round_wide = getenv("FC_WGRAD_ROUND_WIDE") ? atoi(getenv("FC_WGRAD_ROUND_WIDE")) : 256;
    static const int round_narrow = getenv("FC_WGRAD_ROUND_NARROW") ? atoi(getenv("FC_WGRAD_ROUND_NARROW")) : 512;
    s = (wide ? round_wide : round_narrow) / tiles;     // the 128-column variant holds 2 workgroups per CU (registers)
  }
  int64_t max_by_rows = fc_cdiv(n_out > 0 ? n_out : 1, mfma_ok ? 512 : 2048);
  if (s > max_by_rows) s = max_by_rows;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  if ((flags >> 8) & 255) s = (flags >> 8) & 255;         // tuning override
  int64_t rps = fc_align(fc_cdiv(n_out > 0 ? n_out : 1, s), 64);
  s = fc_cdiv(n_out > 0 ? n_out : 1, rps);
  *S = (int)s;
  *rows_per_split = rps;
}

// gW = sum of the S partial slabs, fixed order
static int wgrad_reduce(const float* part, float* gW, int64_t elems, int S, hipStream_t stream) {
  if (elems % 4 == 0) {
    const int64_t e4 = elems / 4;
    if (S >= 64) k_wgrad_reduce4<16><<<(unsigned)fc_cdiv(e4, 16), 256, 0, stream>>>(part, gW, e4, S);
    else if (S >= 16) k_wgrad_reduce4<4><<<(unsigned)fc_cdiv(e4, 64), 256, 0, stream>>>(part, gW, e4, S);
    else k_wgrad_reduce4<1><<<(unsigned)fc_cdiv(e4, 256), 256, 0, stream>>>(part, gW, e4, S);
  } else {
    k_wgrad_reduce<<<(unsigned)fc_cdiv(elems, 256), 256, 0, stream>>>(part, gW, elems, S);
  }
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ---- stem convolution with saved gathered inputs (training): forward writes col (n_out, 84), the weight gradient streams it ----
int fc_stem_conv_fwd(const float* in, const float* W, const int* nbr, float* out, float* col, int64_t n_in, int64_t n_out,
                     int K, hipStream_t stream) {
  if (n_in < 0 || n_out < 0 || K < 1 || K > 27 || !nbr) return FC_EINVAL;
  if (n_out == 0) return FC_OK;
  size_t smem = (size_t)(STEM_ROWS * STEM_FWD_LDA + 96 * 64) * sizeof(float);
  k_stem_fwd<<<(unsigned)fc_cdiv(n_out, STEM_ROWS), 256, smem, stream>>>(in, W, nbr, out, col, n_out, K);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

// ~192 row ranges (multiples of 64 rows): every range ends with a 24 KB cross-wave LDS sum, a 20 KB partial and a pass of
// k_wgrad_reduce over all partials, so FEWER, longer ranges win until the chip runs dry (r2, 580k rows, forward + weight
// gradient: 1 024-row ranges 436 us, 2 048: 365, 3 072: 353, 4 096: 378, 8 192: 512; the gather kernel: 493)
static void stem_col_plan(int64_t n_out, int* S, int64_t* rps) {
  int64_t r = fc_cdiv(fc_cdiv(n_out > 0 ? n_out : 1, 192), STEM_ROWS) * STEM_ROWS;
  *rps = r;
  *S = (int)fc_cdiv(n_out > 0 ? n_out : 1, r);
}

int64_t fc_stem_conv_wgrad_ws_bytes(int64_t n_out, int K) {
  int S; int64_t rps;
  stem_col_plan(n_out, &S, &rps);
  return (int64_t)S * K * STEM_CIN * STEM_COUT * (int64_t)sizeof(float);
}

int fc_stem_conv_wgrad(const float* col, const float* gout, float* gW, int64_t n_out, int K, void* ws, int64_t ws_bytes,
                       hipStream_t stream) {
  if (n_out < 0 || K < 1 || K > 27) return FC_EINVAL;
  const int64_t elems = (int64_t)K * STEM_CIN * STEM_COUT;
  if (n_out == 0) {
    FC_HIP(hipMemsetAsync(gW, 0, elems * sizeof(float), stream));
    return FC_OK;
  }
  int S; int64_t rps;
  stem_col_plan(n_out, &S, &rps);
  if (ws_bytes < (int64_t)S * elems * (int64_t)sizeof(float)) return FC_EWS;
  float* part = (S == 1) ? gW : (float*)ws;
  size_t smem = (size_t)(STEM_ROWS * STEM_JP + STEM_ROWS * 64) * sizeof(float);
  k_stem_wgrad_col<<<(unsigned)S, 256, smem, stream>>>(col, gout, part, n_out, K, rps);
  FC_CHECK_LAUNCH();
  if (S > 1) return wgrad_reduce(part, gW, elems, S, stream);
  return FC_OK;
}

int64_t fc_conv_wgrad_ws_bytes(int64_t n_out, int K, int Cin, int Cout, int flags) {
  int S, S2; int64_t rps;
  wgrad_plan(n_out, K, Cin, Cout, flags, false, &S, &rps);
  wgrad_plan(n_out, K, Cin, Cout, flags, true, &S2, &rps);        // dense tables may take the multi-offset kernel
  if (S2 > S) S = S2;
  return (int64_t)S * K * Cin * Cout * (int64_t)sizeof(float);
}

static int conv_wgrad_impl(const float* in, const float* gout, const int* nbr, const int* row_index, const int* cnt,
                           float* gW, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int flags, void* ws,
                           int64_t ws_bytes, hipStream_t stream) {
  AmaxHintScope hint_scope;
  if (n_in < 0 || n_out < 0 || K < 1 || Cin < 1 || Cout < 1) return FC_EINVAL;
  const int64_t elems = (int64_t)K * Cin * Cout;
  if (n_out == 0) {
    FC_HIP(hipMemsetAsync(gW, 0, elems * sizeof(float), stream));
    return FC_OK;
  }
  if (!nbr && (K != 1 || n_in != n_out)) return FC_EINVAL;
  int S; int64_t rps;
  const bool dense_table = nbr && !cnt;
  wgrad_plan(n_out, K, Cin, Cout, flags, dense_table, &S, &rps);
  if (ws_bytes < (int64_t)S * elems * (int64_t)sizeof(float)) return FC_EWS;
  float* part = (S == 1) ? gW : (float*)ws;
  bool mfma_ok = !(flags & 1) && (Cin % 64 == 0) && (Cout % 64 == 0);
  if (cnt && !mfma_ok) return FC_EINVAL;       // pair lists are an MFMA-path feature
  // h3 (conv_x6.h): the k_wgrad_x6t launches below split both operands into two fp16 pieces, scaled by their amax words
  const bool x6t = mfma_ok && (flags & (1 << 24)) &&
                   (wgrad_multi_ok(n_out, K, Cin, Cout, flags, dense_table) ||
                    (cnt && (Cin % 128 == 0 || Cout % 128 == 0 || wgrad_tr64())) || (!nbr && !cnt && wgrad_tr64()));
  const bool h3 = x6t && split_mode() == 2;
  // buffer addressing of both operands (wgrad_x6.h): below 2 GB each, row indices below 2^24; flags bit27: flat addresses (A/B, tests)
  const int wbuf = (!(flags & (1 << 27)) && (uint64_t)n_in * (uint64_t)Cin * 4u < (1ull << 31) - 4096u && (uint64_t)n_out * (uint64_t)Cout * 4u < (1ull << 31) - 4096u &&
                    n_in < (1 << 24) && n_out < (1 << 24) && (uint64_t)K * (uint64_t)n_out * 4u < (1ull << 31) - 4096u) ? 1 : 0;
  const unsigned *am_a = nullptr, *am_g = nullptr;
  if (h3) {
    int rc = operand_amax(in, n_in * (int64_t)Cin, 0, stream, &am_a);
    if (rc == FC_OK) rc = operand_amax(gout, n_out * (int64_t)Cout, 1, stream, &am_g);
    if (rc != FC_OK) return rc;
  }
  if (!(flags & 1) && nbr && Cin == STEM_CIN && Cout == STEM_COUT && K <= 27) {
    size_t smem = (size_t)(STEM_ROWS * STEM_JP + STEM_ROWS * 64) * sizeof(float);
    k_stem_wgrad<<<(unsigned)S, 256, smem, stream>>>(in, gout, nbr, part, n_out, K, rps);
  } else if (mfma_ok && wgrad_multi_ok(n_out, K, Cin, Cout, flags, dense_table)) {
    const int bn = (Cout % 128 == 0) ? 128 : 64;
    dim3 grid((unsigned)S, (unsigned)((K / WGRAD_KO) * (Cin / 64) * (Cout / bn)));
    if (flags & (1 << 24)) {                     // split-bf16 (wgrad_x6.h)
      if (g_bf16_fast && bn == 128) k_wgrad_x6t<64, 128, WGRAD_KO, false, 1><<<grid, 256, 0, stream>>>(in, gout, nbr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, nullptr, nullptr, wbuf);
      else if (g_bf16_fast) k_wgrad_x6t<64, 64, WGRAD_KO, false, 1><<<grid, 256, 0, stream>>>(in, gout, nbr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, nullptr, nullptr, wbuf);
      else if (h3 && bn == 128) k_wgrad_x6t<64, 128, WGRAD_KO, false, 2><<<grid, 256, 0, stream>>>(in, gout, nbr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, am_a, am_g, wbuf);
      else if (h3) k_wgrad_x6t<64, 64, WGRAD_KO, false, 2><<<grid, 256, 0, stream>>>(in, gout, nbr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, am_a, am_g, wbuf);
      else if (bn == 128) k_wgrad_x6t<64, 128, WGRAD_KO, false><<<grid, 256, 0, stream>>>(in, gout, nbr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, nullptr, nullptr, wbuf);
      else k_wgrad_x6t<64, 64, WGRAD_KO, false><<<grid, 256, 0, stream>>>(in, gout, nbr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, nullptr, nullptr, wbuf);
    } else
    if (bn == 128) k_wgrad_multi<128, WGRAD_KO><<<grid, 256, 0, stream>>>(in, gout, nbr, part, n_out, K, Cin, Cout, rps);
    else k_wgrad_multi<64, WGRAD_KO><<<grid, 256, 0, stream>>>(in, gout, nbr, part, n_out, K, Cin, Cout, rps);
  } else if (mfma_ok && cnt && (flags & (1 << 24)) && (Cin % 128 == 0 || Cout % 128 == 0 || wgrad_tr64())) {
    // split-bf16 over the pair lists (r3 nbench: 128 x 128 tiles 119 -> 95 us on 15k rows 128->128, 111 -> 89 / 109 -> 87 on
    // the 256- and 512-channel levels; 64 x 64 tiles — one accumulator per wave, a dependent MFMA chain — lost to the fp32
    // kernel with the r3 kernel and win with k_wgrad_x6t: wgrad_tr64()).  128-channel Cin tiles only while they still fill the
    // chip (862 rows, 512->128: 216 workgroups of 128 x 128 tiles 49 us, fp32 36 us)
    const int bn = (Cout % 128 == 0) ? 128 : 64;
    int bm = (Cin % 128 == 0) ? 128 : 64;
    if (bm == 128 && bn == 128 && (int64_t)S * K * (Cin / 128) * (Cout / 128) < 512) bm = 64;
    dim3 grid((unsigned)S, (unsigned)(K * (Cin / bm) * (Cout / bn)));
#define FC_WX6(BM_, BN_)                                                                                                             \
  do {                                                                                                                               \
    if (g_bf16_fast) k_wgrad_x6t<BM_, BN_, 1, true, 1><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps, nullptr, nullptr, wbuf); \
    else if (h3) k_wgrad_x6t<BM_, BN_, 1, true, 2><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps, am_a, am_g, wbuf); \
    else k_wgrad_x6t<BM_, BN_, 1, true><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps, nullptr, nullptr, wbuf); \
  } while (0)
    if (bm == 128 && bn == 128) FC_WX6(128, 128);
    else if (bm == 128) FC_WX6(128, 64);
    else if (bn == 128) FC_WX6(64, 128);
    else FC_WX6(64, 64);
#undef FC_WX6
  } else if (mfma_ok && !nbr && !cnt && (flags & (1 << 24)) && wgrad_tr64()) {
    // table-free dense GEMM gW = in^T gout over the rows (K = 1): the same kernel with the row itself as the index
    const int bn = (Cout % 128 == 0) ? 128 : 64;
    int bm = (Cin % 128 == 0) ? 128 : 64;
    if (bm == 128 && bn == 128 && (int64_t)S * (Cin / 128) * (Cout / 128) < 512) bm = 64;
    dim3 grid((unsigned)S, (unsigned)(K * (Cin / bm) * (Cout / bn)));
#define FC_WX6D(BM_, BN_)                                                                                                              \
  do {                                                                                                                                \
    if (g_bf16_fast) k_wgrad_x6t<BM_, BN_, 1, false, 1><<<grid, 256, 0, stream>>>(in, gout, nullptr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, nullptr, nullptr, wbuf); \
    else if (h3) k_wgrad_x6t<BM_, BN_, 1, false, 2><<<grid, 256, 0, stream>>>(in, gout, nullptr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, am_a, am_g, wbuf); \
    else k_wgrad_x6t<BM_, BN_, 1, false><<<grid, 256, 0, stream>>>(in, gout, nullptr, nullptr, nullptr, part, n_out, K, Cin, Cout, rps, nullptr, nullptr, wbuf);   \
  } while (0)
    if (bm == 128 && bn == 128) FC_WX6D(128, 128);
    else if (bm == 128) FC_WX6D(128, 64);
    else if (bn == 128) FC_WX6D(64, 128);
    else FC_WX6D(64, 64);
#undef FC_WX6D
  } else if (mfma_ok) {
    int bm, bn;
    wgrad_tiles(Cin, Cout, flags, &bm, &bn);
    if (cnt) bm = 64;
    dim3 grid((unsigned)S, (unsigned)(K * (Cin / bm) * (Cout / bn)));
    const bool deep = (flags & (1 << 19)) && bm == 64 && nbr;          // 64-row chunks (tuning flag)
    // k_wgrad_mfma_p where it measured ahead (r2 nbench, same box: pair lists with 128-wide gout tiles +3..7 %; 64-wide
    // tiles -5 %, dense tables -7..13 %).  bit16: never, bit20: wherever it applies (tests / A-B).
    const bool wpipe = bm == 64 && !(flags & (1 << 19)) && !(flags & (1 << 16)) && ((flags & (1 << 20)) || (cnt && bn == 128));
    if (wpipe && cnt) {
      if (bn == 128) k_wgrad_mfma_p<128, true, true><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
      else k_wgrad_mfma_p<64, true, true><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
    } else if (wpipe && nbr) {
      if (bn == 128) k_wgrad_mfma_p<128, true, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
      else k_wgrad_mfma_p<64, true, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
    } else if (wpipe) {
      if (bn == 128) k_wgrad_mfma_p<128, false, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
      else k_wgrad_mfma_p<64, false, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
    } else
    if (cnt) {
      if (bn == 128) k_wgrad_mfma<64, 128, true, 32, true><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
      else k_wgrad_mfma<64, 64, true, 32, true><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
    } else
    if (deep) {
      if (bn == 128) k_wgrad_mfma<64, 128, true, 64, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
      else k_wgrad_mfma<64, 64, true, 64, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps);
    } else
    if (bm == 128 && bn == 128) { if (nbr) k_wgrad_mfma<128, 128, true, 32, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps); else k_wgrad_mfma<128, 128, false, 32, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps); }
    else if (bm == 128) { if (nbr) k_wgrad_mfma<128, 64, true, 32, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps); else k_wgrad_mfma<128, 64, false, 32, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps); }
    else if (bn == 128) { if (nbr) k_wgrad_mfma<64, 128, true, 32, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps); else k_wgrad_mfma<64, 128, false, 32, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps); }
    else { if (nbr) k_wgrad_mfma<64, 64, true, 32, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps); else k_wgrad_mfma<64, 64, false, 32, false><<<grid, 256, 0, stream>>>(in, gout, nbr, row_index, cnt, part, n_out, K, Cin, Cout, rps); }
  } else {
    dim3 grid((unsigned)S, (unsigned)K);
    k_wgrad_fma<<<grid, 256, 0, stream>>>(in, gout, nbr, part, n_out, K, Cin, Cout, rps);
  }
  FC_CHECK_LAUNCH();
  if (S > 1) return wgrad_reduce(part, gW, elems, S, stream);
  return FC_OK;
}

int fc_conv_wgrad(const float* in, const float* gout, const int* nbr, const int* row_index, float* gW, int64_t n_in,
                  int64_t n_out, int K, int Cin, int Cout, int flags, void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (row_index) return FC_EINVAL;             // reserved (see k_wgrad_mfma)
  return conv_wgrad_impl(in, gout, nbr, nullptr, nullptr, gW, n_in, n_out, K, Cin, Cout, flags, ws, ws_bytes, stream);
}

int fc_conv_wgrad_pairs(const float* in, const float* gout, const int* pair_in, const int* pair_out, const int* pair_cnt,
                        float* gW, int64_t n_in, int64_t n_out, int K, int Cin, int Cout, int flags, void* ws,
                        int64_t ws_bytes, hipStream_t stream) {
  if (!pair_in || !pair_out || !pair_cnt) return FC_EINVAL;
  return conv_wgrad_impl(in, gout, pair_in, pair_out, pair_cnt, gW, n_in, n_out, K, Cin, Cout, flags & ~(1 << 19), ws,
                         ws_bytes, stream);
}

int fc_transpose_weight(const float* W, float* Wt, int K, int Cin, int Cout, hipStream_t stream) {
  if (K < 1 || Cin < 1 || Cout < 1 || K > 65535) return FC_EINVAL;
  dim3 grid((unsigned)fc_cdiv(Cout, 32), (unsigned)fc_cdiv(Cin, 32), (unsigned)K);
  k_transpose_w<<<grid, 256, 0, stream>>>(W, Wt, K, Cin, Cout);
  FC_CHECK_LAUNCH();
  return FC_OK;
}

}  // extern "C"
