// Shared helpers for the gfx950 kernels of libfcaf3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FC_OK 0
#define FC_EINVAL (-1)
#define FC_EWS (-2)   // workspace too small

#define FC_CHECK_LAUNCH()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

#define FC_HIP(x)                              \
  do {                                         \
    hipError_t e__ = (x);                      \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

static inline int64_t fc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t fc_align(int64_t a, int64_t b) { return fc_cdiv(a, b) * b; }

#define FC_EMPTY_KEY ((unsigned long long)0xFFFFFFFFFFFFFFFFull)

// (b,x,y,z) -> 64-bit key, lexicographic; 16 bits per spatial axis with a 2^15 bias, 16 bits of batch index.
// Valid voxel coordinates are therefore |c| <= FC_COORD_LIMIT (the margin covers the largest kernel offset of the
// pyramid, 64 voxels, twice) — +-326 m at 1 cm voxels; fc_hash_unique reports anything else through its count (-1)
// instead of letting the key alias a neighbouring axis / scene (MinkowskiEngine keeps full int32 coordinates).
#define FC_COORD_LIMIT 32639
__host__ __device__ static inline unsigned long long fc_pack(int b, int x, int y, int z) {
  return ((unsigned long long)(unsigned)b << 48) | ((unsigned long long)(unsigned)(x + 32768) << 32) |
         ((unsigned long long)(unsigned)(y + 32768) << 16) | (unsigned long long)(unsigned)(z + 32768);
}

__device__ static inline unsigned long long fc_mix(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

// probe an open-addressing table; returns value or -1
__device__ static inline int fc_lookup(const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                       unsigned long long mask, unsigned long long key) {
  unsigned long long h = fc_mix(key) & mask;
  while (true) {
    unsigned long long k = keys[h];
    if (k == key) return vals[h];
    if (k == FC_EMPTY_KEY) return -1;
    h = (h + 1) & mask;
  }
}

__device__ static inline int fc_floor_div(int a, int b) {
  int q = a / b, r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

// Loads for data one launch hands to the next through a small table at a FIXED address (per-stream workspace, persistent
// buffers): agent-scope relaxed atomic loads — `global_load ... sc1`, served by the L2, never by the CU's vector L1
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").
// r3 (csrc/assign.hip, tools/trace_det.py): with the target assignment on the coordinate stream while the main stream ran the
// split-bf16 convolutions, ~30 % of the assignment calls returned a few dozen rows computed from the PREVIOUS call's kth / best
// entries although stream order puts the consumer kernel after its producer.  r4: a stand-alone producer -> fixed-address
// table -> consumer loop beside a busy second stream (tools/stale_repro.cpp: 13 configurations, 10^9-10^10 words each) never
// returns a stale word with plain loads — the kernel boundary does invalidate the L1 there — so the mechanism behind the r3
// symptom is NOT established; what is established is the symptom with the real kernels and its absence with L2-served loads
// (tests/test_gpu_model.py::test_target_assignment_is_stable_beside_concurrent_convolutions; -DFC_LD_PLAIN builds the plain-load
// library the A/B of profiles/r4_notes.md was run with).  The loads are therefore defined by SCOPE (r3 used the cache-policy
// builtin __builtin_nontemporal_load, which happens to lower to the same instruction) and used for every cross-launch table a
// kernel re-reads at a fixed address: assignment tables, optimizer clip coefficient, fused-loss partials, the normalisation
// kernels' statistics / partial tables, the executor's small-gradient sums.
typedef float fc_f4v __attribute__((ext_vector_type(4)));
#ifdef FC_LD_PLAIN
__device__ static inline float fc_ld(const float* p) { return *p; }
__device__ static inline int fc_ld(const int* p) { return *p; }
__device__ static inline double fc_ld(const double* p) { return *p; }
__device__ static inline float4 fc_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#else
__device__ static inline float fc_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline int fc_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline double fc_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline float4 fc_ld4(const float* p) {          // two 8-byte agent-scope loads (the widest atomic access)
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                     __uint_as_float((unsigned)(b >> 32)));
}
#endif

// ---- r6: amax slots (csrc/conv_x6.h "h3": max |x| of a convolution operand, the scale of its two-piece fp16 split) ----------------
// A slot is FC_AMAX_SUB sub-words, one per 64-byte line (FC_AMAX_SLOT_BYTES = 2 KB); the operand's amax is the MAXIMUM of the
// sub-words.  Producers that fold their output into a slot (norm.hip amax_commit) pick a sub-word by block index, so that the ~10^5
// waves of an elementwise launch over a 437k-row tensor do not all poll one L2 line (one word: +30 us per launch, r6_notes.md);
// fc_amax publishes into sub-word 0 and uses words 1, 2 of the first line as scratch.  Slots start zeroed.
#define FC_AMAX_SUB 32
#define FC_AMAX_STRIDE 16                                      // dwords between sub-words
#ifndef FC_AMAX_SLOT_BYTES
#define FC_AMAX_SLOT_BYTES (FC_AMAX_SUB * FC_AMAX_STRIDE * 4)      // (= include/fcaf3d_hip.h)
#endif
static_assert(FC_AMAX_SLOT_BYTES == FC_AMAX_SUB * FC_AMAX_STRIDE * 4, "amax slot layout");
// every lane of a wave calls this; returns the slot's amax (bit pattern), wave-uniform
__device__ static inline unsigned fc_amax_read(const unsigned* __restrict__ slot) {
  const int lane = threadIdx.x & 63;
  unsigned m = lane < FC_AMAX_SUB ? __hip_atomic_load(slot + lane * FC_AMAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}

// producers' side of a slot (norm.hip, head.hip): fold what a thread stores, then ONE atomicMax per block on the block's sub-word.
// EVERY thread of the block must reach amax_commit (it holds a __syncthreads).
__device__ __forceinline__ void amax_fold(unsigned& m, float v) {
  const unsigned u = __float_as_uint(v) & 0x7fffffffu;
  m = (u > m && u < 0x7f800000u) ? u : m;
}
// (one atomic per BLOCK: same-line atomics serialise at ~35 ns each in the L2 — with one per wave a 7 us launch of 1 840 blocks carried
// 230 of them per sub-word = +8 us)
__device__ __forceinline__ void amax_commit(unsigned m, unsigned* __restrict__ dst) {
  __shared__ unsigned s_am[16];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0) s_am[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (int)((blockDim.x + 63) >> 6);
    for (int w = 1; w < nw; ++w) m = s_am[w] > m ? s_am[w] : m;
    unsigned* wd = dst + ((blockIdx.x + blockIdx.y) & (FC_AMAX_SUB - 1)) * FC_AMAX_STRIDE;        // this block's sub-word (fc_common.h)
    if (m > __hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(wd, m);
  }
}
extern thread_local unsigned* t_fc_amax_out;           // set by fc_amax_out_hint (norm.hip), consumed by the next producer entry point of the thread
static inline unsigned* take_amax_out() { unsigned* p = t_fc_amax_out; t_fc_amax_out = nullptr; return p; }
