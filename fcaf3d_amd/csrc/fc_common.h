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
// buffers): read it past the CU's vector L1.  r3 (csrc/assign.hip, tools/trace_det.py): with another stream's kernels keeping
// the CUs busy, a consumer wave that lands on a CU late can still hit that CU's line from the previous use of the address —
// the L2 is coherent, the L1 is not.  `nt` / agent-scope loads are served by the L2.
typedef float fc_f4v __attribute__((ext_vector_type(4)));
__device__ static inline float fc_ld(const float* p) { return __builtin_nontemporal_load(p); }
__device__ static inline int fc_ld(const int* p) { return __builtin_nontemporal_load(p); }
__device__ static inline double fc_ld(const double* p) { return __builtin_nontemporal_load(p); }
__device__ static inline float4 fc_ld4(const float* p) {
  const fc_f4v v = __builtin_nontemporal_load(reinterpret_cast<const fc_f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
