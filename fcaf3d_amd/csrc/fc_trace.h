// Optional per-wave timeline of the convolution kernels (built only with -DFC_TRACE: tools/nbench_trace; never in the
// product library).  record = 8 x u64: [block x|y|z|wave, xcc|units, t0..t5] with wall_clock64() ticks (100 MHz).
#pragma once
#ifdef FC_TRACE
#ifdef FC_TRACE_DEFINE
__device__ unsigned long long* g_trace_buf;
__device__ int g_trace_cap;
extern "C" int fc_debug_trace(unsigned long long* buf, int cap) {
  FC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &buf, sizeof(buf)));
  FC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cap), &cap, sizeof(cap)));
  return FC_OK;
}
#define TR_BUF g_trace_buf
#define TR_CAP g_trace_cap
#endif
#define TR_DECL unsigned long long tr_t[6] = {0, 0, 0, 0, 0, 0}
#define TR(i) tr_t[i] = wall_clock64()
#define TR_FLUSH(units)                                                                              \
  do {                                                                                               \
    if ((threadIdx.x & 63) == 0 && TR_BUF) {                                                         \
      const int slot__ = (int)(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)); \
      if (slot__ < TR_CAP) {                                                                         \
        unsigned long long* o__ = TR_BUF + (size_t)slot__ * 8;                                       \
        unsigned int xcc__;                                                                          \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__));                         \
        o__[0] = ((unsigned long long)blockIdx.x << 32) | (blockIdx.y << 20) | (blockIdx.z << 8) | (threadIdx.x >> 6); \
        o__[1] = ((unsigned long long)(xcc__ & 15) << 32) | (unsigned int)(units);                   \
        for (int q__ = 0; q__ < 6; ++q__) o__[2 + q__] = tr_t[q__];                                  \
      }                                                                                              \
    }                                                                                                \
  } while (0)
#else
#define TR_DECL
#define TR(i)
#define TR_FLUSH(units)
#endif
