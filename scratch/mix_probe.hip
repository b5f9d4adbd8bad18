// probe: the h3 split by v_fma_mixlo/hi_f16 + v_fma_mix_f32 against the cvt-based split (scratch; not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned rn2h(float x0, float x1) { const f32x2 v = {x0, x1}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2)); }
__global__ void k(const float* x, float s, unsigned* ref, unsigned* got, int n2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  float x0 = x[2 * i], x1 = x[2 * i + 1];
  { float a = x0 * s, b = x1 * s; unsigned p0 = rn2h(a, b); f16x2 h = __builtin_bit_cast(f16x2, p0);
    float r0 = a - (float)h[0], r1 = b - (float)h[1]; ref[2 * i] = p0; ref[2 * i + 1] = rn2h(r0, r1); }
  unsigned h = 0;
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x0), "v"(s));
  asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  float r0, r1;
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(s), "v"(h));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(s), "v"(h));
  got[2 * i] = h; got[2 * i + 1] = rn2h(r0, r1);
}
int main() {
  const int n = 1 << 22;
  std::mt19937 rng(5); std::normal_distribution<float> N(0.f, 1.f);
  std::vector<float> x(n);
  for (auto& v : x) v = N(rng) * std::exp(3.f * N(rng));
  x[0] = 0.f; x[1] = -0.f; x[2] = 1.f; x[3] = 65504.f / 16384.f; x[4] = 1e-9f; x[5] = -1e-9f; x[6] = 3.999f; x[7] = 1.0009765625f;
  float amax = 0; for (float v : x) amax = std::max(amax, std::fabs(v));
  int e; std::frexp(amax, &e);
  float *dx; unsigned *dr, *dg; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dr, n * 4)); CK(hipMalloc(&dg, n * 4));
  CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
  for (int shift = 0; shift < 3; ++shift) {
    const float s = std::ldexp(1.f, 15 - e - 8 * shift);      // the product scale, and two looser ones (low pieces go subnormal)
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, s, dr, dg, n / 2);
    std::vector<unsigned> r(n), g(n); CK(hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g.data(), dg, n * 4, hipMemcpyDeviceToHost));
    long bad = 0; for (int i = 0; i < n; ++i) if (r[i] != g[i]) { if (bad < 5) printf("  mismatch at %d: ref %08x got %08x (x %g %g)\n", i, r[i], g[i], x[(i / 2) * 2], x[(i / 2) * 2 + 1]); ++bad; }
    printf("scale 2^%d: %ld of %d dwords differ\n", 15 - e - 8 * shift, bad, n);
  }
  return 0;
}
