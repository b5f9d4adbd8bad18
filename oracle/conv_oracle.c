/* ORACLE — test infrastructure only (built by oracle/Makefile into libfcaf3d_oracle.so, called via ctypes from
 * tests/ and bench.py's cpu_baseline leg; never linked into the product).
 *
 * C / OpenMP restatement of how MinkowskiEngine v0.5.4's CPU backend (the reference's un-vendored sparse-conv
 * dependency, pin: docker/Dockerfile:27-32) runs a sparse convolution — SURVEY.md Appendix A.3: for every kernel offset k
 * with its (input row, output row) pair list, gather the input rows -> dense GEMM with W[k] -> scatter-add into the
 * output rows; backward-data walks the same pairs with W[k]^T, backward-weights accumulates in[i]^T (x) gout[o].
 * Call sites in the reference: me_resnet.py:19-21, :56-62 (BasicBlock convs), fcaf3d_neck_with_head.py:52, :69.
 * Checked against oracle/me_oracle.py::conv (and its autograd) in tests/test_oracle_dense.py.
 *
 * r3: the block GEMMs are register-blocked SIMD micro-kernels (4 rows x 2 AVX2 vectors, or 4 rows x 4 AVX-512 vectors
 * where the CPU has them — chosen at run time), i.e. what a BLAS SGEMM does for these skinny shapes, instead of r2's
 * auto-vectorised triple loops (0.3 % of the host's peak: a straw man).  Parallelism: within one offset the
 * output rows (forward) / input rows (backward-data) of the pairs are distinct, so blocks of pairs go to OpenMP threads
 * without atomics; backward-weights splits the pairs into chunks x the input channels into 16-row tiles, every task
 * owns a private partial tile, and the partials are summed in chunk order (deterministic). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLK 96

typedef float v8f __attribute__((vector_size(32), aligned(4)));
typedef float v16f __attribute__((vector_size(64), aligned(4)));

int oc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static int has_avx512(void) {
  static int v = -1;
  if (v < 0) v = __builtin_cpu_supports("avx512f") ? 1 : 0;
  return v;
}
int oc_simd_width(void) { return has_avx512() ? 16 : 8; }

/* C[nb][Co] = A[nb][Ci] @ M[Ci][Co]   (Co % (NV * VW) may leave a tail handled 1 vector at a time; Co % VW == 0) */
#define DEF_GEMM(NAME, VT, VW, NV, TGT)                                                                             \
  TGT static void NAME(const float* A, int nb, const float* M, int Ci, int Co, float* C) {                           \
    for (int r0 = 0; r0 < nb; r0 += 4) {                                                                            \
      const int mr = nb - r0 < 4 ? nb - r0 : 4;                                                                     \
      const float* a0 = A + (size_t)r0 * Ci;                                                                        \
      const float* a1 = A + (size_t)(r0 + (mr > 1 ? 1 : 0)) * Ci;                                                   \
      const float* a2 = A + (size_t)(r0 + (mr > 2 ? 2 : 0)) * Ci;                                                   \
      const float* a3 = A + (size_t)(r0 + (mr > 3 ? 3 : 0)) * Ci;                                                   \
      int j0 = 0;                                                                                                   \
      for (; j0 + NV * VW <= Co; j0 += NV * VW) {                                                                   \
        VT acc[4][NV];                                                                                              \
        for (int r = 0; r < 4; ++r)                                                                                 \
          for (int q = 0; q < NV; ++q) acc[r][q] = (VT){0};                                                         \
        for (int ci = 0; ci < Ci; ++ci) {                                                                           \
          const float* m = M + (size_t)ci * Co + j0;                                                                \
          VT b[NV];                                                                                                 \
          for (int q = 0; q < NV; ++q) b[q] = *(const VT*)(m + q * VW);                                             \
          const float s0 = a0[ci], s1 = a1[ci], s2 = a2[ci], s3 = a3[ci];                                           \
          for (int q = 0; q < NV; ++q) {                                                                            \
            acc[0][q] += s0 * b[q];                                                                                 \
            acc[1][q] += s1 * b[q];                                                                                 \
            acc[2][q] += s2 * b[q];                                                                                 \
            acc[3][q] += s3 * b[q];                                                                                 \
          }                                                                                                         \
        }                                                                                                           \
        for (int r = 0; r < mr; ++r)                                                                                \
          for (int q = 0; q < NV; ++q) *(VT*)(C + (size_t)(r0 + r) * Co + j0 + q * VW) = acc[r][q];                 \
      }                                                                                                             \
      for (; j0 + VW <= Co; j0 += VW) {                                                                             \
        VT acc[4];                                                                                                  \
        for (int r = 0; r < 4; ++r) acc[r] = (VT){0};                                                               \
        for (int ci = 0; ci < Ci; ++ci) {                                                                           \
          const VT b = *(const VT*)(M + (size_t)ci * Co + j0);                                                      \
          acc[0] += a0[ci] * b; acc[1] += a1[ci] * b; acc[2] += a2[ci] * b; acc[3] += a3[ci] * b;                    \
        }                                                                                                           \
        for (int r = 0; r < mr; ++r) *(VT*)(C + (size_t)(r0 + r) * Co + j0) = acc[r];                               \
      }                                                                                                             \
      for (; j0 < Co; ++j0)                                                                                         \
        for (int r = 0; r < mr; ++r) {                                                                              \
          float s = 0.f;                                                                                            \
          const float* a = A + (size_t)(r0 + r) * Ci;                                                               \
          for (int ci = 0; ci < Ci; ++ci) s += a[ci] * M[(size_t)ci * Co + j0];                                     \
          C[(size_t)(r0 + r) * Co + j0] = s;                                                                        \
        }                                                                                                           \
    }                                                                                                               \
  }

DEF_GEMM(gemm_avx2, v8f, 8, 2, __attribute__((target("avx2,fma"))))
DEF_GEMM(gemm_avx512, v16f, 16, 4, __attribute__((target("avx512f"))))

static void block_gemm(const float* A, int nb, const float* M, int Ci, int Co, float* C) {
  if (has_avx512()) gemm_avx512(A, nb, M, Ci, Co, C);
  else gemm_avx2(A, nb, M, Ci, Co, C);
}

/* T[ci0..ci0+tr][Co] += sum_p A[pa[p]][ci0 + r] * G[pg[p]][:]   over p in [p0, p1) — the weight-gradient micro-kernel:
 * 4 input channels x NV vectors of output channels held in registers across the pair loop */
#define DEF_WGRAD(NAME, VT, VW, NV, TGT)                                                                            \
  TGT static void NAME(const float* in, const int* pa, const float* gout, const int* pg, int64_t p0, int64_t p1,    \
                       int Ci, int Co, int ci0, int tr, float* T) {                                                 \
    for (int r0 = 0; r0 < tr; r0 += 4) {                                                                            \
      const int mr = tr - r0 < 4 ? tr - r0 : 4;                                                                     \
      const int c0 = ci0 + r0, c1 = ci0 + r0 + (mr > 1), c2 = ci0 + r0 + (mr > 2 ? 2 : 0), c3 = ci0 + r0 + (mr > 3 ? 3 : 0); \
      int j0 = 0;                                                                                                   \
      for (; j0 + NV * VW <= Co; j0 += NV * VW) {                                                                   \
        VT acc[4][NV];                                                                                              \
        for (int r = 0; r < 4; ++r)                                                                                 \
          for (int q = 0; q < NV; ++q) acc[r][q] = (VT){0};                                                         \
        for (int64_t p = p0; p < p1; ++p) {                                                                         \
          const float* a = in + (size_t)pa[p] * Ci;                                                                 \
          const float* g = gout + (size_t)pg[p] * Co + j0;                                                          \
          VT b[NV];                                                                                                 \
          for (int q = 0; q < NV; ++q) b[q] = *(const VT*)(g + q * VW);                                             \
          const float s0 = a[c0], s1 = a[c1], s2 = a[c2], s3 = a[c3];                                               \
          for (int q = 0; q < NV; ++q) {                                                                            \
            acc[0][q] += s0 * b[q];                                                                                 \
            acc[1][q] += s1 * b[q];                                                                                 \
            acc[2][q] += s2 * b[q];                                                                                 \
            acc[3][q] += s3 * b[q];                                                                                 \
          }                                                                                                         \
        }                                                                                                           \
        for (int r = 0; r < mr; ++r)                                                                                \
          for (int q = 0; q < NV; ++q) *(VT*)(T + (size_t)(r0 + r) * Co + j0 + q * VW) = acc[r][q];                 \
      }                                                                                                             \
      for (; j0 < Co; ++j0)                                                                                         \
        for (int r = 0; r < mr; ++r) {                                                                              \
          float s = 0.f;                                                                                            \
          for (int64_t p = p0; p < p1; ++p) s += in[(size_t)pa[p] * Ci + ci0 + r0 + r] * gout[(size_t)pg[p] * Co + j0]; \
          T[(size_t)(r0 + r) * Co + j0] = s;                                                                        \
        }                                                                                                           \
    }                                                                                                               \
  }

DEF_WGRAD(wgrad_avx2, v8f, 8, 2, __attribute__((target("avx2,fma"))))
DEF_WGRAD(wgrad_avx512, v16f, 16, 4, __attribute__((target("avx512f"))))

/* pair list of offset k from the dense table: returns count; pin/pout sized n_out */
static int64_t pairs_of(const int* nbr_k, int64_t n_out, int* pin, int* pout) {
  int64_t c = 0;
  for (int64_t o = 0; o < n_out; ++o)
    if (nbr_k[o] >= 0) { pin[c] = nbr_k[o]; pout[c] = (int)o; ++c; }
  return c;
}

/* dst[rows[r]] += src[gather[r]] @ M   for r in [0, cnt): src (.,Ci), M (Ci,Co), dst (.,Co) */
static void gather_gemm_scatter(const float* src, const int* gather, const int* rows, int64_t cnt, const float* M, int Ci,
                                int Co, float* dst) {
#pragma omp parallel
  {
    float* A = (float*)aligned_alloc(64, sizeof(float) * BLK * (size_t)Ci);
    float* C = (float*)aligned_alloc(64, sizeof(float) * BLK * (size_t)Co);
#pragma omp for schedule(dynamic, 2)
    for (int64_t b0 = 0; b0 < cnt; b0 += BLK) {
      const int nb = (int)(cnt - b0 < BLK ? cnt - b0 : BLK);
      for (int r = 0; r < nb; ++r) memcpy(A + (size_t)r * Ci, src + (size_t)gather[b0 + r] * Ci, sizeof(float) * Ci);
      block_gemm(A, nb, M, Ci, Co, C);
      for (int r = 0; r < nb; ++r) {
        float* d = dst + (size_t)rows[b0 + r] * Co;
        const float* c = C + (size_t)r * Co;
#pragma omp simd
        for (int j = 0; j < Co; ++j) d[j] += c[j];
      }
    }
    free(A);
    free(C);
  }
}

/* out (n_out,Cout) = sum_k in[nbr[k][o]] @ W[k]      (out is overwritten) */
void oc_conv_fwd(const float* in, const float* W, const int* nbr, int64_t n_out, int K, int Cin, int Cout, float* out) {
  int* pin = (int*)malloc(sizeof(int) * (n_out > 0 ? n_out : 1));
  int* pout = (int*)malloc(sizeof(int) * (n_out > 0 ? n_out : 1));
  memset(out, 0, sizeof(float) * (size_t)n_out * Cout);
  for (int k = 0; k < K; ++k) {
    const int64_t cnt = pairs_of(nbr + (size_t)k * n_out, n_out, pin, pout);
    if (cnt) gather_gemm_scatter(in, pin, pout, cnt, W + (size_t)k * Cin * Cout, Cin, Cout, out);
  }
  free(pin);
  free(pout);
}

/* gin (n_in,Cin) = sum_k scatter_i( gout[o] @ W[k]^T )      (gin is overwritten) */
void oc_conv_dgrad(const float* gout, const float* W, const int* nbr, int64_t n_in, int64_t n_out, int K, int Cin, int Cout,
                   float* gin) {
  int* pin = (int*)malloc(sizeof(int) * (n_out > 0 ? n_out : 1));
  int* pout = (int*)malloc(sizeof(int) * (n_out > 0 ? n_out : 1));
  float* Wt = (float*)malloc(sizeof(float) * (size_t)Cin * Cout);
  memset(gin, 0, sizeof(float) * (size_t)n_in * Cin);
  for (int k = 0; k < K; ++k) {
    const int64_t cnt = pairs_of(nbr + (size_t)k * n_out, n_out, pin, pout);
    if (!cnt) continue;
    const float* Wk = W + (size_t)k * Cin * Cout;
    for (int ci = 0; ci < Cin; ++ci)
      for (int co = 0; co < Cout; ++co) Wt[(size_t)co * Cin + ci] = Wk[(size_t)ci * Cout + co];
    gather_gemm_scatter(gout, pout, pin, cnt, Wt, Cout, Cin, gin);
  }
  free(pin);
  free(pout);
  free(Wt);
}

/* gW (K,Cin,Cout): gW[k] = sum over the pairs of k of in[i]^T (x) gout[o]      (gW is overwritten) */
void oc_conv_wgrad(const float* in, const float* gout, const int* nbr, int64_t n_out, int K, int Cin, int Cout, float* gW) {
  int* pin = (int*)malloc(sizeof(int) * (n_out > 0 ? n_out : 1));
  int* pout = (int*)malloc(sizeof(int) * (n_out > 0 ? n_out : 1));
  memset(gW, 0, sizeof(float) * (size_t)K * Cin * Cout);
  const int TR = 16;                                  /* input channels per task tile */
  const int ntiles = (Cin + TR - 1) / TR;
  const int nthreads = oc_num_threads();
  const int avx512 = has_avx512();
  for (int k = 0; k < K; ++k) {
    const int64_t cnt = pairs_of(nbr + (size_t)k * n_out, n_out, pin, pout);
    if (!cnt) continue;
    /* enough (chunk, tile) tasks for every core; chunks of at least 256 pairs */
    int64_t nchunks = (4 * (int64_t)nthreads + ntiles - 1) / ntiles;
    if (nchunks > (cnt + 255) / 256) nchunks = (cnt + 255) / 256;
    if (nchunks < 1) nchunks = 1;
    const int64_t per = (cnt + nchunks - 1) / nchunks;
    float* part = (float*)aligned_alloc(64, sizeof(float) * (size_t)nchunks * Cin * Cout);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int64_t c = 0; c < nchunks; ++c)
      for (int t = 0; t < ntiles; ++t) {
        const int64_t p0 = c * per, p1 = p0 + per < cnt ? p0 + per : cnt;
        const int ci0 = t * TR, tr = Cin - ci0 < TR ? Cin - ci0 : TR;
        float* T = part + ((size_t)c * Cin + ci0) * Cout;
        if (avx512) wgrad_avx512(in, pin, gout, pout, p0, p1, Cin, Cout, ci0, tr, T);
        else wgrad_avx2(in, pin, gout, pout, p0, p1, Cin, Cout, ci0, tr, T);
      }
    float* g = gW + (size_t)k * Cin * Cout;
    const int64_t elems = (int64_t)Cin * Cout;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < elems; ++e) {
      float s = 0.f;
      for (int64_t c = 0; c < nchunks; ++c) s += part[(size_t)c * elems + e];      /* chunk order: deterministic */
      g[e] = s;
    }
    free(part);
  }
  free(pin);
  free(pout);
}
