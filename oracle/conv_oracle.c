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
 * Parallelism: within one offset the output rows (forward) / input rows (backward-data) of the pairs are distinct, so
 * blocks of pairs go to OpenMP threads without atomics; backward-weights gives every thread a slice of the input
 * channels.  The block GEMMs are plain loops written for the auto-vectoriser (-O3 -march=x86-64-v3: AVX2 + FMA, so the prebuilt library runs on any current server CPU). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLK 64

int oc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

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
    float* A = (float*)malloc(sizeof(float) * BLK * Ci);
    float* C = (float*)malloc(sizeof(float) * BLK * Co);
#pragma omp for schedule(dynamic, 4)
    for (int64_t b0 = 0; b0 < cnt; b0 += BLK) {
      const int nb = (int)(cnt - b0 < BLK ? cnt - b0 : BLK);
      for (int r = 0; r < nb; ++r) memcpy(A + (size_t)r * Ci, src + (size_t)gather[b0 + r] * Ci, sizeof(float) * Ci);
      memset(C, 0, sizeof(float) * nb * Co);
      for (int r = 0; r < nb; ++r) {
        float* c = C + (size_t)r * Co;
        const float* a = A + (size_t)r * Ci;
        for (int ci = 0; ci < Ci; ++ci) {
          const float av = a[ci];
          const float* m = M + (size_t)ci * Co;
          for (int j = 0; j < Co; ++j) c[j] += av * m[j];
        }
      }
      for (int r = 0; r < nb; ++r) {
        float* d = dst + (size_t)rows[b0 + r] * Co;
        const float* c = C + (size_t)r * Co;
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
  const int SL = 8;                                   /* input channels per task */
  for (int k = 0; k < K; ++k) {
    const int64_t cnt = pairs_of(nbr + (size_t)k * n_out, n_out, pin, pout);
    if (!cnt) continue;
    float* g = gW + (size_t)k * Cin * Cout;
#pragma omp parallel for schedule(dynamic, 1)
    for (int c0 = 0; c0 < Cin; c0 += SL) {
      const int c1 = c0 + SL < Cin ? c0 + SL : Cin;
      for (int64_t p = 0; p < cnt; ++p) {
        const float* a = in + (size_t)pin[p] * Cin;
        const float* go = gout + (size_t)pout[p] * Cout;
        for (int ci = c0; ci < c1; ++ci) {
          const float av = a[ci];
          float* row = g + (size_t)ci * Cout;
          for (int j = 0; j < Cout; ++j) row[j] += av * go[j];
        }
      }
    }
  }
  free(pin);
  free(pout);
}
