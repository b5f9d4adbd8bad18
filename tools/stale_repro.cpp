// Stand-alone reproduction attempt of the r3 hand-off finding (profiles/r3_notes.md, "A stale-L1 hazard"): a small table at a
// FIXED address, rewritten by a producer kernel and read by a consumer kernel of the NEXT launch on the same stream, over and
// over, while a second stream keeps the CUs busy.  Stream order makes the consumer's reads happen-after the producer's writes;
// a consumer wave that still sees the previous round's values has hit a line that survived the kernel boundary in its CU's
// vector L1.
//
//   hipcc -O2 --offload-arch=gfx950 tools/stale_repro.cpp -o tools/stale_repro
//   tools/stale_repro [--rounds 4000] [--table-bytes 4096] [--blocks 1024] [--busy alu|lds|mem|none] [--busy-ms 1.0]
//
// For every load flavour it prints the number of (round, word) mismatches:
//   plain   — ordinary global loads (served by the vector L1)
//   agent   — __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)  (global_load ... sc1: served by the L2)
//   nt      — __builtin_nontemporal_load                                         (the r3 form)
// The consumer is L1-WARM by construction (every round reads the same addresses from every CU) and checks every word.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                                      \
  do {                                                                                                             \
    hipError_t e__ = (x);                                                                                          \
    if (e__ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); exit(2); } \
  } while (0)

__global__ void k_produce(int* __restrict__ T, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) T[i] = v + i;
}

template <int MODE>
__device__ __forceinline__ int ld(const int* p) {
  if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 2) return __builtin_nontemporal_load(p);
  return *p;
}

// every block reads the WHOLE table (the pattern of the assignment / normalisation finalisers), and re-reads it `passes` times
template <int MODE>
__global__ void k_consume(const int* __restrict__ T, int n, int v, int passes, unsigned long long* __restrict__ bad,
                          int* __restrict__ first_bad) {
  int local = 0;
  for (int p = 0; p < passes; ++p)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int got = ld<MODE>(T + i);
      if (got != v + i) {
        ++local;
        if (first_bad[0] == 0) { first_bad[0] = 1; first_bad[1] = got; first_bad[2] = v + i; first_bad[3] = blockIdx.x; }
      }
    }
  if (local) atomicAdd(bad, (unsigned long long)local);
}

// the other stream's load: long-running workgroups that hold every CU (3 per CU), ALU / LDS / memory flavoured
__global__ __launch_bounds__(256) void k_busy(float* __restrict__ buf, long long iters, int flavour, long long n) {
  __shared__ float lds[12 * 1024];               // 48 KB: three workgroups per CU
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = 0; i < iters; ++i) {
    if (flavour == 0) {
#pragma unroll
      for (int k = 0; k < 64; ++k) a = a * b + 1e-7f;
    } else if (flavour == 1) {
      lds[(threadIdx.x + i * 33) % (12 * 1024)] = a;
      __syncthreads();
      a += lds[(threadIdx.x * 7 + i) % (12 * 1024)];
    } else {
      const long long j = (t * 4 + i * 1048576) % n;
      a += buf[j];
      buf[(j + 13) % n] = a;
    }
  }
  if (a == 123.456f) buf[0] = a;
}

int main(int argc, char** argv) {
  int rounds = 4000, table_bytes = 4096, blocks = 1024, passes = 1;
  std::string busy = "alu";
  double busy_ms = 1.0;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--rounds") rounds = atoi(argv[++i]);
    else if (a == "--table-bytes") table_bytes = atoi(argv[++i]);
    else if (a == "--blocks") blocks = atoi(argv[++i]);
    else if (a == "--passes") passes = atoi(argv[++i]);
    else if (a == "--busy") busy = argv[++i];
    else if (a == "--busy-ms") busy_ms = atof(argv[++i]);
  }
  const int n = table_bytes / 4;
  int* T; unsigned long long* bad; int* first_bad; float* buf;
  const long long nbuf = 64ll << 20;
  CK(hipMalloc(&T, table_bytes)); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first_bad, 16)); CK(hipMalloc(&buf, nbuf * 4));
  CK(hipMemset(buf, 0, nbuf * 4));
  hipStream_t sa, sb;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo));      // the hand-off stream (like the coordinate stream: default priority)
  CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));      // the busy stream (like bench.py's priority main stream)
  const int flavour = busy == "alu" ? 0 : busy == "lds" ? 1 : 2;
  // calibrate the busy kernel to ~busy_ms per launch
  long long iters = 2000;
  if (busy != "none") {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, sb));
      k_busy<<<768, 256, 0, sb>>>(buf, iters, flavour, nbuf);
      CK(hipEventRecord(e1, sb));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      iters = (long long)(iters * busy_ms / (ms > 1e-3 ? ms : 1e-3)) + 1;
    }
  }
  printf("table %d B, consumer %d blocks x 256 threads, %d rounds, busy stream: %s (%lld iterations per launch)\n", table_bytes, blocks, rounds,
         busy.c_str(), iters);
  const char* names[3] = {"plain", "agent", "nt"};
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(first_bad, 0, 16));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r) {
      if (busy != "none" && (r % 4) == 0) k_busy<<<768, 256, 0, sb>>>(buf, iters, flavour, nbuf);
      const int v = (r + 1) * 100003 + mode;
      k_produce<<<(n + 255) / 256, 256, 0, sa>>>(T, n, v);
      if (mode == 0) k_consume<0><<<blocks, 256, 0, sa>>>(T, n, v, passes, bad, first_bad);
      else if (mode == 1) k_consume<1><<<blocks, 256, 0, sa>>>(T, n, v, passes, bad, first_bad);
      else k_consume<2><<<blocks, 256, 0, sa>>>(T, n, v, passes, bad, first_bad);
      if ((r & 63) == 63) CK(hipStreamSynchronize(sa));        // keep the queues short: the busy stream stays a few launches ahead
    }
    CK(hipDeviceSynchronize());
    unsigned long long hbad; int fb[4];
    CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(fb, first_bad, 16, hipMemcpyDeviceToHost));
    printf("  %-6s loads: %llu stale words of %llu", names[mode], hbad, (unsigned long long)rounds * blocks * n * passes);
    if (hbad) printf("   (first: got %d, expected %d, block %d)", fb[1], fb[2], fb[3]);
    printf("\n");
  }
  return 0;
}
