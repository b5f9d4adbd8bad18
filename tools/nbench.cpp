// Native micro-benchmark of the sparse-convolution entry points of libfcaf3d_hip.so on benchmark-shaped kernel maps
// (no Python, no torch: a fresh GPU box spends its minutes on kernels, not on `import torch`).
//
//   hipcc -O2 --offload-arch=gfx950 tools/nbench.cpp -Iinclude -Lfcaf3d_amd -lfcaf3d_hip -Wl,-rpath,'$ORIGIN/../fcaf3d_amd' -o tools/nbench
//   tools/nbench [--batch 8] [--only L3] [--mode fwd|wgrad|all] [--reps 10] [--s-sweep] [--no-check] [--x6] [--prio N]
//   (--x6: the default fp32 routes beside the split-bf16 kernels only; every forward run also prints its rms / max error against an
//    fp64 evaluation of 48 sampled output rows; --prio: wave-priority mode of the kernels, conv.hip g_fc_prio)
//
// Scenes follow fcaf3d_amd/synthetic.py (room 6 x 5 x 2.7 m, floor + walls + 15 cuboids, 100 000 points, 5 mm noise,
// 2 cm voxels); coordinate sets and kernel maps are built on the HOST with ME's rules (first-occurrence row order,
// floor-strided sets, generative 2x2x2 children) — only the convolution kernels under test run on the GPU.
// Every case is checked against the library's generic FMA kernel (flags bit0) before it is timed.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "fcaf3d_hip.h"

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e__ = (x);                                                                       \
    if (e__ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); exit(2); } \
  } while (0)
#define FC(x)                                                                       \
  do {                                                                              \
    int rc__ = (x);                                                                 \
    if (rc__ != 0) { fprintf(stderr, "%s -> %d at %s:%d\n", #x, rc__, __FILE__, __LINE__); exit(3); } \
  } while (0)

struct V4 { int b, x, y, z; };
static inline uint64_t pack(const V4& c) {
  return ((uint64_t)(uint32_t)c.b << 48) | ((uint64_t)(uint32_t)(c.x + 32768) << 32) | ((uint64_t)(uint32_t)(c.y + 32768) << 16) |
         (uint64_t)(uint32_t)(c.z + 32768);
}
static inline uint64_t mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }

struct Hash {                      // open addressing, key -> row
  std::vector<uint64_t> keys; std::vector<int> vals; uint64_t mask;
  void init(size_t n) { size_t cap = 2; while (cap < 2 * n + 2) cap *= 2; keys.assign(cap, ~0ull); vals.assign(cap, -1); mask = cap - 1; }
  int find(uint64_t k) const { uint64_t h = mix(k) & mask; while (true) { if (keys[h] == k) return vals[h]; if (keys[h] == ~0ull) return -1; h = (h + 1) & mask; } }
  bool insert(uint64_t k, int v) { uint64_t h = mix(k) & mask; while (true) { if (keys[h] == k) return false; if (keys[h] == ~0ull) { keys[h] = k; vals[h] = v; return true; } h = (h + 1) & mask; } }
};

struct CSet {                      // coordinate set in first-occurrence order + hash
  std::vector<V4> c; Hash h; int stride;
  int n() const { return (int)c.size(); }
};
static int fdiv(int a, int b) { int q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; }

static CSet unique_of(const std::vector<V4>& in, int q, int stride) {
  CSet s; s.stride = stride; s.h.init(in.size());
  for (const V4& v : in) {
    V4 w{v.b, fdiv(v.x, q) * q, fdiv(v.y, q) * q, fdiv(v.z, q) * q};
    if (s.h.insert(pack(w), (int)s.c.size())) s.c.push_back(w);
  }
  return s;
}
static CSet generate(const CSet& p) {   // children 8i+k, x fastest
  CSet s; s.stride = p.stride / 2; s.h.init(p.c.size() * 8);
  const int hs = s.stride;
  for (const V4& v : p.c)
    for (int k = 0; k < 8; ++k) {
      V4 w{v.b, v.x + (k & 1) * hs, v.y + ((k >> 1) & 1) * hs, v.z + ((k >> 2) & 1) * hs};
      s.h.insert(pack(w), (int)s.c.size()); s.c.push_back(w);
    }
  return s;
}
// nbr[k][o] = row in `in` of out.c[o] + offset_k * in.stride (k3: centred, x fastest)
static std::vector<int> kernel_map(const CSet& in, const CSet& out, int ks) {
  const int K = ks * ks * ks, n = out.n();
  std::vector<int> nbr((size_t)K * n);
  int k = 0;
  for (int dz = 0; dz < ks; ++dz) for (int dy = 0; dy < ks; ++dy) for (int dx = 0; dx < ks; ++dx, ++k) {
    const int ox = (dx - ks / 2) * in.stride, oy = (dy - ks / 2) * in.stride, oz = (dz - ks / 2) * in.stride;
    for (int o = 0; o < n; ++o) {
      const V4& v = out.c[o];
      nbr[(size_t)k * n + o] = in.h.find(pack(V4{v.b, v.x + ox, v.y + oy, v.z + oz}));
    }
  }
  return nbr;
}

static std::vector<V4> make_points(int batch, int npts, float voxel) {
  std::vector<V4> pts; pts.reserve((size_t)batch * npts);
  for (int b = 0; b < batch; ++b) {
    std::mt19937_64 rng(1000 + b);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 0.005);
    struct R { double o[3], u[3], v[3], area; };
    std::vector<R> rects;
    auto add = [&](double ox, double oy, double oz, double ux, double uy, double uz, double vx, double vy, double vz) {
      R r{{ox, oy, oz}, {ux, uy, uz}, {vx, vy, vz}, 0};
      double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
      r.area = std::sqrt(cx * cx + cy * cy + cz * cz); rects.push_back(r);
    };
    const double X = 6.0, Y = 5.0, Z = 2.7;
    add(0, 0, 0, X, 0, 0, 0, Y, 0); add(0, 0, 0, X, 0, 0, 0, 0, Z); add(0, Y, 0, X, 0, 0, 0, 0, Z);
    add(0, 0, 0, 0, Y, 0, 0, 0, Z); add(X, 0, 0, 0, Y, 0, 0, 0, Z);
    for (int i = 0; i < 15; ++i) {
      double w = 0.4 + 1.4 * U(rng), l = 0.4 + 0.8 * U(rng), hh = 0.4 + 1.1 * U(rng);
      double cx = w / 2 + (X - w) * U(rng), cy = l / 2 + (Y - l) * U(rng);
      double bx = cx - w / 2, by = cy - l / 2;
      add(bx, by, hh, w, 0, 0, 0, l, 0); add(bx, by, 0, w, 0, 0, 0, 0, hh); add(bx, by + l, 0, w, 0, 0, 0, 0, hh);
      add(bx, by, 0, 0, l, 0, 0, 0, hh); add(bx + w, by, 0, 0, l, 0, 0, 0, hh);
    }
    std::vector<double> cum; double tot = 0; for (auto& r : rects) { tot += r.area; cum.push_back(tot); }
    for (int i = 0; i < npts; ++i) {
      double t = U(rng) * tot; size_t w = std::lower_bound(cum.begin(), cum.end(), t) - cum.begin(); if (w >= rects.size()) w = rects.size() - 1;
      const R& r = rects[w]; double a = U(rng), bb = U(rng);
      double p[3]; for (int d = 0; d < 3; ++d) p[d] = r.o[d] + a * r.u[d] + bb * r.v[d] + N(rng);
      pts.push_back(V4{b, (int)std::floor((float)p[0] / voxel), (int)std::floor((float)p[1] / voxel), (int)std::floor((float)p[2] / voxel)});
    }
  }
  return pts;
}

template <class T> struct Dev {
  T* p = nullptr; size_t n = 0;
  void alloc(size_t m) { if (m > n) { if (p) CK(hipFree(p)); CK(hipMalloc(&p, std::max<size_t>(m, 1) * sizeof(T))); n = m; } }
  void up(const std::vector<T>& v) { alloc(v.size()); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> down(size_t m) { std::vector<T> v(m); CK(hipMemcpy(v.data(), p, m * sizeof(T), hipMemcpyDeviceToHost)); return v; }
};

struct Case { std::string name; const CSet* in; const CSet* out; int ks, Cin, Cout; bool dense; };

static double time_us(int reps, const std::function<void()>& fn) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  fn(); fn(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) fn();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms * 1e3 / reps;
}

static double max_rel_err(const std::vector<float>& a, const std::vector<float>& ref) {
  double mx = 0, md = 0;
  for (size_t i = 0; i < a.size(); ++i) { mx = std::max(mx, (double)std::fabs(ref[i])); md = std::max(md, (double)std::fabs(a[i] - ref[i])); }
  return md / (mx > 0 ? mx : 1);
}

extern "C" int fc_debug_set_prio(int mode);
int main(int argc, char** argv) {
  int prio = 0;
  int batch = 8, reps = 10, npts = 100000; std::string only, mode = "all", trace_file; bool check = true;
  int trace_variant = 0, trace_tbl = 0; bool popc_sort = false, s_sweep = false, x6_only = false; bool n64 = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--batch") batch = atoi(argv[++i]);
    else if (a == "--reps") reps = atoi(argv[++i]);
    else if (a == "--points") npts = atoi(argv[++i]);
    else if (a == "--only") only = argv[++i];
    else if (a == "--mode") mode = argv[++i];
    else if (a == "--no-check") check = false;
    else if (a == "--prio") prio = atoi(argv[++i]);
    else if (a == "--popc-sort") popc_sort = true;
    else if (a == "--s-sweep") s_sweep = true;
    else if (a == "--n64") n64 = true;
    else if (a == "--x6") x6_only = true;                      // default fp32 routes + the split-bf16 kernel only
    else if (a == "--trace") trace_file = argv[++i];            // needs the FC_TRACE build (tools/nbench_trace)
    else if (a == "--trace-variant") trace_variant = atoi(argv[++i]);
    else if (a == "--trace-tbl") trace_tbl = atoi(argv[++i]);
  }
  // ---- coordinate pyramid ------------------------------------------------------------------------------
  std::vector<V4> pts = make_points(batch, npts, 0.02f);
  CSet s1 = unique_of(pts, 1, 1);
  CSet s2 = unique_of(s1.c, 2, 2), s4 = unique_of(s2.c, 4, 4);
  CSet l1 = unique_of(s4.c, 8, 8), l2 = unique_of(l1.c, 16, 16), l3 = unique_of(l2.c, 32, 32), l4 = unique_of(l3.c, 64, 64);
  CSet n2 = generate(l4), n1 = generate(n2), n0 = generate(n1);       // backbone level is a subset of the generated set (floor rule)
  printf("# batch %d: N0 %d | conv1 %d | pool %d | L1 %d L2 %d L3 %d L4 %d | neck %d %d %d\n", batch, s1.n(), s2.n(), s4.n(), l1.n(), l2.n(),
         l3.n(), l4.n(), n2.n(), n1.n(), n0.n());
  std::vector<Case> cases = {
      {"L1 k3s1 64->64", &l1, &l1, 3, 64, 64, false},     {"L2 k3s1 128->128", &l2, &l2, 3, 128, 128, false},
      {"L3 k3s1 256->256", &l3, &l3, 3, 256, 256, false},  {"L4 k3s1 512->512", &l4, &l4, 3, 512, 512, false},
      {"L4 out 512->128", &l4, &l4, 3, 512, 128, false},   {"L1 k3s2 64->64", &s4, &l1, 3, 64, 64, false},
      {"L2 k3s2 64->128", &l1, &l2, 3, 64, 128, false},    {"L3 k3s2 128->256", &l2, &l3, 3, 128, 256, false},
      {"N2 k3s1 256->256", &n2, &n2, 3, 256, 256, true},   {"N2 out 256->128", &n2, &n2, 3, 256, 128, true},
      {"N1 k3s1 128->128", &n1, &n1, 3, 128, 128, true},   {"N0 k3s1 64->64", &n0, &n0, 3, 64, 64, true},
      {"N0 out 64->128", &n0, &n0, 3, 64, 128, true},      {"N0 dgrad 128->64", &n0, &n0, 3, 128, 64, true},
  };
  std::mt19937 rng(7);
  std::normal_distribution<float> Nf(0.f, 1.f);
  if (prio) { FC(fc_debug_set_prio(prio)); printf("# wave-priority experiment mode %d\n", prio); }
  if (mode == "density") {            // host only: MFMA work issued by a dense table at G-row skip granularity / exact pair work
    for (const Case& cs : cases) {
      if (!only.empty() && cs.name.find(only) == std::string::npos) continue;
      const int K = 27, n = cs.out->n();
      std::vector<int> nbr = kernel_map(*cs.in, *cs.out, cs.ks);
      std::vector<unsigned> masks(n, 0u); int64_t P = 0;
      for (int k = 0; k < K; ++k) for (int o = 0; o < n; ++o) if (nbr[(size_t)k * n + o] >= 0) { masks[o] |= 1u << k; ++P; }
      printf("%-18s n %7d occupancy %.2f | issued/useful at G =", cs.name.c_str(), n, (double)P / (27.0 * n));
      for (int sortmode = 0; sortmode < 3; ++sortmode) {
        std::vector<int> order(n); for (int i = 0; i < n; ++i) order[i] = i;
        if (sortmode == 1) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return masks[a] < masks[b]; });
        if (sortmode == 2) std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
          int pa = __builtin_popcount(masks[a]), pb = __builtin_popcount(masks[b]); return pa != pb ? pa > pb : masks[a] < masks[b]; });
        printf("  [%s]", sortmode == 0 ? "natural" : sortmode == 1 ? "mask" : "popc,mask");
        for (int G : {16, 32, 64, 128}) {
          int64_t issued = 0;
          for (int g0 = 0; g0 < n; g0 += G) { unsigned m = 0; for (int i = g0; i < std::min(n, g0 + G); ++i) m |= masks[order[i]]; issued += (int64_t)__builtin_popcount(m) * G; }
          printf(" %d:%.2f", G, (double)issued / P);
        }
      }
      printf("\n");
    }
    return 0;
  }
  if (mode == "hbm") {        // the bandwidth-bound kernels on benchmark-sized tensors: compulsory bytes / time vs the 8 TB/s peak
    Dev<float> x, y, gy, gx, gres, res, mean, var, cnt, gamma, beta, sums, rm, rv; Dev<long long> nbt; Dev<unsigned char> wsb; Dev<int> seg;
    struct T { const char* name; int64_t n; int C; bool segd; };
    std::vector<T> ts = {{"N0 64", n0.n(), 64, false}, {"N0 128", n0.n(), 128, false}, {"N1 128", n1.n(), 128, false},
                         {"stem IN 64 (8 scenes)", s2.n(), 64, true}, {"L1 64", l1.n(), 64, false}, {"L3 256", l3.n(), 256, false}};
    for (const T& t : ts) {
      const size_t e = (size_t)t.n * t.C;
      x.alloc(e); y.alloc(e); gy.alloc(e); gx.alloc(e); gres.alloc(e); res.alloc(e);
      std::vector<float> hx(e); for (auto& v : hx) v = Nf(rng);
      x.up(hx); gy.up(hx); res.up(hx);
      const int nseg = t.segd ? batch : 1;
      mean.alloc((size_t)nseg * t.C); var.alloc((size_t)nseg * t.C); cnt.alloc(nseg); gamma.alloc(t.C); beta.alloc(t.C); sums.alloc((size_t)nseg * 2 * t.C);
      rm.alloc(t.C); rv.alloc(t.C); nbt.alloc(1);
      CK(hipMemset(gamma.p, 0x3c, t.C * 4)); CK(hipMemset(beta.p, 0, t.C * 4));
      const int* segp = nullptr;
      if (t.segd) { std::vector<int> c4((size_t)t.n * 4); for (int64_t i = 0; i < t.n; ++i) { c4[i * 4] = s2.c[i].b; } seg.up(c4); segp = seg.p; }
      int64_t w1 = fc_col_stats_ws_bytes(t.n, t.C, nseg), w2 = fc_norm_act_bwd_ws_bytes(t.n, t.C, nseg), w3 = fc_bn_stats_ws_bytes(t.n, t.C);
      wsb.alloc((size_t)std::max(std::max(w1, w2), w3) + 256);
      const double mb = e * 4.0 / 1e6;
      auto report = [&](const char* k, double bytes_mb, double us) { printf("%-24s %-22s %8.1f MB %8.1f us %7.1f GB/s (%.3f of 8 TB/s)\n", t.name, k, bytes_mb, us, bytes_mb / us * 1e3, bytes_mb / us * 1e3 / 8000.0); };
      double us;
      us = time_us(reps, [&]() { FC(fc_col_stats(x.p, segp, 4, t.n, t.C, nseg, mean.p, var.p, cnt.p, wsb.p, w1, 0)); });
      report("fc_col_stats", mb, us);
      if (!t.segd) {
        us = time_us(reps, [&]() { FC(fc_bn_stats_train(x.p, t.n, t.C, 0.1f, mean.p, var.p, cnt.p, rm.p, rv.p, nbt.p, wsb.p, w3, 0)); });
        report("fc_bn_stats_train", mb, us);
      }
      us = time_us(reps, [&]() { FC(fc_norm_act_fwd(x.p, segp, 4, t.n, t.C, mean.p, var.p, 1e-5f, gamma.p, beta.p, nullptr, 2, y.p, 0)); });
      report("fc_norm_act_fwd", 2 * mb, us);
      us = time_us(reps, [&]() { FC(fc_norm_act_fwd(x.p, segp, 4, t.n, t.C, mean.p, var.p, 1e-5f, gamma.p, beta.p, res.p, 1, y.p, 0)); });
      report("fc_norm_act_fwd +res", 3 * mb, us);
      us = time_us(reps, [&]() { FC(fc_norm_act_bwd(x.p, y.p, gy.p, segp, 4, t.n, t.C, nseg, mean.p, var.p, cnt.p, 1e-5f, gamma.p, gamma.p, 2, gx.p, nullptr, sums.p, wsb.p, w2, 0)); });
      report("fc_norm_act_bwd", 4 * mb, us);
      us = time_us(reps, [&]() { FC(fc_norm_act_bwd(x.p, y.p, gy.p, segp, 4, t.n, t.C, nseg, mean.p, var.p, cnt.p, 1e-5f, gamma.p, gamma.p, 1, gx.p, gres.p, sums.p, wsb.p, w2, 0)); });
      report("fc_norm_act_bwd +gres", 5 * mb, us);
    }
    {   // stem conv (3 -> 64, k3s2), max-pool k2s2
      std::vector<int> nb = kernel_map(s1, s2, 3);
      Dev<int> dn; dn.up(nb);
      Dev<float> in3, w3, o64, g64, gw; in3.alloc((size_t)s1.n() * 3); w3.alloc(27 * 3 * 64); o64.alloc((size_t)s2.n() * 64); g64.alloc((size_t)s2.n() * 64); gw.alloc(27 * 3 * 64);
      CK(hipMemset(in3.p, 0x3c, (size_t)s1.n() * 12)); CK(hipMemset(w3.p, 0x3c, 27 * 3 * 64 * 4)); CK(hipMemset(g64.p, 0x3c, (size_t)s2.n() * 256));
      double bytes = 4.0 * ((double)s1.n() * 3 + (double)s2.n() * 64) + 4.0 * 27 * s2.n();
      double us = time_us(reps, [&]() { FC(fc_conv_fwd(in3.p, w3.p, dn.p, nullptr, o64.p, s1.n(), s2.n(), 27, 3, 64, 0, nullptr, 0, 0)); });
      printf("%-24s %-22s %8.1f MB %8.1f us %7.1f GB/s (%.3f of 8 TB/s)\n", "stem 3->64", "fc_conv_fwd", bytes / 1e6, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
      int64_t wb = fc_conv_wgrad_ws_bytes(s2.n(), 27, 3, 64, 0); wsb.alloc(wb + 256);
      us = time_us(reps, [&]() { FC(fc_conv_wgrad(in3.p, g64.p, dn.p, nullptr, gw.p, s1.n(), s2.n(), 27, 3, 64, 0, wsb.p, wb, 0)); });
      printf("%-24s %-22s %8.1f MB %8.1f us %7.1f GB/s (%.3f of 8 TB/s)\n", "stem 3->64", "fc_conv_wgrad", bytes / 1e6, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
      std::vector<int> nbp = kernel_map(s2, s4, 2);
      // k2s2 offsets are {0,1}^3 * stride: kernel_map() centres odd kernels only, ks = 2 gives offsets (d - 1): shift by +1
      {
        const int n = s4.n(); nbp.assign((size_t)8 * n, -1); int k = 0;
        for (int dz = 0; dz < 2; ++dz) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx, ++k)
          for (int o = 0; o < n; ++o) { const V4& v = s4.c[o]; nbp[(size_t)k * n + o] = s2.h.find(pack(V4{v.b, v.x + dx * 2, v.y + dy * 2, v.z + dz * 2})); }
      }
      Dev<int> dp, arg; dp.up(nbp); arg.alloc((size_t)s4.n() * 64);
      Dev<float> po; po.alloc((size_t)s4.n() * 64);
      bytes = 4.0 * 64 * ((double)s2.n() + 2.0 * s4.n()) + 4.0 * 8 * s4.n();
      us = time_us(reps, [&]() { FC(fc_maxpool_fwd(o64.p, dp.p, s4.n(), 8, 64, po.p, arg.p, 0)); });
      printf("%-24s %-22s %8.1f MB %8.1f us %7.1f GB/s (%.3f of 8 TB/s)\n", "maxpool k2s2 64", "fc_maxpool_fwd", bytes / 1e6, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
    }
    return 0;
  }
  if (mode == "gemm") {       // bounds: the LDS-tiled kernel as a plain dense GEMM, and as a conv whose neighbour is the row itself
    Dev<float> a, w, o; Dev<int> idt; Dev<unsigned char> wsb;
    const int n = n0.n();
    for (int Cout : {64, 128}) {
      const int Cin = 1728;
      a.alloc((size_t)n * Cin); w.alloc((size_t)27 * 64 * Cout); o.alloc((size_t)n * Cout);
      CK(hipMemset(a.p, 0x3c, (size_t)n * Cin * 4)); CK(hipMemset(w.p, 0x3c, (size_t)27 * 64 * Cout * 4));
      const double gf = 2.0 * n * Cin * Cout / 1e9;
      double us = time_us(reps, [&]() { FC(fc_conv_fwd(a.p, w.p, nullptr, nullptr, o.p, n, n, 1, Cin, Cout, 0, nullptr, 0, 0)); });
      printf("dense GEMM  %d x %d x %d          %9.1f us %7.1f TF\n", n, Cin, Cout, us, gf / us * 1e3);
      std::vector<int> same((size_t)27 * n);
      for (int k = 0; k < 27; ++k) for (int i = 0; i < n; ++i) same[(size_t)k * n + i] = i;
      idt.up(same);
      const double gf2 = 2.0 * 27 * n * 64.0 * Cout / 1e9;
      us = time_us(reps, [&]() { FC(fc_conv_fwd(a.p, w.p, idt.p, nullptr, o.p, n, n, 27, 64, Cout, 0, nullptr, 0, 0)); });
      printf("conv K=27, neighbour = own row, 64 -> %d  %9.1f us %7.1f TF\n", Cout, us, gf2 / us * 1e3);
      for (int k = 0; k < 27; ++k) for (int i = 0; i < n; ++i) same[(size_t)k * n + i] = (int)(((int64_t)i * 7919 + k * 104729) % n);
      idt.up(same);
      us = time_us(reps, [&]() { FC(fc_conv_fwd(a.p, w.p, idt.p, nullptr, o.p, n, n, 27, 64, Cout, 0, nullptr, 0, 0)); });
      printf("conv K=27, neighbour = scattered row, 64 -> %d  %9.1f us %7.1f TF\n", Cout, us, gf2 / us * 1e3);
    }
    return 0;
  }
  Dev<float> d_in, d_w, d_wt, d_out, d_ref, d_gout, d_gw, d_gwref; Dev<int> d_nbr, d_sorted, d_oidx, d_pi, d_po, d_pos, d_cnt, d_masks;
  Dev<unsigned char> d_ws, d_img;
  Dev<unsigned> d_amax; d_amax.alloc(512);
  for (const Case& cs : cases) {
    if (!only.empty() && cs.name.find(only) == std::string::npos) continue;
    const int K = cs.ks * cs.ks * cs.ks, n_in = cs.in->n(), n_out = cs.out->n(), Cin = cs.Cin, Cout = cs.Cout;
    std::vector<int> nbr = kernel_map(*cs.in, *cs.out, cs.ks);
    int64_t P = 0; for (int v : nbr) P += v >= 0;
    const double gflop = 2.0 * P * Cin * Cout / 1e9;
    std::vector<float> hin((size_t)n_in * Cin), hw((size_t)K * Cin * Cout), hg((size_t)n_out * Cout);
    for (auto& v : hin) v = Nf(rng); for (auto& v : hw) v = Nf(rng) * 0.05f; for (auto& v : hg) v = Nf(rng);
    d_in.up(hin); d_w.up(hw); d_gout.up(hg); d_nbr.up(nbr);
    { std::vector<float> hwt(hw.size());
      for (int k = 0; k < K; ++k) for (int ci = 0; ci < Cin; ++ci) for (int co = 0; co < Cout; ++co) hwt[((size_t)k * Cout + co) * Cin + ci] = hw[((size_t)k * Cin + ci) * Cout + co];
      d_wt.up(hwt); }
    if (Cin % 32 == 0 && Cout % 64 == 0) { d_img.alloc((size_t)fc_x6_weight_image_bytes(K, Cin, Cout)); FC(fc_x6_weight_image(d_w.p, d_img.p, K, Cin, Cout, 0, 0)); }
    d_out.alloc((size_t)n_out * Cout); d_ref.alloc((size_t)n_out * Cout); d_gw.alloc(hw.size()); d_gwref.alloc(hw.size());
    // derived tables (library kernels): mask-sorted rows, pair lists
    d_masks.alloc(n_out); d_sorted.alloc(nbr.size()); d_oidx.alloc(n_out);
    FC(fc_nbr_row_masks(d_nbr.p, n_out, K, d_masks.p, 0));
    std::vector<int> masks = d_masks.down(n_out), order(n_out);
    for (int i = 0; i < n_out; ++i) order[i] = i;
    if (popc_sort) std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      int pa = __builtin_popcount((unsigned)masks[a]), pb = __builtin_popcount((unsigned)masks[b]); return pa != pb ? pa > pb : masks[a] < masks[b]; });
    else std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return masks[a] < masks[b]; });
    d_oidx.up(order);
    FC(fc_permute_nbr(d_nbr.p, d_oidx.p, n_out, K, d_sorted.p, 0));
    d_pi.alloc(nbr.size()); d_po.alloc(nbr.size()); d_pos.alloc(nbr.size()); d_cnt.alloc(K);
    { int64_t wb = fc_kernel_map_pairs_ws_bytes(n_out, K); d_ws.alloc(wb); FC(fc_kernel_map_pairs(d_nbr.p, n_out, K, d_pi.p, d_po.p, d_pos.p, d_cnt.p, d_ws.p, wb, 0)); }
    CK(hipDeviceSynchronize());
    printf("%-18s n_in %7d n_out %7d P %9lld %7.1f GF (occupancy %.2f)\n", cs.name.c_str(), n_in, n_out, (long long)P, gflop, (double)P / ((double)K * n_out));
    auto ws_for = [&](int64_t b) { d_ws.alloc((size_t)std::max<int64_t>(b, 16)); return b; };

    if (mode == "all" || mode == "fwd") {
      // reference: generic FMA kernel
      std::vector<float> ref;
      if (check) {
        FC(fc_conv_fwd(d_in.p, d_w.p, d_nbr.p, nullptr, d_ref.p, n_in, n_out, K, Cin, Cout, 1, nullptr, 0, 0));
        ref = d_ref.down((size_t)n_out * Cout);
      }
      // fp64 on the host for 48 sampled output rows: how far each route is from the exact result (rms / max, of the output scale)
      std::vector<int> srows; std::vector<double> s64;
      if (check) {
        for (int q = 0; q < 48; ++q) srows.push_back((int)(((int64_t)q * 2654435761ll) % n_out));
        s64.assign(srows.size() * (size_t)Cout, 0.0);
        for (size_t q = 0; q < srows.size(); ++q)
          for (int k = 0; k < K; ++k) {
            const int i = nbr[(size_t)k * n_out + srows[q]];
            if (i < 0) continue;
            for (int ci = 0; ci < Cin; ++ci) {
              const double a = hin[(size_t)i * Cin + ci];
              const float* wr = &hw[((size_t)k * Cin + ci) * Cout];
              for (int co = 0; co < Cout; ++co) s64[q * Cout + co] += a * (double)wr[co];
            }
          }
      }
      struct Run { const char* what; int flags; int tbl; bool wt = false; bool img = false; int smode = 0; bool hint = false; };   // smode: fc_set_split_mode (image runs); hint: amax precomputed     // tbl: 0 plain table, 1 mask-sorted, 2 pair lists, 3 live-tile pair lists
      std::vector<Run> runs;
      const int X6 = 1 << 24, WTF = 1 << 23;
      runs.push_back({"plain ", 0, 0});
      if (!cs.dense) { runs.push_back({"sorted", 0, 1}); if (!x6_only) runs.push_back({"pairs ", 0, 2}); runs.push_back({"pairsL", 0, 3}); }
      runs.push_back({"plainT", WTF, 0, true});
      if (!x6_only) {
        runs.push_back({"pipe  ", 1 << 18, 0});
        if (!cs.dense) { runs.push_back({"pipeS ", 1 << 18, 1}); runs.push_back({"pipeL ", 1 << 18, 3}); }
        if (Cout == 64) { runs.push_back({"256x64", 3 << 4, 0}); if (!cs.dense) runs.push_back({"256x64s", 3 << 4, 1}); }
        runs.push_back({"glds  ", 1 << 21, 0});
        if (!cs.dense) { runs.push_back({"gldsS ", 1 << 21, 1}); runs.push_back({"gldsL ", 1 << 21, 3}); }
        if (Cout == 64) { runs.push_back({"glds256", (1 << 21) | (3 << 4), 0}); runs.push_back({"glds128", (1 << 21) | (2 << 4), 0}); }
      }
      // the split-bf16 kernel: weights split in the staging (x6), transposed weights (x6T), pre-split weight image (x6I), row orders
      const int X6I = X6 | (1 << 26);
      runs.push_back({"x6    ", X6, 0});
      runs.push_back({"x6T   ", X6 | WTF, 0, true});
      runs.push_back({"x6I   ", X6I, 0, false, true});
      if (!cs.dense) { runs.push_back({"x6S   ", X6, 1}); runs.push_back({"x6IS  ", X6I, 1, false, true}); runs.push_back({"x6L   ", X6, 3}); runs.push_back({"x6IL  ", X6I, 3, false, true}); }
      if (Cout == 64) { runs.push_back({"x6 128", X6 | (2 << 4), 0}); runs.push_back({"x6I128", X6I | (2 << 4), 0, false, true}); }
      // r6: the two-piece fp16 split (mode 2): same launches, image rebuilt in that mode; "+a": the operand's amax word handed in
      runs.push_back({"h3I   ", X6I, 0, false, true, 2});
      runs.push_back({"h3I+a ", X6I, 0, false, true, 2, true});
      if (!cs.dense) { runs.push_back({"h3IS  ", X6I, 1, false, true, 2}); runs.push_back({"h3IS+a", X6I, 1, false, true, 2, true}); runs.push_back({"h3IL  ", X6I, 3, false, true, 2}); runs.push_back({"h3IL+a", X6I, 3, false, true, 2, true}); }
      if (Cout == 64) runs.push_back({"h3I256+a", X6I | (3 << 4), 0, false, true, 2, true});       // 256 x 64 tiles
      if (Cout % 128 == 0 && n64) {      // r6: 64-column tiles on the launches that leave CUs empty with 128-column tiles
        runs.push_back({"x6I n64", X6I | (1 << 6), 0, false, true});
        if (!cs.dense) { runs.push_back({"x6ISn64", X6I | (1 << 6), 1, false, true}); runs.push_back({"x6ILn64", X6I | (1 << 6), 3, false, true}); }
      }
      static const char* snames[] = {"S=1", "S=2", "S=3", "S=4", "S=5", "S=6", "S=7", "S=8", "S=9", "S=10", "S=12", "S=14"};
      static const int svals[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14};
      if (s_sweep) for (int q = 0; q < 12; ++q) runs.push_back({snames[q], svals[q] << 8, cs.dense ? 0 : 1});
      for (const Run& r : runs) {
        const int fl = r.flags;
        const float* wp = r.img ? (const float*)d_img.p : (r.wt ? d_wt.p : d_w.p);
        if (r.img) { FC(fc_set_split_mode(r.smode)); FC(fc_x6_weight_image(d_w.p, d_img.p, K, Cin, Cout, 0, 0)); }
        const unsigned* hint = nullptr;
        if (r.hint) { CK(hipMemset(d_amax.p, 0, 2048)); FC(fc_amax(d_in.p, (int64_t)n_in * Cin, d_amax.p, 0)); hint = d_amax.p; }
        std::function<void()> fn;
        if (r.tbl == 3) {
          int64_t wb = ws_for(fc_conv_fwd_pairs_ws_bytes(n_out, K, Cout));
          std::vector<int> hc = d_cnt.down(K); int64_t live = 0; for (int v : hc) live += (v + 127) / 128;
          fn = [&, wb, fl, live, wp, hint]() { if (hint) FC(fc_conv_amax_hint(hint, nullptr)); FC(fc_conv_fwd_pairs_tiles(d_in.p, wp, d_pi.p, d_cnt.p, d_pos.p, d_out.p, n_in, n_out, K, Cin, Cout, live, fl, d_ws.p, wb, 0)); };
        } else if (r.tbl == 2) {
          int64_t wb = ws_for(fc_conv_fwd_pairs_ws_bytes(n_out, K, Cout));
          fn = [&, wb, fl, wp, hint]() { if (hint) FC(fc_conv_amax_hint(hint, nullptr)); FC(fc_conv_fwd_pairs(d_in.p, wp, d_pi.p, d_cnt.p, d_pos.p, d_out.p, n_in, n_out, K, Cin, Cout, fl, d_ws.p, wb, 0)); };
        } else {
          int64_t wb = ws_for(fc_conv_fwd_ws_bytes(n_out, K, Cin, Cout, fl));
          const int* tab = r.tbl ? d_sorted.p : d_nbr.p; const int* oi = r.tbl ? d_oidx.p : nullptr;
          fn = [&, wb, fl, tab, oi, wp, hint]() { if (hint) FC(fc_conv_amax_hint(hint, nullptr)); FC(fc_conv_fwd(d_in.p, wp, tab, oi, d_out.p, n_in, n_out, K, Cin, Cout, fl, d_ws.p, wb, 0)); };
        }
        CK(hipMemset(d_out.p, 0xff, (size_t)n_out * Cout * 4));
        fn(); CK(hipDeviceSynchronize());
        double err = -1;
        double e64rms = -1, e64max = -1;
        if (check) {
          std::vector<float> got = d_out.down((size_t)n_out * Cout);
          err = max_rel_err(got, ref);
          double sc = 0, ss = 0, mx = 0;
          for (double v : s64) sc = std::max(sc, std::fabs(v));
          for (size_t q = 0; q < srows.size(); ++q)
            for (int co = 0; co < Cout; ++co) { const double d = (double)got[(size_t)srows[q] * Cout + co] - s64[q * Cout + co]; ss += d * d; mx = std::max(mx, std::fabs(d)); }
          e64rms = std::sqrt(ss / s64.size()) / sc; e64max = mx / sc;
        }
        double us = time_us(reps, fn);
        printf("   fwd  flags %#x %s %9.1f us %7.1f TF  err %.2e  vs fp64 rms %.2e max %.2e%s\n", fl, r.what, us, gflop / us * 1e3, err, e64rms, e64max, (check && !(err < 1e-4)) ? "  <-- MISMATCH" : "");
        fflush(stdout);
      }
    }
    if (!trace_file.empty()) {
      typedef int (*trace_fn)(unsigned long long*, int);
      typedef int (*count_fn)(int*);
      trace_fn set = (trace_fn)dlsym(RTLD_DEFAULT, "fc_debug_trace_lds");
      count_fn cntf = nullptr;
      if (!set) { fprintf(stderr, "--trace needs tools/nbench_trace (FC_TRACE build)\n"); return 4; }
      const int cap = 1 << 18;
      Dev<unsigned long long> d_tr; d_tr.alloc((size_t)cap * 8);
      const int fl = trace_variant;      // flags of the traced launch
      std::function<void()> fn;
      if (trace_tbl == 2) {
        int64_t wb = ws_for(fc_conv_fwd_pairs_ws_bytes(n_out, K, Cout));
        fn = [&, wb, fl]() { FC(fc_conv_fwd_pairs(d_in.p, d_w.p, d_pi.p, d_cnt.p, d_pos.p, d_out.p, n_in, n_out, K, Cin, Cout, fl, d_ws.p, wb, 0)); };
      } else {
        int64_t wb = ws_for(fc_conv_fwd_ws_bytes(n_out, K, Cin, Cout, fl));
        const int* tab = trace_tbl ? d_sorted.p : d_nbr.p; const int* oi = trace_tbl ? d_oidx.p : nullptr;
        fn = [&, wb, fl, tab, oi]() { FC(fc_conv_fwd(d_in.p, d_w.p, tab, oi, d_out.p, n_in, n_out, K, Cin, Cout, fl, d_ws.p, wb, 0)); };
      }
      int wrate = 0; CK(hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0));
      double us_off = time_us(reps, fn);
      FC(set(d_tr.p, cap));
      double us_on = time_us(reps, fn);
      printf("   trace build: %.1f us per call with tracing off, %.1f us on; wall clock rate %d kHz\n", us_off, us_on, wrate);
      CK(hipMemset(d_tr.p, 0, (size_t)cap * 64));
      FC(set(d_tr.p, cap));
      fn(); CK(hipDeviceSynchronize());
      FC(set(nullptr, 0));
      std::vector<unsigned long long> all = d_tr.down((size_t)cap * 8), rec;
      for (int i = 0; i < cap; ++i) if (all[(size_t)i * 8 + 2]) rec.insert(rec.end(), all.begin() + (size_t)i * 8, all.begin() + (size_t)i * 8 + 8);
      int nrec = (int)(rec.size() / 8); (void)cntf;
      std::string fnm = trace_file + "." + cs.name.substr(0, cs.name.find(' ')) + ".v" + std::to_string(trace_variant) + "t" + std::to_string(trace_tbl) + ".bin";
      FILE* f = fopen(fnm.c_str(), "wb"); fwrite(rec.data(), 8, rec.size(), f); fclose(f);
      printf("   trace: %d wave records -> %s\n", nrec, fnm.c_str());
    }
    if (mode == "all" || mode == "wgrad") {
      std::vector<float> ref;
      if (check) {
        int64_t wb = ws_for(fc_conv_wgrad_ws_bytes(n_out, K, Cin, Cout, 1));
        FC(fc_conv_wgrad(d_in.p, d_gout.p, d_nbr.p, nullptr, d_gwref.p, n_in, n_out, K, Cin, Cout, 1, d_ws.p, wb, 0));
        ref = d_gwref.down(hw.size());
      }
      std::vector<int> wg_s = {0};
      if (s_sweep) wg_s = {0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32};
      for (int fsw : wg_s)
      for (int reg = 0; reg < (s_sweep ? 1 : 5); reg += (reg == 0 ? 2 : 1))          // 0: default, 2: r1 LDS kernel, 3: multi-offset kernel only under its first rule (Cin = 64, >= 32768 rows), 4: split-bf16
        for (int pairs = (s_sweep && !cs.dense ? 1 : 0); pairs < (cs.dense ? 1 : 2); ++pairs) {
          if (reg == 3 && pairs) continue;
          if (x6_only && (reg == 2 || reg == 3)) continue;
          for (int smode = 0; smode <= (reg == 4 ? 3 : 0); smode += (smode == 2 ? 1 : 2)) {       // 3: mode 2 with flat addresses (flags bit27)
          FC(fc_set_split_mode(smode == 3 ? 2 : smode));
          const int fl = ((reg == 2) << 16) | ((reg == 3) << 30) | ((reg == 4) << 24) | (fsw << 8) | ((smode == 3) << 27);
          int64_t wb = ws_for(fc_conv_wgrad_ws_bytes(n_out, K, Cin, Cout, fl));
          std::function<void()> fn;
          if (pairs) fn = [&, wb, fl]() { FC(fc_conv_wgrad_pairs(d_in.p, d_gout.p, d_pi.p, d_po.p, d_cnt.p, d_gw.p, n_in, n_out, K, Cin, Cout, fl, d_ws.p, wb, 0)); };
          else fn = [&, wb, fl]() { FC(fc_conv_wgrad(d_in.p, d_gout.p, d_nbr.p, nullptr, d_gw.p, n_in, n_out, K, Cin, Cout, fl, d_ws.p, wb, 0)); };
          CK(hipMemset(d_gw.p, 0xff, hw.size() * 4));
          fn(); CK(hipDeviceSynchronize());
          double err = -1;
          if (check) err = max_rel_err(d_gw.down(hw.size()), ref);
          double us = time_us(reps, fn);
          printf("   wgrad %s %s S=%-2d %9.1f us %7.1f TF  err %.2e%s\n", reg == 2 ? "ldsr1" : reg == 3 ? "m-r1 " : reg == 4 ? (smode == 3 ? "h3flt" : smode == 2 ? "h3   " : "x6   ") : "lds ", pairs ? "pairs" : "table", fsw, us, gflop / us * 1e3, err, (check && !(err < 2e-4)) ? "  <-- MISMATCH" : "");
          fflush(stdout);
          }
        }
    }
  }
  return 0;
}
