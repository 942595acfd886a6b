// fps_probe.hip -- per-phase cycle accounting of the FPS round (tuning aid, not shipped).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DPVN3D_FPS_PROBE tools/fps_probe.hip -o tools/fps_probe.bin
#include "../pvn3d_amd/csrc/sampling.hip"
#include <cstdio>
#include <vector>
#include <random>

template <int THREADS, int PPT>
void probe(int b, int n, int m) {
  std::vector<float> h((size_t)b * n * 3);
  std::mt19937 g(1);
  std::uniform_real_distribution<float> u(0.2f, 1.0f);
  for (auto& v : h) v = u(g);
  float* d; int* idx; long long* dbg;
  hipMalloc(&d, h.size() * 4); hipMalloc(&idx, (size_t)b * m * 4); hipMalloc(&dbg, 64);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int bs = pvn3d_opt_n_threads(n);
  int L = 0; while ((1 << L) < bs) ++L;
  const int Q = (n + bs - 1) / bs;
  size_t lds = (size_t)THREADS * PPT * 12;
  auto kern = fps_reg_kernel<THREADS, PPT, true>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(b), dim3(THREADS), lds, 0, n, m, L, Q, d, idx, FpsNest{nullptr, nullptr, 0}, dbg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hd[6]; hipMemcpy(hd, dbg, 48, hipMemcpyDeviceToHost);
    double tot = 0; for (int i = 0; i < 5; ++i) tot += hd[i];
    printf("T=%d PPT=%d n=%d m=%d b=%d: %.1f us total, %.3f us/round | cycles/round: scan %.0f tree %.0f wave-red %.0f xwave %.0f fetch %.0f  sum %.0f | wall100MHz ticks %lld -> shader clk %.2f GHz\n",
           THREADS, PPT, n, m, b, ms * 1e3, ms * 1e3 / m, hd[0] / (double)m, hd[1] / (double)m, hd[2] / (double)m,
           hd[3] / (double)m, hd[4] / (double)m, tot / m, hd[5], tot / (hd[5] * 10.0));
  }
}

int main() {
  probe<64, 8>(64, 512, 128);
  probe<256, 4>(64, 1024, 512);
  probe<256, 8>(64, 2048, 1024);
  probe<256, 48>(64, 12288, 2048);
  probe<256, 48>(256, 12288, 2048);
  return 0;
}
