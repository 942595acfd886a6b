// fps_cells_probe.hip -- per-phase cycle accounting of the culled FPS rounds (tuning aid, not shipped).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFC_PROBE tools/fps_cells_probe.hip -o tools/fps_cells_probe.bin
// Run:   python -c "..." > cloud.bin (float32 n*3) ; tools/fps_cells_probe.bin cloud.bin n m [b]
#include "../pvn3d_amd/csrc/fps_cells.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

template <int SPC>
void probe(const std::vector<float>& h, int b, int n, int m) {
  float* d; int* idx; long long* dbg; int* ws;
  const int words = pvn3d_fps_cells_ws_words(n);
  hipMalloc(&d, (size_t)b * n * 12); hipMalloc(&idx, (size_t)b * m * 4); hipMalloc(&dbg, 128);
  hipMalloc(&ws, (size_t)b * words * 4);
  for (int i = 0; i < b; ++i) hipMemcpy(d + (size_t)i * n * 3, h.data(), (size_t)n * 12, hipMemcpyHostToDevice);
  int bs = 1 << (int)(log((double)n) / log(2.0)); if (bs > 512) bs = 512;
  int L = 0; while ((1 << L) < bs) ++L;
  const int Q = (n + bs - 1) / bs;
  const size_t lds = (size_t)(3 * 64 * SPC * 64 + FC_AUX_INTS) * 4;
  auto kern = fps_cells_kernel<SPC>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(b), dim3(256), lds, 0, n, m, L, Q, d, ws, idx, (int*)nullptr, dbg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hd[12]; hipMemcpy(hd, dbg, 96, hipMemcpyDeviceToHost);
    const double r = m - 1, cells = (double)hd[8];
    printf("n=%d m=%d b=%d: %.1f us, %.3f us/round | build %lld cyc, rounds %lld cyc (%.0f/round) | cells/round %.2f | "
           "per round: cull %.0f  loopctl %.0f  flush %.0f  Gred %.0f | per cell: wait+dist %.0f  dpp %.0f  locate %.0f  tail %.0f\n",
           n, m, b, ms * 1e3, ms * 1e3 / m, hd[10], hd[9], hd[9] / r, cells / r, hd[0] / r, hd[7] / r, hd[6] / r, hd[5] / r,
           hd[1] / cells, hd[2] / cells, hd[3] / cells, hd[4] / cells);
  }
}

int main(int argc, char** argv) {
  if (argc < 4) return 1;
  const int n = atoi(argv[2]), m = atoi(argv[3]), b = argc > 4 ? atoi(argv[4]) : 1;
  std::vector<float> h((size_t)n * 3);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(h.data(), 4, h.size(), f) != h.size()) { printf("cannot read cloud\n"); return 1; }
  fclose(f);
  const int spc = (n + 4095) / 4096;
  if (spc == 1) probe<1>(h, b, n, m);
  else if (spc == 2) probe<2>(h, b, n, m);
  else probe<3>(h, b, n, m);
  return 0;
}
