// issue_bench.hip -- what a wave that runs ALONE on its SIMD pays per instruction (tuning aid, not shipped).
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_bench.hip -o tools/issue_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP10(x) x x x x x x x x x x
#define REP100(x) REP10(REP10(x))

template <int K>
__global__ void bench(long long* out, int n, int zero) {
  int a = threadIdx.x, s = zero, b = threadIdx.x * 3;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
    if (K == 0) { REP100(asm volatile("v_add_u32 %0, %0, 1" : "+v"(a));) }                         // dependent VALU
    if (K == 1) { REP100(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1" : "+v"(a), "+v"(b));) }  // 2 chains
    if (K == 2) { REP100(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");) }                         // dependent SALU
    if (K == 3) { REP100(asm volatile("s_cmp_eq_u32 %0, 12345\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:" : "+s"(s) :: "scc");) }  // untaken branch
    if (K == 4) { REP100(asm volatile("s_cmp_lg_u32 %0, 12345\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1: s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");) }  // taken branch (skips 1)
    if (K == 5) { REP100(asm volatile("v_readlane_b32 %0, %1, 3\n s_nop 3\n v_add_u32 %1, %1, %0" : "+s"(s), "+v"(a));) }   // readlane -> VALU chain
    if (K == 6) { REP100(asm volatile("v_cmp_eq_u32 vcc, %1, %0\n s_and_b64 vcc, vcc, exec\n s_cmp_lg_u64 vcc, 0\n s_cselect_b32 %0, 1, 2" : "+s"(s) : "v"(a) : "vcc", "scc");) }  // ballot-style chain
    if (K == 7) { REP100(asm volatile("s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));) }  // dpp step
    if (K == 8) { REP100(asm volatile("v_add_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3" : "+v"(a), "+v"(b), "+v"(s), "+v"(zero));) }  // 4 independent VALU
    if (K == 9) { REP100(asm volatile("s_cmp_lg_u32 %0, 12345\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n1: s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");) }  // taken branch over 20 instr
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + s + b; }
}

template <int K>
void run(const char* name, int per_rep) {
  long long* d; hipMalloc(&d, 16);
  const int n = 100;
  hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64), 0, 0, d, n, 0);
  hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64), 0, 0, d, n, 0);
  long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-44s %7.2f cycles per group (%d instr) = %.2f per instr\n", name, h[0] / (100.0 * n), per_rep, h[0] / (100.0 * n * per_rep));
  fflush(stdout); hipFree(d);
}

int main() {
  run<0>("dependent v_add_u32", 1);
  run<1>("two interleaved v_add chains", 2);
  run<8>("4 independent VALU (dep on self each rep)", 4);
  run<2>("dependent s_add_u32", 1);
  run<3>("s_cmp + untaken s_cbranch + s_add", 3);
  run<4>("s_cmp + TAKEN s_cbranch (skip 1) + s_add", 3);
  run<9>("s_cmp + TAKEN s_cbranch (skip 20) + s_add", 3);
  run<5>("v_readlane + s_nop 3 + dependent v_add", 3);
  run<6>("v_cmp + s_and + s_cmp + s_cselect chain", 4);
  run<7>("s_nop 1 + v_max_i32_dpp (dependent)", 2);
  return 0;
}
