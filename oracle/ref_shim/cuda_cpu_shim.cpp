// Fiber scheduler behind cuda_cpu_shim.h (see there).  TEST INFRASTRUCTURE ONLY.
//
// Resume order.  Fibers run from barrier to barrier, one after another, so "who runs first between
// two barriers" has to be chosen.  The reference's FPS kernel (sampling_gpu.cu:86-172) depends on it:
// after the last __syncthreads() of a round every thread reads `old = dists_i[0]`, and with no further
// barrier thread 0 overwrites dists_i[0] at the end of its next scan (:111-112).  On a GPU that is
// benign -- all warps read the slot hundreds of instructions before any warp finishes its scan -- but a
// scheduler that ran thread 0 first would let it overwrite the slot before the others have read it.
// Threads are therefore resumed in DESCENDING linear id: thread 0, the only writer of slot 0, runs last,
// which gives every thread the value the hardware gives it.  No other kernel of the reference
// communicates between threads except through atomicAdd.
#include "cuda_cpu_shim.h"

#include <ucontext.h>

#include <vector>

uint3_shim threadIdx, blockIdx;
dim3 blockDim(1, 1, 1), gridDim(1, 1, 1);

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
  ucontext_t ctx;
  char *stack = nullptr;
  bool done = false;
  uint3_shim tid;
};
ucontext_t g_sched;
std::vector<Fiber> g_fibers;
Fiber *g_cur = nullptr;
const std::function<void()> *g_body = nullptr;

void trampoline() {
  (*g_body)();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_sched);
}
}  // namespace

void __syncthreads() { swapcontext(&g_cur->ctx, &g_sched); }

namespace shim {
void run_grid(const LaunchCfg &cfg, const std::function<void()> &body) {
  const unsigned nthr = cfg.block.x * cfg.block.y * cfg.block.z;
  if (g_fibers.size() < nthr) {
    size_t old = g_fibers.size();
    g_fibers.resize(nthr);
    for (size_t i = old; i < nthr; ++i) g_fibers[i].stack = static_cast<char *>(malloc(kStack));
  }
  g_body = &body;
  gridDim = cfg.grid;
  blockDim = cfg.block;
  for (unsigned bz = 0; bz < cfg.grid.z; ++bz)
    for (unsigned by = 0; by < cfg.grid.y; ++by)
      for (unsigned bx = 0; bx < cfg.grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        unsigned t = 0;
        for (unsigned tz = 0; tz < cfg.block.z; ++tz)
          for (unsigned ty = 0; ty < cfg.block.y; ++ty)
            for (unsigned tx = 0; tx < cfg.block.x; ++tx, ++t) {
              Fiber &f = g_fibers[t];
              f.done = false;
              f.tid = {tx, ty, tz};
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = f.stack;
              f.ctx.uc_stack.ss_size = kStack;
              f.ctx.uc_link = nullptr;
              makecontext(&f.ctx, trampoline, 0);
            }
        // one pass = every live thread runs up to its next barrier (or to the end)
        for (unsigned live = nthr; live;) {
          live = 0;
          for (unsigned i = nthr; i-- > 0;) {  // descending: see the note on the FPS kernel above
            Fiber &f = g_fibers[i];
            if (f.done) continue;
            g_cur = &f;
            threadIdx = f.tid;
            swapcontext(&g_sched, &f.ctx);
            if (!f.done) ++live;
          }
        }
      }
  g_body = nullptr;
}
}  // namespace shim
