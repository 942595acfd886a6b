// Exports the reference's inline launch heuristic opt_n_threads (cuda_utils.h:15-19), compiled from
// the reference's own header.  TEST INFRASTRUCTURE ONLY.
#include "cuda_utils.h"
extern "C" int ref_opt_n_threads(int w) { return opt_n_threads(w); }
extern "C" void ref_opt_block_config(int x, int y, int *out_xy) {
  dim3 c = opt_block_config(x, y);
  out_xy[0] = static_cast<int>(c.x);
  out_xy[1] = static_cast<int>(c.y);
}
