// TEST-ONLY kernel emulation runtime: runs the HIP kernel sources of luminair_amd/csrc on the host,
// one ucontext fiber per GPU thread, so that host orchestration and kernel indexing can be checked
// against the oracle on a machine without a GPU.  Never linked into the product library.
#include <ucontext.h>

#include <vector>

#include "../../luminair_amd/csrc/platform.h"

thread_local uint3e threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
thread_local unsigned char* lmn_emu_dyn_smem = nullptr;
unsigned lmn_emu_shfl_scratch[1024];

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
  ucontext_t ctx;
  bool done;
  uint3e tid;
};
ucontext_t g_main;
std::vector<Fiber> g_fibers;
std::vector<unsigned char> g_stacks;
const std::function<void()>* g_body = nullptr;
int g_cur = -1;

void trampoline() {
  (*g_body)();
  g_fibers[g_cur].done = true;
  swapcontext(&g_fibers[g_cur].ctx, &g_main);
}
}  // namespace

void lmn_emu_syncthreads() { swapcontext(&g_fibers[g_cur].ctx, &g_main); }

void lmn_emu_run(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem) {
  const unsigned nthreads = block.x * block.y * block.z;
  if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
  if (g_stacks.size() < (size_t)nthreads * kStack) g_stacks.resize((size_t)nthreads * kStack);
  std::vector<unsigned char> dyn(smem ? smem : 16);
  lmn_emu_dyn_smem = dyn.data();
  g_body = &body;
  blockDim = block;
  gridDim = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        unsigned t = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
              Fiber& f = g_fibers[t];
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = g_stacks.data() + (size_t)t * kStack;
              f.ctx.uc_stack.ss_size = kStack;
              f.ctx.uc_link = &g_main;
              f.done = false;
              f.tid = {tx, ty, tz};
              makecontext(&f.ctx, trampoline, 0);
            }
        unsigned remaining = nthreads;
        while (remaining) {
          for (unsigned i = 0; i < nthreads; ++i) {
            Fiber& f = g_fibers[i];
            if (f.done) continue;
            g_cur = (int)i;
            threadIdx = f.tid;
            swapcontext(&g_main, &f.ctx);
            if (f.done) --remaining;
          }
        }
      }
  g_body = nullptr;
}
