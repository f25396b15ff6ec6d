// TEST-ONLY kernel emulation runtime: runs the HIP kernel sources of luminair_amd/csrc on the host,
// one ucontext fiber per GPU thread, so that host orchestration and kernel indexing can be checked
// against the oracle on a machine without a GPU.  Never linked into the product library.
#include <ucontext.h>

#include <mutex>
#include <vector>

// Sanitizer builds (tests/emu/build_emu.sh asan | tsan): the sanitizers must be told about every stack switch, or
// AddressSanitizer reads a fiber's frames as overflows of the thread's stack and ThreadSanitizer its shadow stack as corrupt
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define LMN_EMU_ASAN 1
#endif
#if defined(__SANITIZE_THREAD__)
#include <sanitizer/tsan_interface.h>
#define LMN_EMU_TSAN 1
#endif

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
#ifdef LMN_EMU_ASAN
  void* fake_stack = nullptr;
#endif
#ifdef LMN_EMU_TSAN
  void* tsan_fiber = nullptr;
#endif
};
ucontext_t g_main;
std::vector<Fiber> g_fibers;
std::vector<unsigned char> g_stacks;
const std::function<void()>* g_body = nullptr;
int g_cur = -1;
#ifdef LMN_EMU_ASAN
void* g_main_fake = nullptr;
const void* g_main_bottom = nullptr;
size_t g_main_size = 0;
#endif
#ifdef LMN_EMU_TSAN
void* g_main_tsan = nullptr;
#endif

// fiber -> scheduler (`last`: the fiber never runs again)
void to_main(Fiber& f, bool last) {
#ifdef LMN_EMU_ASAN
  __sanitizer_start_switch_fiber(last ? nullptr : &f.fake_stack, g_main_bottom, g_main_size);
#endif
#ifdef LMN_EMU_TSAN
  __tsan_switch_to_fiber(g_main_tsan, 0);
#endif
  swapcontext(&f.ctx, &g_main);
#ifdef LMN_EMU_ASAN
  __sanitizer_finish_switch_fiber(f.fake_stack, &g_main_bottom, &g_main_size);
#endif
}
// scheduler -> fiber
void to_fiber(Fiber& f, void* stack, size_t stack_bytes) {
#ifdef LMN_EMU_ASAN
  __sanitizer_start_switch_fiber(&g_main_fake, stack, stack_bytes);
#endif
#ifdef LMN_EMU_TSAN
  __tsan_switch_to_fiber(f.tsan_fiber, 0);
#endif
  swapcontext(&g_main, &f.ctx);
#ifdef LMN_EMU_ASAN
  __sanitizer_finish_switch_fiber(g_main_fake, nullptr, nullptr);
#endif
}

void trampoline() {
#ifdef LMN_EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_main_bottom, &g_main_size);
#endif
  (*g_body)();
  g_fibers[g_cur].done = true;
  to_main(g_fibers[g_cur], true);
}
}  // namespace

void lmn_emu_syncthreads() { to_main(g_fibers[g_cur], false); }

void lmn_emu_run(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem) {
  // one emulated launch at a time in the process (the fibers, their stacks and the kernels' `static` shared arrays are
  // global): contexts driven from several host threads run their HOST code concurrently - which is what the thread
  // sanitizer build is for - and take turns on the "device"
  static std::mutex device_mu;
  std::lock_guard<std::mutex> device_lock(device_mu);
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
#ifdef LMN_EMU_TSAN
              g_main_tsan = __tsan_get_current_fiber();   // the launching thread's own context (launches take turns)
              if (f.tsan_fiber) __tsan_destroy_fiber(f.tsan_fiber);
              f.tsan_fiber = __tsan_create_fiber(0);
#endif
            }
        unsigned remaining = nthreads;
        while (remaining) {
          for (unsigned i = 0; i < nthreads; ++i) {
            Fiber& f = g_fibers[i];
            if (f.done) continue;
            g_cur = (int)i;
            threadIdx = f.tid;
            to_fiber(f, g_stacks.data() + (size_t)i * kStack, kStack);
            if (f.done) --remaining;
          }
        }
      }
  g_body = nullptr;
}
