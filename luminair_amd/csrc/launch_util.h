// Host-side launch helpers shared by the kernel translation units.
#pragma once
#include "platform.h"

#ifndef LMN_EMU
#include <mutex>
#include <set>
#include <utility>
namespace lmn {
// Kernels that use more than 64 KiB of dynamic LDS need the attribute once per (device, function): contexts of one
// process may live on several GPUs and are created from several threads.
inline void allow_big_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  LMN_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, fn})) return;
  LMN_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, fn});
}
}  // namespace lmn
#endif
