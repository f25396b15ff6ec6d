// Shared by the extern "C" translation units (capi.cpp, level2.cpp): the opaque context and the exception guard.
#pragma once
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "prover.h"

// lmn_prove_submit / lmn_prove_wait: one worker thread per context (created on first use) that runs the submitted proof
struct lmn_async {
  std::thread worker;
  std::mutex m;
  std::condition_variable cv;
  enum { IDLE, SUBMITTED, DONE, QUIT } state = IDLE;
  const lmn_table* tables = nullptr;
  size_t n_tables = 0;
  const lmn_settings* settings = nullptr;
  int rc = 0;
  std::vector<uint8_t> proof;
};

struct lmn_ctx {
  lmn::Context* impl;
  std::string last_error;
  // Calls on ONE context are serialised (a context owns one stream, one arena and one pinned staging buffer);
  // different contexts run concurrently.  Recursive: a shard collective callback may call back into its context.
  std::recursive_mutex mu;
  lmn_async* async = nullptr;
  std::mutex async_mu;   // guards the lazy creation of `async` (two first submits from different threads)
};

namespace lmn {
template <typename F>
int capi_guard(lmn_ctx* ctx, F&& f) {
  std::unique_lock<std::recursive_mutex> lock;
  if (ctx) lock = std::unique_lock<std::recursive_mutex>(ctx->mu);
  try {
    f();
    return LMN_OK;
  } catch (const LmnError& e) {
    if (ctx) ctx->last_error = e.what();
    int c = e.code;
    return (c == -100 || (c <= -1 && c >= -10)) ? c : LMN_ERR_INTERNAL;
  } catch (const std::bad_alloc&) {
    if (ctx) ctx->last_error = "host allocation failed";
    return LMN_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    if (ctx) ctx->last_error = e.what();
    return LMN_ERR_INTERNAL;
  }
}
}  // namespace lmn
