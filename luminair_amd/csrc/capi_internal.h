// Shared by the extern "C" translation units (capi.cpp, level2.cpp): the opaque context and the exception guard.
#pragma once
#include <mutex>
#include <new>
#include <string>

#include "prover.h"

struct lmn_ctx {
  lmn::Context* impl;
  std::string last_error;
  // Calls on ONE context are serialised (a context owns one stream, one arena and one pinned staging buffer);
  // different contexts run concurrently.  Recursive: a shard collective callback may call back into its context.
  std::recursive_mutex mu;
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
