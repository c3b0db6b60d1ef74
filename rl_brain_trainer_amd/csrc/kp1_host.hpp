// kp1_host.hpp -- host-side helpers shared by the translation units of libkp1.so.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/kp1.h"

namespace kp1 {
inline thread_local std::string g_last_error;
inline int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
}  // namespace kp1

#define HIP_TRY(expr)                                                                                              \
  do {                                                                                                             \
    hipError_t _e = (expr);                                                                                        \
    if (_e != hipSuccess) return kp1::fail(KP1_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));   \
  } while (0)
