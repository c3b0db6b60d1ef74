// kp1_host.hpp -- host-side helpers shared by the translation units of libkp1.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include "../../include/kp1.h"

namespace kp1 {
inline thread_local std::string g_last_error;
inline int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

// hipError_t -> kp1_status: the caller can tell "no GPU" from "out of memory" from "launch rejected" from "an earlier kernel faulted"
inline int status_of(hipError_t e) {
  switch (e) {
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorInsufficientDriver:
      return KP1_ERR_NO_DEVICE;
    case hipErrorOutOfMemory:
      return KP1_ERR_ALLOC;
    case hipErrorInvalidConfiguration:
    case hipErrorLaunchOutOfResources:
    case hipErrorInvalidDeviceFunction:
    case hipErrorSharedObjectInitFailed:
      return KP1_ERR_LAUNCH;
    default:
      return KP1_ERR_RUNTIME;
  }
}
// Launch check.  Kernel launches are asynchronous: hipGetLastError() only reports a launch the runtime refused, a fault inside the
// kernel surfaces at a later, unrelated call (as KP1_ERR_RUNTIME).  KP1_SYNC_CHECK=1 in the environment makes every entry point wait
// for its own kernels and report their faults itself (debugging aid: serialises the stream, cannot be used under hipGraph capture).
inline bool sync_check_enabled() {
  static const bool on = [] { const char* e = std::getenv("KP1_SYNC_CHECK"); return e && e[0] && e[0] != '0'; }();
  return on;
}
inline hipError_t launch_status() {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && sync_check_enabled()) e = hipDeviceSynchronize();
  return e;
}
// kp1_env.hip -> kp1_mlp.hip: the kernel-argument block of one fp32 env step (StepArgs<float> of kp1_env_step.inc, copied into `out`), for the
// rollout kernel that runs the policy forward and the env step in one launch.  Fails for fp64 handles and while reward components are on.
int env_step_args_f32(kp1_env* env, void* out, size_t out_bytes, const void* actions, float* obs, void* reward, uint8_t* done, float* terminal_obs,
                      int auto_reset, int* mode, int64_t* n_envs, int* device);
}  // namespace kp1

#define HIP_TRY(expr)                                                                                              \
  do {                                                                                                             \
    hipError_t _e = (expr);                                                                                        \
    if (_e != hipSuccess) return kp1::fail(kp1::status_of(_e), std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
