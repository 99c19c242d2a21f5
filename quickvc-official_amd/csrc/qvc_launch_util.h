// qvc_launch_util.h -- launch-side helper shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

namespace qvc {

// One-time (per kernel AND per device) opt-in for more than 64 KiB of dynamic LDS.  `done` is a per-kernel static
// bit mask over device ordinals: a process that drives several GPUs must set the attribute on each of them, and
// two host threads racing here just set it twice.  Not a stream operation, so it is legal during graph capture.
inline bool allow_big_lds(std::atomic<uint32_t>& done, const void* kernel) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const uint32_t bit = 1u << (dev & 31);
  if (done.load(std::memory_order_acquire) & bit) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  done.fetch_or(bit, std::memory_order_release);
  return true;
}

}  // namespace qvc
