// launch_util.h — host-side launch helpers shared by the scan and the forward kernels.  Internal to libcqs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace cqs {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE and costs a driver call: issue it once per (kernel,
// device) - again only if a later launch of the same kernel asks for MORE - instead of on every launch.  One object per
// call site (a function-local static: one per template instantiation).  Devices >= 64 always take the slow path.
struct DynLdsOnce {
    std::atomic<uint32_t> set_bytes[64] = {};
    hipError_t ensure(const void* kernel, size_t bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && set_bytes[dev].load(std::memory_order_acquire) >= (uint32_t)bytes) return hipSuccess;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess && dev >= 0 && dev < 64) {
            uint32_t cur = set_bytes[dev].load(std::memory_order_relaxed);
            while (cur < (uint32_t)bytes && !set_bytes[dev].compare_exchange_weak(cur, (uint32_t)bytes, std::memory_order_release)) {}
        }
        return e;
    }
};

}  // namespace cqs
