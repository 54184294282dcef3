// abi_guard.h — the exception barrier of the C ABI (include/cqs_hip.h: "no exception crosses the boundary").
// Every `extern "C"` entry point of libcqs_hip.so is a function-try-block:
//
//     int32_t cqs_hip_foo(cqs_hip_index* x, ...) CQS_ABI_TRY { ...body... } CQS_ABI_CATCH(x)
//
// so a std::bad_alloc / std::length_error out of a std::vector, std::string, std::map or `new` inside the body (host
// out-of-memory at 10M rows of id lists is not hypothetical) comes back as a status code + `last_error`, not as an
// unwind through C frames into the Rust daemon (= abort).  The reference's convention for a backend is the same: never
// panic out of a search, log and fall through to the next backend (src/cagra.rs:445-470, :1797-1800).
//   CQS_ABI_CATCH(h)        int32_t entry points with a (possibly NULL) mutable handle `h` that has `mu` + `last_error`
//   CQS_ABI_CATCH_NOHANDLE  int32_t entry points without one (constructors: *out stays NULL)
//   CQS_ABI_CATCH_VAL(v)    value getters (uint32_t / uint64_t / size_t / float / const char*): return `v`
//   CQS_ABI_CATCH_VOID      void entry points
// tests/test_abi.py greps the sources: an exported definition without the pair fails the CPU suite.
#pragma once
#include <cstdint>
#include <exception>
#include <mutex>
#include <new>
#include <string>

#include "../../include/cqs_hip.h"

namespace cqs_abi {

// Lippincott function: called inside a catch (...) handler, maps the in-flight exception to a status code and a short
// message (static storage or the exception's own what(): copied into `buf`, never allocating).
inline int32_t classify(char* buf, size_t cap) noexcept {
    auto put = [&](const char* a, const char* b) {
        size_t n = 0;
        for (const char* s : {a, b})
            for (; s && *s && n + 1 < cap; ++s) buf[n++] = *s;
        if (cap) buf[n] = 0;
    };
    try {
        throw;
    } catch (const std::bad_alloc&) {
        put("out of host memory (std::bad_alloc)", nullptr);
        return CQS_HIP_ERR_NOMEM;
    } catch (const std::exception& ex) {
        put("C++ exception stopped at the C ABI: ", ex.what());
        return CQS_HIP_ERR_INVALID;
    } catch (...) {
        put("unknown C++ exception stopped at the C ABI", nullptr);
        return CQS_HIP_ERR_INVALID;
    }
}

template <class H>
inline int32_t on_exception(H* h) noexcept {
    char msg[192];
    const int32_t code = classify(msg, sizeof msg);
    if (h) {
        try {   // the body's lock_guard was released by the unwind; assigning a short string may itself throw: then no message
            std::lock_guard<std::mutex> g(h->mu);
            h->last_error.assign(msg);
        } catch (...) {
        }
    }
    return code;
}
inline int32_t on_exception_nohandle() noexcept {
    char msg[192];
    return classify(msg, sizeof msg);
}

}  // namespace cqs_abi

#define CQS_ABI_TRY try
#define CQS_ABI_CATCH(h) catch (...) { return cqs_abi::on_exception(h); }
#define CQS_ABI_CATCH_NOHANDLE catch (...) { return cqs_abi::on_exception_nohandle(); }
#define CQS_ABI_CATCH_VAL(v) catch (...) { (void)cqs_abi::on_exception_nohandle(); return (v); }
#define CQS_ABI_CATCH_VOID catch (...) { (void)cqs_abi::on_exception_nohandle(); }
