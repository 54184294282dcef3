// roctx.h — rocTX ranges around the C ABI's hot calls, bound at run time (no link-time dependency).
// The reference wraps the same calls in tracing spans (`embed_documents` / `embed_batch` / `embed_query`,
// src/embedder/core.rs:719,999,1096; `cagra_search`, src/cagra.rs:444); here a range shows up on the marker track of
// `rocprofv3 --marker-trace`.  Binding: a rocTX library already in the process (the profiler preloads one) is used as
// is; otherwise nothing is loaded and a range costs one predictable branch - unless CQS_HIP_ROCTX=1 asks for
// libroctx64 / librocprofiler-sdk-roctx explicitly.
#pragma once
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

namespace cqs_roctx {

struct Api {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};

inline const Api& api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "librocprofiler-sdk-roctx.so", "libroctx64.so"};
        void* h = nullptr;
        for (const char* n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        const char* want = getenv("CQS_HIP_ROCTX");
        if (!h && want && want[0] == '1')
            for (const char* n : names)
                if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        a.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        a.pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
    });
    return a;
}

struct Range {   // RAII: pops on every exit of the enclosing scope (early returns, exceptions)
    bool on;
    explicit Range(const char* name) : on(api().push != nullptr) {
        if (on) api().push(name);
    }
    ~Range() {
        if (on) api().pop();
    }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

}  // namespace cqs_roctx

#define CQS_ROCTX_RANGE(name) cqs_roctx::Range _cqs_roctx_range(name)
