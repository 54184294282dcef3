// Driver for tests/test_abi.py::test_exception_barrier_turns_bad_alloc_into_a_status: an `extern "C"` entry point
// written exactly like the library's (CQS_ABI_TRY / CQS_ABI_CATCH around a body that runs the untrusted-file readers,
// i.e. the body of cqs_hip_embedder_load_dir without the device), with a TEST HOOK in place of the allocator: the
// replaced global operator new throws std::bad_alloc at its N-th call.  Built for the CPU with
// -fsanitize=address,undefined; every N must end in "rc=<status>", never in an abort / terminate.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>

#include "../cqs_amd/csrc/abi_guard.h"
#include "../cqs_amd/csrc/onnx_reader.h"
#include "../cqs_amd/csrc/safetensors_reader.h"

static long g_allocs_left = -1;   // < 0: never fail

void* operator new(size_t n) {
    if (g_allocs_left == 0) { g_allocs_left = -1; throw std::bad_alloc(); }   // one failed allocation, then memory is back
    if (g_allocs_left > 0) --g_allocs_left;
    void* p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new[](size_t n) { return operator new(n); }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, size_t) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }

struct fake_handle {
    std::mutex mu;
    std::string last_error;
};

extern "C" int32_t fake_load(fake_handle* h, const char* kind, const char* path, long fail_after) CQS_ABI_TRY {
    std::lock_guard<std::mutex> lk(h->mu);
    g_allocs_left = fail_after;
    std::string err;
    unsigned long long floats = 0;
    int fed;
    if (std::string(kind) == "onnx")
        fed = cqs_onnx::load(path, 64, 128, [&](const std::string&, const float*, uint64_t n, const std::vector<uint64_t>&) { floats += n; return 1; }, err);
    else
        fed = cqs_st::load(path, [&](const std::string&, const float*, uint64_t n) { floats += n; return 1; }, err);
    g_allocs_left = -1;
    if (fed < 0) { h->last_error = err; return CQS_HIP_ERR_INVALID; }
    return fed;
} CQS_ABI_CATCH(h)

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    fake_handle h;
    const int32_t rc = fake_load(&h, argv[1], argv[2], atol(argv[3]));
    g_allocs_left = -1;
    printf("rc=%d msg=%s\n", rc, h.last_error.c_str());
    return 0;
}
