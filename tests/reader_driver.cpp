// Driver for tests/test_readers_sanitized.py: runs the two untrusted-file readers of libcqs_hip.so (ONNX initialisers,
// safetensors) over a file and prints how many tensors / floats they handed out.  Built with -fsanitize=address,undefined
// on the CPU (GPU sanitizers are not available on the pool); a malformed file must end in a clean "error:" line.
#include <cstdio>
#include <cstring>
#include <string>

#include "../cqs_amd/csrc/onnx_reader.h"
#include "../cqs_amd/csrc/safetensors_reader.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string kind = argv[1], path = argv[2];
    std::string err;
    unsigned long long floats = 0;
    double sum = 0.0;
    int fed;
    if (kind == "onnx") {
        fed = cqs_onnx::load(path, 64, 128, [&](const std::string&, const float* d, uint64_t n, const std::vector<uint64_t>&) {
            floats += n;
            for (uint64_t i = 0; i < n; i += (n / 64 + 1)) sum += d[i];     // touch the data (ASAN checks the reads)
            if (n) sum += d[n - 1];
            return 1;
        }, err);
    } else {
        fed = cqs_st::load(path, [&](const std::string&, const float* d, uint64_t n) {
            floats += n;
            for (uint64_t i = 0; i < n; i += (n / 64 + 1)) sum += d[i];
            if (n) sum += d[n - 1];
            return 1;
        }, err);
    }
    if (fed < 0) { printf("error: %s\n", err.c_str()); return 0; }
    printf("ok: %d tensors, %llu floats, checksum %.6g\n", fed, floats, sum);
    return 0;
}
