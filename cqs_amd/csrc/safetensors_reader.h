// safetensors_reader.h — tensors of a .safetensors file handed to a sink as f32.  Internal to libcqs_hip.so.
#pragma once
#include <stdint.h>

#include <functional>
#include <string>

namespace cqs_st {

// sink(name, data f32 row-major, count) -> 1 consumed, 0 skipped, < 0 error (aborts the load).
using Sink = std::function<int(const std::string&, const float*, uint64_t)>;

// F32 / BF16 / F16 tensors; untrusted header: offsets and shapes are bounds-checked.  Returns the number of tensors
// the sink consumed, or -1 with `err` set.
int load(const std::string& path, const Sink& sink, std::string& err);

}  // namespace cqs_st
