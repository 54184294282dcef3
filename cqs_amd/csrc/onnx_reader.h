// onnx_reader.h — float initialisers of an ONNX model (+ external-data sidecar) handed to a sink under
// Hugging Face parameter names.  Internal to libcqs_hip.so; see onnx_reader.cpp.
#pragma once
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

namespace cqs_onnx {

// sink(name, data, count, dims) -> 1 consumed, 0 not a tensor the engine knows, < 0 error (aborts the load).
// `data` is f32 row-major with Linear weights as [out, in] (MatMul operands are transposed back).
using Sink = std::function<int(const std::string&, const float*, uint64_t, const std::vector<uint64_t>&)>;

// Returns the number of tensors the sink consumed, or -1 with `err` set.
int load(const std::string& model_path, uint32_t hidden, uint32_t dense_hidden, const Sink& sink, std::string& err);

}  // namespace cqs_onnx
