// onnx_reader.cpp — weights out of an ONNX file without ONNX Runtime or protobuf: a hand-rolled reader of the
// protobuf wire format, restricted to what a weight loader needs.
//
// Why: the reference's local-weights hook (CQS_ONNX_DIR, src/embedder/download.rs:12-41) delivers exactly what ORT
// loads: `onnx/model.onnx` (structured layout, src/embedder/models.rs:455-457) or `model.onnx` (flat layout,
// download.rs:33-41) plus the external-data sidecar `model.onnx_data` next to it (download.rs:82).  For the HIP
// engine to "drop in unchanged" behind `create_session(model_path, ..)` (src/embedder/provider.rs:349-447) it must
// read its weights from those files.
//
// Wire subset (onnx.proto3): ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5;
// NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5;
// TensorProto.dims = 1, .data_type = 2, .float_data = 4, .name = 8, .raw_data = 9, .external_data = 13
// (StringStringEntryProto key = 1, value = 2: "location", "offset", "length"), .data_location = 14.
// Data types read: FLOAT (1), FLOAT16 (10), BFLOAT16 (16); anything else is skipped (not a weight we need).
//
// Name resolution: exporters keep parameter names for tensors used as they are (`model.embed_tokens.weight`,
// norm scales) but constant-fold `Linear` weights into anonymous, TRANSPOSED initialisers (`onnx::MatMul_123`,
// [in, out]) consumed as input 1 of a MatMul node whose NAME carries the module path
// (`/model/layers.0/self_attn/q_proj/MatMul`).  So: an initialiser fed to a MatMul as its second input is
// [K, N] and is handed out transposed under the module path + ".weight"; the two sentence-transformers Dense
// layers, whose node names vary by exporter, are recognised by shape ([hidden, dense_hidden] / [dense_hidden,
// hidden]).  Everything else keeps its own name (minus a leading "model.").
#include "onnx_reader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <map>

namespace cqs_onnx {

namespace {

struct Span { const uint8_t* p = nullptr; size_t n = 0; };

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    bool more() const { return ok && p < end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) { ok = false; return 0; }
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    // next field: number + wire type; for length-delimited fields `s` spans the payload, for varints `v` holds it
    bool field(uint32_t& num, uint32_t& wt, uint64_t& v, Span& s) {
        const uint64_t key = varint();
        if (!ok) return false;
        num = (uint32_t)(key >> 3);
        wt = (uint32_t)(key & 7);
        s = Span{};
        v = 0;
        switch (wt) {
            case 0: v = varint(); return ok;
            case 1: if ((size_t)(end - p) < 8) { ok = false; return false; } memcpy(&v, p, 8); p += 8; return true;
            case 2: {
                const uint64_t len = varint();
                if (!ok || len > (uint64_t)(end - p)) { ok = false; return false; }
                s = Span{p, (size_t)len};
                p += len;
                return true;
            }
            case 5: if ((size_t)(end - p) < 4) { ok = false; return false; } { uint32_t t; memcpy(&t, p, 4); v = t; } p += 4; return true;
            default: ok = false; return false;   // groups (3, 4) do not occur in ONNX
        }
    }
};

std::string str(const Span& s) { return std::string((const char*)s.p, s.n); }

struct Mapped {
    const uint8_t* p = nullptr;
    size_t n = 0;
    bool open(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); return false; }
        n = (size_t)st.st_size;
        if (n == 0) { close(fd); p = nullptr; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return false;
        p = (const uint8_t*)m;
        return true;
    }
    ~Mapped() { if (p) munmap((void*)p, n); }
};

float f16_to_f32(uint16_t h) {
    const uint32_t s = (h >> 15) & 1u, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s << 31;
        else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; ++sh; }
            u = (s << 31) | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3FFu) << 13);
        }
    } else if (e == 31) u = (s << 31) | 0x7F800000u | (m << 13);
    else u = (s << 31) | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Init {
    std::string name;
    std::vector<uint64_t> dims;
    int32_t dtype = 0;
    Span raw, float_data;
    bool external = false;
    std::string location;
    uint64_t ext_offset = 0, ext_length = 0;
    bool has_ext_length = false;
};

bool parse_tensor(const Span& t, Init& out) {
    Reader r(t.p, t.n);
    uint32_t num, wt; uint64_t v; Span s;
    while (r.more()) {
        if (!r.field(num, wt, v, s)) return false;
        switch (num) {
            case 1:   // dims: packed or one by one
                if (wt == 2) { Reader d(s.p, s.n); while (d.more()) { const uint64_t x = d.varint(); if (!d.ok) return false; out.dims.push_back(x); } }
                else if (wt == 0) out.dims.push_back(v);
                break;
            case 2: if (wt == 0) out.dtype = (int32_t)v; break;
            case 4: if (wt == 2) out.float_data = s; break;       // packed floats
            case 8: if (wt == 2) out.name = str(s); break;
            case 9: if (wt == 2) out.raw = s; break;
            case 13:
                if (wt == 2) {
                    Reader e(s.p, s.n);
                    std::string key, val;
                    uint32_t n2, w2; uint64_t v2; Span s2;
                    while (e.more()) {
                        if (!e.field(n2, w2, v2, s2)) return false;
                        if (n2 == 1 && w2 == 2) key = str(s2);
                        else if (n2 == 2 && w2 == 2) val = str(s2);
                    }
                    if (key == "location") out.location = val;
                    else if (key == "offset") out.ext_offset = strtoull(val.c_str(), nullptr, 10);
                    else if (key == "length") { out.ext_length = strtoull(val.c_str(), nullptr, 10); out.has_ext_length = true; }
                }
                break;
            case 14: if (wt == 0) out.external = (v == 1); break;
            default: break;
        }
    }
    return r.ok;
}

struct Node { std::string op, name; std::vector<std::string> inputs; };

bool parse_node(const Span& t, Node& out) {
    Reader r(t.p, t.n);
    uint32_t num, wt; uint64_t v; Span s;
    while (r.more()) {
        if (!r.field(num, wt, v, s)) return false;
        if (wt != 2) continue;
        if (num == 1) out.inputs.push_back(str(s));
        else if (num == 3) out.name = str(s);
        else if (num == 4) out.op = str(s);
    }
    return r.ok;
}

// "/model/layers.0/self_attn/q_proj/MatMul" -> "layers.0.self_attn.q_proj.weight"
std::string module_path_of(const std::string& node_name) {
    std::string s = node_name;
    const size_t cut = s.rfind('/');
    if (cut == std::string::npos) return std::string();
    s = s.substr(0, cut);                       // drop "/MatMul"
    while (!s.empty() && s[0] == '/') s.erase(0, 1);
    for (char& c : s) if (c == '/') c = '.';
    for (const char* pre : {"model.", "0.auto_model.", "auto_model.", "encoder."})
        if (s.rfind(pre, 0) == 0) { s = s.substr(strlen(pre)); break; }
    if (s.empty()) return s;
    return s + ".weight";
}

std::string strip_model(const std::string& n) {
    for (const char* pre : {"model.", "0.auto_model.", "auto_model."})
        if (n.rfind(pre, 0) == 0) return n.substr(strlen(pre));
    return n;
}

}  // namespace

int load(const std::string& model_path, uint32_t hidden, uint32_t dense_hidden, const Sink& sink, std::string& err) {
    Mapped model;
    if (!model.open(model_path)) { err = "cannot open " + model_path; return -1; }
    // ModelProto -> graph
    Span graph{};
    {
        Reader r(model.p, model.n);
        uint32_t num, wt; uint64_t v; Span s;
        while (r.more()) {
            if (!r.field(num, wt, v, s)) { err = "malformed ModelProto in " + model_path; return -1; }
            if (num == 7 && wt == 2) graph = s;
        }
    }
    if (!graph.p) { err = "no graph in " + model_path; return -1; }
    std::vector<Init> inits;
    std::vector<Node> nodes;
    {
        Reader r(graph.p, graph.n);
        uint32_t num, wt; uint64_t v; Span s;
        while (r.more()) {
            if (!r.field(num, wt, v, s)) { err = "malformed GraphProto in " + model_path; return -1; }
            if (wt != 2) continue;
            if (num == 5) {
                Init t;
                if (!parse_tensor(s, t)) { err = "malformed TensorProto in " + model_path; return -1; }
                inits.push_back(std::move(t));
            } else if (num == 1) {
                Node n;
                if (!parse_node(s, n)) { err = "malformed NodeProto in " + model_path; return -1; }
                if (n.op == "MatMul" && n.inputs.size() >= 2) nodes.push_back(std::move(n));
            }
        }
    }
    // initialiser name -> module path of the MatMul that consumes it as its [K, N] operand
    std::map<std::string, std::string> matmul_of;
    for (const Node& n : nodes) matmul_of[n.inputs[1]] = module_path_of(n.name);

    const std::string dir = model_path.substr(0, model_path.find_last_of('/') == std::string::npos ? 0 : model_path.find_last_of('/') + 1);
    std::map<std::string, Mapped> sidecars;
    std::vector<float> vals, tr;
    int fed = 0;
    for (const Init& t : inits) {
        if (t.dtype != 1 && t.dtype != 10 && t.dtype != 16) continue;   // not a float weight
        uint64_t count = 1;
        bool sane = !t.dims.empty();
        for (uint64_t d : t.dims) {
            if (d == 0 || count > (1ull << 40) / d) { sane = false; break; }
            count *= d;
        }
        if (!sane) continue;                                             // scalars / empty tensors: not weights
        const size_t esz = t.dtype == 1 ? 4 : 2;
        const uint8_t* data = nullptr;
        size_t avail = 0;
        if (t.external) {
            if (t.location.empty() || t.location.find("..") != std::string::npos || t.location[0] == '/') {
                err = "external data of " + t.name + " has an unsafe location";   // path escape (download.rs:17-29 spirit)
                return -1;
            }
            Mapped& m = sidecars[t.location];
            if (!m.p && !m.open(dir + t.location)) { err = "cannot open external data file " + dir + t.location; return -1; }
            if (t.ext_offset > m.n) { err = "external data offset of " + t.name + " is past the end of " + t.location; return -1; }
            data = m.p + t.ext_offset;
            avail = m.n - t.ext_offset;
            if (t.has_ext_length && t.ext_length < avail) avail = t.ext_length;
        } else if (t.raw.n) {
            data = t.raw.p; avail = t.raw.n;
        } else if (t.float_data.n && t.dtype == 1) {
            data = t.float_data.p; avail = t.float_data.n;
        } else continue;
        if (avail < count * esz) { err = "tensor " + t.name + " is truncated"; return -1; }
        vals.resize(count);
        if (esz == 4) memcpy(vals.data(), data, count * 4);
        else {
            for (uint64_t i = 0; i < count; ++i) {
                uint16_t h;
                memcpy(&h, data + 2 * i, 2);
                if (t.dtype == 16) { const uint32_t u = (uint32_t)h << 16; memcpy(&vals[i], &u, 4); }
                else vals[i] = f16_to_f32(h);
            }
        }
        // candidate names, most specific first: the consumer's module path / the tensor's own name, then the
        // sentence-transformers Dense layers by shape (their node names vary by exporter)
        std::vector<std::string> names;
        const float* out = vals.data();
        std::vector<uint64_t> dims = t.dims;
        const auto mm = matmul_of.find(t.name);
        if (mm != matmul_of.end() && t.dims.size() == 2) {
            // [K, N] operand of a MatMul -> hand out the Linear weight [N, K]
            const uint64_t K = t.dims[0], N = t.dims[1];
            tr.resize(count);
            for (uint64_t k = 0; k < K; ++k)
                for (uint64_t n = 0; n < N; ++n) tr[n * K + k] = vals[k * N + n];
            out = tr.data();
            dims = {N, K};
            if (!mm->second.empty()) names.push_back(mm->second);
        } else {
            names.push_back(strip_model(t.name));
        }
        if (dims.size() == 2 && dims[0] == dense_hidden && dims[1] == hidden) names.push_back("dense1.weight");
        if (dims.size() == 2 && dims[0] == hidden && dims[1] == dense_hidden) names.push_back("dense2.weight");
        for (const std::string& name : names) {
            const int rc = sink(name, out, count, dims);
            if (rc < 0) { err = "tensor " + t.name + " (as " + name + ") rejected"; return -1; }
            if (rc > 0) { fed += rc; break; }
        }
    }
    return fed;
}

}  // namespace cqs_onnx
