// safetensors_reader.cpp — minimal safetensors reader (header = u64 length + JSON {"name":{"dtype","shape",
// "data_offsets"}}), shared by the embedding engine's and the BERT engines' `load_dir`.  Hugging Face checkpoint
// directories are what `ensure_model` (src/embedder/download.rs) leaves on disk next to the ONNX export.
#include "safetensors_reader.h"

#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

namespace cqs_st {

namespace {

float bf16_bits_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
float f16_bits_to_f32(uint16_t h) {
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

struct StEntry { std::string dtype; std::vector<uint64_t> shape; uint64_t lo = 0, hi = 0; };

bool parse_safetensors_header(const std::string& js, std::map<std::string, StEntry>& out) {
    size_t i = 0;
    auto skip = [&]() { while (i < js.size() && (js[i] == ' ' || js[i] == '\n' || js[i] == '\t' || js[i] == '\r')) ++i; };
    auto str = [&](std::string& s) -> bool {
        skip();
        if (i >= js.size() || js[i] != '"') return false;
        ++i; s.clear();
        while (i < js.size() && js[i] != '"') { if (js[i] == '\\' && i + 1 < js.size()) ++i; s.push_back(js[i++]); }
        if (i >= js.size()) return false;
        ++i; return true;
    };
    auto num = [&](uint64_t& v) -> bool {
        skip();
        if (i >= js.size() || js[i] < '0' || js[i] > '9') return false;
        v = 0;
        while (i < js.size() && js[i] >= '0' && js[i] <= '9') v = v * 10 + (uint64_t)(js[i++] - '0');
        return true;
    };
    auto numlist = [&](std::vector<uint64_t>& v) -> bool {
        skip();
        if (i >= js.size() || js[i] != '[') return false;
        ++i; v.clear(); skip();
        if (i < js.size() && js[i] == ']') { ++i; return true; }
        for (;;) {
            uint64_t x;
            if (!num(x)) return false;
            v.push_back(x); skip();
            if (i < js.size() && js[i] == ',') { ++i; continue; }
            if (i < js.size() && js[i] == ']') { ++i; return true; }
            return false;
        }
    };
    // skip a JSON value we do not care about (the __metadata__ object)
    std::function<bool()> skipval = [&]() -> bool {
        skip();
        if (i >= js.size()) return false;
        if (js[i] == '"') { std::string t; return str(t); }
        if (js[i] == '{' || js[i] == '[') {
            const char open = js[i], close = open == '{' ? '}' : ']';
            ++i; skip();
            if (i < js.size() && js[i] == close) { ++i; return true; }
            for (;;) {
                if (open == '{') { std::string k; if (!str(k)) return false; skip(); if (js[i++] != ':') return false; }
                if (!skipval()) return false;
                skip();
                if (i < js.size() && js[i] == ',') { ++i; continue; }
                if (i < js.size() && js[i] == close) { ++i; return true; }
                return false;
            }
        }
        while (i < js.size() && js[i] != ',' && js[i] != '}' && js[i] != ']') ++i;
        return true;
    };
    skip();
    if (i >= js.size() || js[i] != '{') return false;
    ++i;
    for (;;) {
        skip();
        if (i < js.size() && js[i] == '}') return true;
        std::string name;
        if (!str(name)) return false;
        skip();
        if (i >= js.size() || js[i++] != ':') return false;
        if (name == "__metadata__") { if (!skipval()) return false; }
        else {
            skip();
            if (i >= js.size() || js[i++] != '{') return false;
            StEntry en;
            for (;;) {
                std::string k;
                if (!str(k)) return false;
                skip();
                if (i >= js.size() || js[i++] != ':') return false;
                if (k == "dtype") { if (!str(en.dtype)) return false; }
                else if (k == "shape") { if (!numlist(en.shape)) return false; }
                else if (k == "data_offsets") {
                    std::vector<uint64_t> o;
                    if (!numlist(o) || o.size() != 2) return false;
                    en.lo = o[0]; en.hi = o[1];
                } else if (!skipval()) return false;
                skip();
                if (i < js.size() && js[i] == ',') { ++i; continue; }
                if (i < js.size() && js[i] == '}') { ++i; break; }
                return false;
            }
            out[name] = en;
        }
        skip();
        if (i < js.size() && js[i] == ',') { ++i; continue; }
        if (i < js.size() && js[i] == '}') return true;
        return false;
    }
}

// Feed every tensor of a safetensors file to set_tensor, renaming with `rename(name)` ("" = skip).

}  // namespace

int load(const std::string& path, const Sink& sink, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return -1; }
    uint64_t hl = 0;
    if (fread(&hl, 8, 1, f) != 1 || hl > (64ull << 20)) { fclose(f); err = "bad safetensors header in " + path; return -1; }
    std::string js(hl, '\0');
    if (fread(&js[0], 1, hl, f) != hl) { fclose(f); err = "truncated header in " + path; return -1; }
    std::map<std::string, StEntry> ents;
    if (!parse_safetensors_header(js, ents)) { fclose(f); err = "cannot parse header of " + path; return -1; }
    struct stat fst;
    const uint64_t file_size = fstat(fileno(f), &fst) == 0 ? (uint64_t)fst.st_size : 0;
    std::vector<uint8_t> raw;
    std::vector<float> vals;
    int fed = 0;
    for (auto& kv : ents) {
        const StEntry& en = kv.second;
        uint64_t count = 1;
        bool sane = en.lo <= en.hi && en.hi <= file_size - std::min<uint64_t>(file_size, 8 + hl);   // offsets inside the file
        for (uint64_t d : en.shape) {
            if (d != 0 && count > (1ull << 40) / d) { sane = false; break; }   // untrusted dims: no overflow, no absurd resize
            count *= d;
        }
        const uint64_t esz = en.dtype == "F32" ? 4 : ((en.dtype == "BF16" || en.dtype == "F16") ? 2 : 0);
        if (!esz) continue;                                                    // integer buffers (position_ids ...): not weights
        if (!sane || en.hi - en.lo != count * esz) { fclose(f); err = "unsupported dtype/shape for " + kv.first + " in " + path; return -1; }
        raw.resize(count * esz);
        if (fseek(f, (long)(8 + hl + en.lo), SEEK_SET) != 0 || fread(raw.data(), 1, raw.size(), f) != raw.size()) {
            fclose(f);
            err = "truncated data for " + kv.first + " in " + path;
            return -1;
        }
        vals.resize(count);
        if (esz == 4) memcpy(vals.data(), raw.data(), count * 4);
        else {
            const uint16_t* h = (const uint16_t*)raw.data();
            for (uint64_t i = 0; i < count; ++i) vals[i] = en.dtype == "BF16" ? bf16_bits_to_f32(h[i]) : f16_bits_to_f32(h[i]);
        }
        const int rc = sink(kv.first, vals.data(), count);
        if (rc < 0) { fclose(f); err = "tensor " + kv.first + " rejected"; return -1; }
        fed += rc;
    }
    fclose(f);
    return fed;
}

}  // namespace cqs_st
