"""The Rust shim (rust_shim/src/*.rs) cannot be compiled in this image (no cargo / rustc), so nothing but this test
ties its `extern "C"` blocks and `#[repr(C)]` structs to include/cqs_hip.h - the C ABI it would link against behind the
reference's traits (`VectorIndex` / `IndexBackend`, /root/reference/src/index.rs:139-291; `Embedder`; `SpladeEncoder`).

Checked mechanically, for every function the shim declares:
  * the header declares it, with the same number of arguments,
  * every argument type and the return type agree after mapping Rust FFI types to C
    (`*const f32` <-> `const float*`, `u64` <-> `uint64_t`, `*mut *mut CqsHipIndex` <-> `cqs_hip_index**`, ...),
and for every `#[repr(C)]` struct with fields: field names, order and types against the header's struct, and its size and
field offsets against a tiny C program compiled here with gcc from the header itself.
The checker is itself checked: perturbing one argument (or one struct field) on either side must make it fail."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cqs_hip.h")
SHIM_DIR = os.path.join(ROOT, "rust_shim", "src")

# Rust opaque / struct names -> the header's
STRUCT_NAMES = {"CqsHipIndex": "cqs_hip_index", "CqsHipEmbedder": "cqs_hip_embedder", "CqsHipBert": "cqs_hip_bert",
                "CqsHipSparseIndex": "cqs_hip_sparse_index",
                "CqsHipEmbedConfig": "cqs_hip_embed_config", "CqsHipBertConfig": "cqs_hip_bert_config"}
RUST_SCALARS = {"i8": "int8_t", "u8": "uint8_t", "i16": "int16_t", "u16": "uint16_t", "i32": "int32_t", "u32": "uint32_t",
                "i64": "int64_t", "u64": "uint64_t", "f32": "float", "f64": "double", "usize": "size_t", "isize": "ptrdiff_t",
                "c_char": "char", "c_void": "void", "c_int": "int", "c_uint": "unsigned", "bool": "bool"}
C_SIZES = {"int8_t": 1, "uint8_t": 1, "int16_t": 2, "uint16_t": 2, "int32_t": 4, "uint32_t": 4, "int64_t": 8, "uint64_t": 8,
           "float": 4, "double": 8, "size_t": 8}


# ---- the header ------------------------------------------------------------------------------------------------------
def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def canon_c_type(t):
    """'const  float *' -> 'const float*'; parameter names are removed by the caller."""
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*\s*", "*", t)
    t = re.sub(r"\bstruct\s+", "", t)
    return t


def split_c_param(p):
    """'const float* rows' -> 'const float*';  'uint32_t' -> 'uint32_t'."""
    p = p.strip()
    m = re.match(r"^(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$", p)
    if m and m.group(2) not in C_SIZES and m.group(2) not in ("void", "char", "int", "float", "double", "unsigned"):
        # the last identifier is a parameter name unless the whole thing is a bare type
        if re.search(r"[A-Za-z0-9_]", m.group(1)):
            return canon_c_type(m.group(1))
    return canon_c_type(p)


def parse_header(text):
    text = strip_c_comments(text)
    funcs = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*\b(cqs_hip_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = canon_c_type(m.group(1)), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [split_c_param(p) for p in args.split(",")]
        funcs[name] = (ret, params)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(cqs_hip_[a-z0-9_]+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty, names = decl.split(None, 1)
            for nme in names.split(","):
                fields.append((nme.strip(), canon_c_type(ty)))
        structs[m.group(1)] = fields
    return funcs, structs


# ---- the shim --------------------------------------------------------------------------------------------------------
def strip_rust_comments(text):
    return re.sub(r"//[^\n]*", " ", text)


def rust_type_to_c(t):
    t = t.strip()
    m = re.match(r"^\*(const|mut)\s+(.*)$", t)
    if m:
        inner = rust_type_to_c(m.group(2))
        if m.group(1) == "const":
            # `*const T` -> `const T*`; for a pointee that is itself a pointer the const sits after it
            return ("const " + inner + "*") if "*" not in inner else (inner + " const*")
        return inner + "*"
    if t in RUST_SCALARS:
        return RUST_SCALARS[t]
    if t in STRUCT_NAMES:
        return STRUCT_NAMES[t]
    raise AssertionError("rust type not understood by the checker: %r" % t)


def split_top_level(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def parse_shim(text):
    text = strip_rust_comments(text)
    funcs = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"\bfn\s+(cqs_hip_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", blk.group(1), flags=re.S):
            name, args, ret = m.group(1), m.group(2), m.group(3)
            params = []
            for p in split_top_level(args):
                if not p.strip():
                    continue
                _pname, ty = p.split(":", 1)
                params.append(rust_type_to_c(ty))
            funcs[name] = (rust_type_to_c(ret) if ret else "void", params)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*(?:pub\s+)?struct\s+([A-Za-z0-9_]+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in split_top_level(m.group(2)):
            f = f.strip()
            if not f:
                continue
            nme, ty = f.split(":", 1)
            nme = nme.replace("pub", "").strip()
            if nme == "_private":
                continue                                    # opaque handle marker: [u8; 0]
            fields.append((nme, rust_type_to_c(ty)))
        structs[m.group(1)] = fields
    return funcs, structs


def norm(t):
    """`const char*` and `char const*` style differences do not occur in either file; only whitespace is normalised."""
    return re.sub(r"\s+", " ", t).strip()


def compare(header_text, shim_texts):
    """-> list of mismatch strings (empty = the shim matches the header)."""
    hf, hs = parse_header(header_text)
    problems = []
    seen = 0
    for fname, text in shim_texts.items():
        sf, ss = parse_shim(text)
        for name, (ret, params) in sf.items():
            seen += 1
            if name not in hf:
                problems.append("%s: %s is not declared in cqs_hip.h" % (fname, name))
                continue
            hret, hparams = hf[name]
            if norm(ret) != norm(hret):
                problems.append("%s: %s returns %s, header says %s" % (fname, name, ret, hret))
            if len(params) != len(hparams):
                problems.append("%s: %s takes %d arguments, header says %d" % (fname, name, len(params), len(hparams)))
                continue
            for i, (a, b) in enumerate(zip(params, hparams)):
                if norm(a) != norm(b):
                    problems.append("%s: %s argument %d is %s, header says %s" % (fname, name, i, a, b))
        for sname, fields in ss.items():
            if not fields:
                continue
            cname = STRUCT_NAMES.get(sname)
            if cname is None or cname not in hs:
                problems.append("%s: #[repr(C)] struct %s has no counterpart in cqs_hip.h" % (fname, sname))
                continue
            if fields != hs[cname]:
                problems.append("%s: struct %s fields %s, header %s has %s" % (fname, sname, fields, cname, hs[cname]))
    return problems, seen, hf, hs


def read_shim():
    return {f: open(os.path.join(SHIM_DIR, f)).read() for f in sorted(os.listdir(SHIM_DIR)) if f.endswith(".rs")}


def test_every_shim_declaration_matches_the_header():
    problems, seen, hf, _ = compare(open(HEADER).read(), read_shim())
    assert not problems, "\n".join(problems)
    assert seen >= 35, "the checker found only %d extern declarations in rust_shim/src - is it still parsing them?" % seen
    assert len(hf) >= 60, len(hf)


def test_the_checker_fails_when_either_side_is_perturbed():
    header = open(HEADER).read()
    shim = read_shim()
    assert compare(header, shim)[0] == []
    # one argument type in the shim
    bad = dict(shim)
    assert "keep_bitset: *const u32," in bad["hip.rs"]
    bad["hip.rs"] = bad["hip.rs"].replace("keep_bitset: *const u32,", "keep_bitset: *const u64,", 1)
    p = compare(header, bad)[0]
    assert any("cqs_hip_index_search argument 5" in x for x in p), p
    # one argument dropped from the header
    h2 = header.replace("uint32_t k, const uint32_t* keep_bitset, uint32_t mode, float threshold,\n                             uint64_t* out_rows",
                        "uint32_t k, const uint32_t* keep_bitset, float threshold,\n                             uint64_t* out_rows", 1)
    assert h2 != header
    p = compare(h2, shim)[0]
    assert any("cqs_hip_index_search takes 11 arguments, header says 10" in x for x in p), p
    # a return type
    bad = dict(shim)
    bad["hip.rs"] = bad["hip.rs"].replace("fn cqs_hip_index_len(idx: *const CqsHipIndex) -> u64;", "fn cqs_hip_index_len(idx: *const CqsHipIndex) -> u32;", 1)
    assert any("cqs_hip_index_len returns uint32_t" in x for x in compare(header, bad)[0])
    # const-ness of a pointer
    bad = dict(shim)
    bad["hip.rs"] = bad["hip.rs"].replace("fn cqs_hip_index_metric(idx: *const CqsHipIndex)", "fn cqs_hip_index_metric(idx: *mut CqsHipIndex)", 1)
    assert any("cqs_hip_index_metric argument 0" in x for x in compare(header, bad)[0])
    # two struct fields swapped in the shim
    bad = dict(shim)
    bad["hip_embed.rs"] = bad["hip_embed.rs"].replace("pub kv_heads: u32,\n    pub head_dim: u32,", "pub head_dim: u32,\n    pub kv_heads: u32,", 1)
    assert bad["hip_embed.rs"] != shim["hip_embed.rs"]
    assert any("struct CqsHipEmbedConfig" in x for x in compare(header, bad)[0])
    # a function the header does not have
    bad = dict(shim)
    bad["hip.rs"] = bad["hip.rs"].replace("fn cqs_hip_device_count() -> i32;", "fn cqs_hip_device_count() -> i32;\n    fn cqs_hip_made_up(x: u32) -> i32;", 1)
    assert any("cqs_hip_made_up is not declared" in x for x in compare(header, bad)[0])


def repr_c_layout(fields):
    """Size and offsets the Rust compiler gives a #[repr(C)] struct of these (C-typed) scalar fields."""
    off, offsets, align = 0, [], 1
    for _, ty in fields:
        sz = C_SIZES[ty]
        off = (off + sz - 1) // sz * sz
        offsets.append(off)
        off += sz
        align = max(align, sz)
    return (off + align - 1) // align * align, offsets


def test_repr_c_structs_have_the_header_layout(tmp_path):
    """sizeof / offsetof as gcc sees the header's structs == what #[repr(C)] gives the shim's."""
    _, hs = parse_header(open(HEADER).read())
    shim_structs = {}
    for text in read_shim().values():
        for sname, fields in parse_shim(text)[1].items():
            if fields:
                shim_structs[sname] = fields
    assert set(shim_structs) == {"CqsHipEmbedConfig", "CqsHipBertConfig"}, sorted(shim_structs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cqs_hip.h"', "int main(void) {"]
    for sname, fields in shim_structs.items():
        c = STRUCT_NAMES[sname]
        lines.append('  printf("%s %%zu", sizeof(%s));' % (c, c))
        for nme, _ in fields:
            lines.append('  printf(" %%zu", offsetof(%s, %s));' % (c, nme))
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        parts = ln.split()
        got[parts[0]] = [int(x) for x in parts[1:]]
    for sname, fields in shim_structs.items():
        size, offsets = repr_c_layout(fields)
        assert got[STRUCT_NAMES[sname]] == [size] + offsets, (sname, got[STRUCT_NAMES[sname]], size, offsets)
        assert hs[STRUCT_NAMES[sname]] == fields


if __name__ == "__main__":
    sys.exit(pytest.main([__file__, "-q"]))
