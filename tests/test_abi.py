"""The C-ABI library loads and exports every symbol include/cqs_hip.h declares (CPU; no compute)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from cqs_amd import _lib
    return _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cqs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cqs_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built):
    lib = built.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cqs_hip.h but not exported"
    bound = {s[0] for s in built.SIGNATURES}
    assert set(declared) == bound, f"binding table and header disagree: {set(declared) ^ bound}"


def test_exports_are_c_abi(built):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH], text=True)
    syms = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for name in _declared_symbols():
        assert name in syms  # unmangled


def test_gfx950_code_object(built):
    """The shared library embeds a gfx950 code object (hipcc --offload-arch=gfx950)."""
    data = open(built.LIB_PATH, "rb").read()
    assert b"gfx950" in data


def test_no_device_is_reported_not_crashed(built):
    lib = built.load()
    n = lib.cqs_hip_device_count()
    assert n >= 0
    assert lib.cqs_hip_version().startswith(b"cqs-hip")
    if n == 0:
        import ctypes as C
        import numpy as np
        h = C.c_void_p()
        rows = np.zeros((4, 8), np.float32)
        rc = lib.cqs_hip_index_create(rows.ctypes.data, 4, 8, 0, 0, 0, C.byref(h))
        assert rc == built.ERR_NO_DEVICE and not h.value  # product path fails loudly, no CPU fallback


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under cqs_amd/ may import, link or call it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cqs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for line in text.splitlines():
                    s = line.strip()
                    if s.startswith(("import ", "from ", "#include")) or "dlopen" in s or "CDLL" in s or "-l" in s:
                        assert "oracle" not in s, f"{f}: product code references the oracle: {s}"


def test_host_merge_and_unpack(built):
    """cqs_hip_merge_keys / cqs_hip_unpack_keys are host-only helpers: exercised on CPU."""
    import numpy as np
    from cqs_amd import merge_keys, unpack_keys

    def pack(score, row):
        b = np.float32(score).view(np.uint32)
        ok = (~b) & np.uint32(0xFFFFFFFF) if b >> 31 else b ^ np.uint32(0x80000000)
        return (int(ok) << 32) | (0xFFFFFFFF - row)

    a = sorted([pack(0.9, 5), pack(0.5, 1), pack(-0.25, 7)], reverse=True)
    b = sorted([pack(0.9, 2), pack(0.7, 9)], reverse=True)
    lists = np.zeros((2, 4), np.uint64)
    lists[0, :3] = a
    lists[1, :2] = b
    out = merge_keys(lists, np.array([3, 2], np.uint32), 4)
    rows, scores = unpack_keys(out)
    assert list(rows) == [2, 5, 9, 1]  # 0.9 tie -> smaller row first
    assert np.allclose(scores, [0.9, 0.9, 0.7, 0.5])
    out = merge_keys(lists, np.array([3, 2], np.uint32), 10)
    assert len(out) == 5 and unpack_keys(out)[1][-1] == np.float32(-0.25)


_DEF = re.compile(r"^(?:int32_t|uint32_t|uint64_t|void|float|size_t|const char\*) (cqs_hip_[a-z_0-9]+)\(", re.M)


def test_every_entry_point_has_the_exception_barrier():
    """include/cqs_hip.h promises that no C++ exception crosses the boundary (reference convention: a backend never
    panics out of a call, src/cagra.rs:445-470).  Every exported definition under cqs_amd/csrc must be a
    function-try-block: `... ) CQS_ABI_TRY { body } CQS_ABI_CATCH*` (abi_guard.h)."""
    src_dir = os.path.join(ROOT, "cqs_amd", "csrc")
    guarded = set()
    for f in sorted(os.listdir(src_dir)):
        if not f.endswith((".hip", ".cpp")):
            continue
        text = open(os.path.join(src_dir, f)).read()
        for m in _DEF.finditer(text):
            close = text.index(")", m.end())
            depth, i = 1, m.end()
            while depth:                                  # the parameter list's closing parenthesis
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            tail = text[i:i + 40].lstrip()
            if tail.startswith(";"):
                continue                                  # a declaration
            assert tail.startswith("CQS_ABI_TRY {"), f"{f}: {m.group(1)} has no exception barrier"
            guarded.add(m.group(1))
        assert text.count("CQS_ABI_TRY") == sum(text.count(c) for c in ("CQS_ABI_CATCH(", "CQS_ABI_CATCH_NOHANDLE", "CQS_ABI_CATCH_VAL(", "CQS_ABI_CATCH_VOID")), f
    missing = set(_declared_symbols()) - guarded
    assert not missing, f"declared in the header but not found guarded in the sources: {missing}"


def test_length_error_inside_the_library_is_a_return_value_not_an_abort(built):
    """A std::vector sized by a caller-provided count throws std::length_error inside cqs_hip_merge_keys (host-only,
    runs without a GPU): through the barrier that is `0 keys merged`, not std::terminate.  In a child process, so that
    a regression shows up as a failed assertion instead of killing the test runner."""
    code = (
        "import ctypes as C, numpy as np, sys; sys.path.insert(0, %r)\n"
        "from cqs_amd import _lib; lib = _lib.load()\n"
        "lists = np.zeros(4, np.uint64); counts = np.zeros(4, np.uint32); out = np.zeros(4, np.uint64)\n"
        "n = lib.cqs_hip_merge_keys(lists.ctypes.data, counts.ctypes.data, C.c_size_t(1 << 62), C.c_size_t(1), C.c_size_t(4), out.ctypes.data)\n"
        "print('merged', n)\n" % ROOT)
    p = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "merged 0" in p.stdout, (p.returncode, p.stdout, p.stderr[-800:])


def test_exception_barrier_turns_bad_alloc_into_a_status(tmp_path):
    """The barrier macros around the untrusted-file readers with a throwing operator new (test hook in
    tests/abi_guard_driver.cpp), under ASAN + UBSan on the CPU: whichever allocation fails, the call returns a status."""
    import shutil
    import numpy as np
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    here = os.path.join(ROOT, "tests")
    exe = tmp_path / "abi_guard_driver"
    subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    os.path.join(here, "abi_guard_driver.cpp"), os.path.join(ROOT, "cqs_amd/csrc/onnx_reader.cpp"),
                    os.path.join(ROOT, "cqs_amd/csrc/safetensors_reader.cpp"), "-o", str(exe)], check=True, capture_output=True)
    import torch
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(3)
    st = tmp_path / "model.safetensors"
    save_file({"a.weight": torch.randn(8, 16, generator=g), "b.weight": torch.randn(4, 4, generator=g).to(torch.bfloat16)}, str(st))
    os.sys.path.insert(0, here)
    import onnx_bytes as ob
    rng = np.random.default_rng(4)
    onnx = tmp_path / "model.onnx"
    onnx.write_bytes(ob.model([ob.node("MatMul", "/model/layers.0/self_attn/q_proj/MatMul", ["h", "onnx::MatMul_1"], ["o"])],
                              [ob.tensor("onnx::MatMul_1", rng.standard_normal((64, 64)).astype(np.float32)),
                               ob.tensor("model.norm.weight", rng.standard_normal(64).astype(np.float32))]))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23:alloc_dealloc_mismatch=0", UBSAN_OPTIONS="halt_on_error=1:exitcode=24")
    seen = set()
    for kind, path in (("st", st), ("onnx", onnx)):
        for fail_after in [-1] + list(range(0, 40)):
            p = subprocess.run([str(exe), kind, str(path), str(fail_after)], capture_output=True, text=True, env=env, timeout=60)
            assert p.returncode == 0 and p.stdout.startswith("rc="), (kind, fail_after, p.returncode, p.stderr[-1500:])
            rc = int(p.stdout.split()[0][3:])
            seen.add(rc)
            if fail_after == -1:
                assert rc >= 1
            if fail_after == 0:
                assert rc == -3 and "bad_alloc" in p.stdout     # CQS_HIP_ERR_NOMEM
    assert -3 in seen and any(r >= 1 for r in seen)
