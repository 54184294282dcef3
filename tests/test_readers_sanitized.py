"""The two readers of untrusted model files (cqs_amd/csrc/onnx_reader.cpp, safetensors_reader.cpp) under
AddressSanitizer + UBSan on the CPU: valid files parse, and truncated / bit-flipped files end in a clean error (or a
clean parse) - never in an out-of-bounds access.  (They run inside `load_dir` of both engines on whatever a model
directory holds; GPU sanitizers are not available on the pool, and these two files have no device code.)"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("rd") / "reader_driver"
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           os.path.join(HERE, "reader_driver.cpp"), os.path.join(ROOT, "cqs_amd/csrc/onnx_reader.cpp"),
           os.path.join(ROOT, "cqs_amd/csrc/safetensors_reader.cpp"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    return str(out)


def run(driver, kind, path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="halt_on_error=1:exitcode=24")
    p = subprocess.run([driver, kind, str(path)], capture_output=True, text=True, env=env, timeout=60)
    assert p.returncode == 0, (kind, path, p.returncode, p.stderr[-2000:])
    return p.stdout.strip()


def _onnx_file(d):
    import onnx_bytes as ob
    rng = np.random.default_rng(1)
    side = bytearray(b"\0" * 16)
    big = rng.standard_normal((64, 128)).astype(np.float32)
    raw = ob.encode_values(big, ob.FLOAT)
    off = len(side); side.extend(raw)
    nodes = [ob.node("MatMul", "/model/layers.0/self_attn/q_proj/MatMul", ["h", "onnx::MatMul_1"], ["o"])]
    inits = [ob.tensor("onnx::MatMul_1", rng.standard_normal((64, 64)).astype(np.float32)),
             ob.tensor("model.norm.weight", rng.standard_normal(64).astype(np.float32), ob.FLOAT, "float_data"),
             ob.tensor("model.layers.0.k_norm.weight", rng.standard_normal(64).astype(np.float32), ob.BFLOAT16),
             ob.tensor("half", rng.standard_normal((8, 8)).astype(np.float32), ob.FLOAT16),
             ob.tensor("big", big, ob.FLOAT, "external", external=("model.onnx_data", off, len(raw))),
             ob.tensor("shape", np.array([1, -1, 64]), ob.INT64)]
    (d / "model.onnx").write_bytes(ob.model(nodes, inits))
    (d / "model.onnx_data").write_bytes(bytes(side))
    return d / "model.onnx"


def _st_file(d):
    import torch
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(2)
    save_file({"a.weight": torch.randn(33, 17, generator=g), "b": torch.randn(64, generator=g).to(torch.bfloat16),
               "c": torch.randn(5, 5, generator=g).to(torch.float16), "ids": torch.arange(12)[None]},
              str(d / "model.safetensors"), metadata={"format": "pt"})
    return d / "model.safetensors"


@pytest.mark.parametrize("kind", ["onnx", "st"])
def test_valid_then_damaged_files(driver, tmp_path, kind):
    good = _onnx_file(tmp_path) if kind == "onnx" else _st_file(tmp_path)
    out = run(driver, kind, good)
    assert out.startswith("ok:") and " 0 tensors" not in out, out
    data = good.read_bytes()
    rng = np.random.default_rng(3)
    bad = tmp_path / ("damaged.onnx" if kind == "onnx" else "damaged.safetensors")
    if kind == "onnx":
        shutil.copy(tmp_path / "model.onnx_data", tmp_path / "damaged.onnx_data")     # (not referenced by name: external data goes missing)
    n_err = 0
    cases = [data[:n] for n in sorted(set(int(x) for x in np.linspace(0, len(data) - 1, 40)))]
    for _ in range(160):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, min(len(b), 4096)))            # headers live at the front
            b[i] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    for _ in range(120):                                              # anywhere in the file (tensor headers are interspersed)
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    for blob in cases:
        bad.write_bytes(blob)
        out = run(driver, kind, bad)                                  # returncode 0 = no sanitizer report
        assert out.startswith(("ok:", "error:")), out
        n_err += out.startswith("error:")
    assert n_err > 20                                                 # the damage is actually noticed
    if kind == "onnx":                                                # a sidecar shorter than the recorded offsets
        (tmp_path / "model.onnx_data").write_bytes(b"\0" * 64)
        assert run(driver, kind, good).startswith(("error:", "ok:"))
