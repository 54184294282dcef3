import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cqs_amd import _lib
lib = _lib.load()
f = lib.cqs_hip_debug_gemm_ms
f.restype = C.c_float
f.argtypes = [C.c_uint32] * 4 + [C.c_int32]
for N in (768,):
    for tiles_m in (40, 42, 43, 64, 84, 85, 86, 100, 128, 170, 171, 172, 256):
        M = tiles_m * 128
        ms = f(M, N, 768, 30, 0)
        print(f"M={M:6d} ({tiles_m:3d} x {N//128} = {tiles_m*N//128:5d} tiles) N={N}: {ms*1e3:7.1f} us {2.0*M*N*768/ms/1e9:7.1f} TF")
