#!/usr/bin/env python3
"""Debug aid: one GEMM kernel variant against torch on exact integer data; prints where the output differs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cqs_amd import _lib
tile = sys.argv[1] if len(sys.argv) > 1 else "p8:4"
os.environ["CQS_HIP_GEMM_TILE"] = tile
f = _lib.load().cqs_hip_debug_gemm_run
f.restype = C.c_int32
f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_uint32] * 4 + [C.c_int32, C.c_void_p]
g = torch.Generator(device="cuda"); g.manual_seed(7)
for M, N, K in [(256, 768, 64), (256, 768, 128), (256, 768, 192), (256, 768, 256), (256, 768, 320), (256, 768, 384), (256, 768, 768),
                (512, 1536, 768), (16384, 768, 768)]:
    A = torch.randint(-4, 5, (M, K), generator=g, device="cuda").to(torch.bfloat16)
    W = torch.randint(-3, 4, (N, K), generator=g, device="cuda").to(torch.bfloat16)
    ref = A.float() @ W.float().T
    bad_runs = 0
    for rep in range(5):
        out = torch.full((M, N), 7.0, device="cuda", dtype=torch.float32)
        assert f(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, N, 1, None) == 0
        torch.cuda.synchronize()
        bad = (out != ref)
        if bad.any():
            bad_runs += 1
            if bad_runs == 1:
                idx = bad.nonzero()
                rows = idx[:, 0].unique().cpu().tolist(); cols = idx[:, 1].unique().cpu().tolist()
                print(f"  {tile} M={M} N={N} K={K}: {int(bad.sum())} wrong of {M*N}; rows {rows[:8]}..{rows[-3:]} ({len(rows)}), cols {cols[:8]}..{cols[-3:]} ({len(cols)})")
                # which k-tiles explain the difference?  diff = sum over missing / extra k-tiles
                r, c = int(idx[0, 0]), int(idx[0, 1])
                parts = [(A[r, k:k+64].float() @ W[c, k:k+64].float()).item() for k in range(0, K, 64)]
                print(f"    first bad ({r},{c}): got {out[r,c].item()} want {ref[r,c].item()} diff {out[r,c].item()-ref[r,c].item()}; k-tile parts {parts}")
    print(f"{tile} M={M} N={N} K={K}: bad runs {bad_runs}/5", flush=True)
