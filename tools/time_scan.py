"""Time idx.search_device for a query block without any result checks (kernel ablations).
usage: [CQS_HIP_LIB=...] python tools/time_scan.py ROWS BATCH [K]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cqs_amd import HipIndex
n, b = int(sys.argv[1]), int(sys.argv[2]); k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = torch.randn((n, 768), generator=g, device="cuda"); rows /= rows.norm(dim=1, keepdim=True)
q = torch.randn((b, 768), generator=g, device="cuda"); q /= q.norm(dim=1, keepdim=True)
idx = HipIndex.build_from_device(None, rows.data_ptr(), n, 768, borrow=True, keepalive=rows)
keys = torch.zeros((b, k), dtype=torch.int64, device="cuda"); cnt = torch.zeros((b,), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(): idx.search_device(q.data_ptr(), b, k, keys.data_ptr(), cnt.data_ptr(), stream=st)
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print("%-40s rows=%d b=%d ms/step=%.4f" % (os.path.basename(os.environ.get("CQS_HIP_LIB", "default")), n, b, e0.elapsed_time(e1) / 200))
