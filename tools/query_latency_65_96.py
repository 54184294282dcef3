import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tools.embed_two_streams_lib import make_engine
e, cfg = make_engine(0)
rng = np.random.default_rng(3)
for n in (64, 65, 72, 80, 81, 96):
    ids = rng.integers(1, 262144, size=(1, n)).astype(np.int64); mask = np.ones((1, n), np.int64)
    for _ in range(8): e.run(ids, mask)
    ms = []
    for _ in range(30):
        e.run(ids, mask); ms.append(e.last_ms())
    print("query of %3d tokens: device %.3f ms" % (n, float(np.median(ms))), flush=True)
