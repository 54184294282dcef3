"""Diagnostic probe (not a test): prints per-wave scan stamps and select phase times.
Run on the GPU box: python tools/stamp_probe.py"""


def main():
    import os, sys
    os.environ["CQS_HIP_DEBUG_STAMPS"]="1"
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
    import numpy as np, torch
    from cqs_amd import HipIndex
    n=int(os.environ.get("ROWS","1000000"))
    g=torch.Generator(device="cuda"); g.manual_seed(1)
    rows=torch.randn((n,768),generator=g,device="cuda"); rows/=rows.norm(dim=1,keepdim=True)
    idx=HipIndex.build_from_device(None, rows.data_ptr(), n, 768, borrow=True, keepalive=rows)
    q=torch.randn((768,),generator=g,device="cuda"); q/=q.norm(); q=q.cpu().numpy()
    for k in (20,20,500,500,1024):
        print("k",k, file=sys.stderr); idx.search_batch(q,k)


if __name__ == "__main__":
    main()
