#!/usr/bin/env python3
"""Sparse (SPLADE) index search on one GPU: synthetic 1M-chunk corpus, timing of the blocking host API and of the
accumulate launch; optional oracle check + CPU timing."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=1_000_000)
    ap.add_argument("--vocab", type=int, default=30522)
    ap.add_argument("--terms", type=int, default=64)
    ap.add_argument("--queries", type=int, default=50)
    ap.add_argument("--k", type=int, default=500)
    ap.add_argument("--oracle", type=int, default=0)
    a = ap.parse_args()
    from cqs_amd.splade_index import HipSpladeIndex
    from cqs_amd import synth
    off, tok, w = synth.sparse_corpus(a.chunks, a.vocab)
    t0 = time.perf_counter()
    h = HipSpladeIndex.build_from_csr(None, off, tok, w)
    print("build %.2f s, postings %d, tokens %d" % (time.perf_counter() - t0, h.postings(), h.unique_tokens()))
    qs = synth.sparse_queries(a.queries, a.terms, a.vocab)
    for qt, qw in qs[:5]:
        h.search_raw(qt, qw, a.k)
    t0 = time.perf_counter()
    for qt, qw in qs:
        h.search_raw(qt, qw, a.k)
    wall = (time.perf_counter() - t0) / len(qs)
    acc, touched = [], []
    h.last_search()                                 # (asking for the launch's time makes later searches record two events)
    for qt, qw in qs:
        h.search_raw(qt, qw, a.k)
        ms, tp = h.last_search(); acc.append(ms); touched.append(tp)
    acc = np.array(acc); touched = np.array(touched, dtype=np.float64)
    bytes_ = touched * 8 + a.chunks * 4
    print("host api %.3f ms/query (%.0f q/s); accumulate %.3f ms mean (min %.3f); touched %.1f M postings mean; %.0f GB/s" %
          (wall * 1e3, 1 / wall, acc.mean(), acc.min(), touched.mean() / 1e6, (bytes_ / (acc * 1e-3)).mean() / 1e9))
    if a.oracle:
        from oracle import oracle as O
        o = O.SpladeIndex(off, tok, w)
        t0 = time.perf_counter()
        for qt, qw in qs[:a.oracle]:
            oc, os_ = o.search_raw(qt, qw, a.k)
        cpu = (time.perf_counter() - t0) / a.oracle
        hc, hs, _ = h.search_raw(qt, qw, a.k)
        print("oracle %.1f ms/query; last query identical: %s" % (cpu * 1e3, bool(np.array_equal(hc, oc) and np.array_equal(hs.view(np.uint32), os_.view(np.uint32)))))


if __name__ == "__main__":
    main()
