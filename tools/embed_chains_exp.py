#!/usr/bin/env python3
"""Experiment: 2 chains (one engine) vs 4 chains (two engines) of 32 x 512 batches."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine
from tools.embed_split_exp import run  # noqa

if __name__ == "__main__":
    e1, _ = make_engine(0)
    print("2 chains (one engine) 32x512: %.0f" % run([e1], 32, 512, 24), flush=True)
    e2, _ = make_engine(1)
    print("4 chains (two engines) 32x512: %.0f" % run([e1, e2], 32, 512, 24), flush=True)
    print("2 chains (one engine) 32x512: %.0f" % run([e1], 32, 512, 24), flush=True)
