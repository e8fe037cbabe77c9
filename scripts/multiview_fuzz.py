#!/usr/bin/env python3
"""Random (depth, views, batch, image size) runs of tests/test_model_gpu.py::_multiview_case (one training step of the
V-view model against the CPU oracle): multiview_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_model_gpu as T
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    V, B = int(rng.integers(2, 8)), int(rng.choice([2, 3, 4, 5, 9, 16]))   # B = 1: 4-sample BatchNorm, covered by its own test
    depth = 50 if (rng.random() < 0.3 and V * B <= 24) else 18
    hw = int(rng.choice([64, 96, 128])) if V * B <= 32 else 64
    try:
        T._multiview_case(depth, V, B, hw, seed=int(rng.integers(0, 1000)))
        print("ok  ", depth, V, B, hw, flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", depth, V, B, hw, str(e)[:200], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
