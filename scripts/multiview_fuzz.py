#!/usr/bin/env python3
"""Random (views, batch) runs of tests/test_model_gpu.py::test_multiview_against_oracle: multiview_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_model_gpu as T
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    V, B = int(rng.integers(2, 8)), int(rng.choice([2, 3, 4, 5, 9, 16, 31]))   # B = 1: 4-sample BatchNorm, covered by its own test
    try:
        T.test_multiview_against_oracle.__wrapped__(V, B) if hasattr(T.test_multiview_against_oracle, "__wrapped__") else T.test_multiview_against_oracle(V, B)
        print("ok  ", V, B, flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", V, B, str(e)[:200], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
