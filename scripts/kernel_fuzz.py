#!/usr/bin/env python3
"""Random-shape runs of the parametrised kernel tests (BatchNorm apply/backward, fused stem, pooling):
kernel_fuzz.py [cases] [seed]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_kernels_gpu as T

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    G, N = int(rng.integers(1, 4)), int(rng.integers(1, 7))
    H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    C = 4 * int(rng.choice([1, 2, 4, 8, 16, 32, 64, 128, 256, 512]))      # c/4 must divide 256 or be a multiple of it (checked by the library)
    relu, res = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if N * H * W < 2:
        continue
    for name, fn, args in (("bn", T.test_bn_apply_and_backward, (G, N, H, W, C, relu, res)),
                           ("stem", T.test_fused_stem_bn_relu_maxpool, (G, N, max(H, 3), max(W, 3)))):
        try:
            fn(*args)
            print("ok  ", name, args, flush=True)
        except Exception as e:
            bad += 1
            print("FAIL", name, args, str(e)[:200], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
