#!/usr/bin/env python3
"""DEVELOPMENT-CONTAINER ONLY (imports /root/reference like tests/golden/make_golden.py): the C++ pair index
(mvg_pair_index_build) and the oracle's against the reference's own GazeDataset.__init__ (fake in-memory
h5py) for random file sizes, seeds and camera tags - beyond the 25 committed fixtures.
pair_index_fuzz.py [cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG                      # sets up the reference imports with inert stubs
import numpy as np
import dataset.gaze as ref_gaze
from rot_mvgaze_amd.pair_index import PairIndexRNG, build_pair_index
from oracle import restatement as R

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


class FakeDS:
    def __init__(self, n):
        self.shape = (n, 224, 224, 3)


class FakeFile:
    rows = {}
    swmr_mode = True

    def __init__(self, path, mode="r", swmr=False):
        self.n = FakeFile.rows[os.path.basename(path)]

    def __getitem__(self, key):
        return FakeDS(self.n)

    def __bool__(self):
        return True

    def close(self):
        pass


sys.modules["h5py"].File = FakeFile
ref_gaze.h5py.File = FakeFile
bad = 0
for it in range(cases):
    rows = [int(rng.choice([0, 1, 17, 18, 19, 35, 36, 37, int(rng.integers(1, 700))])) for _ in range(int(rng.integers(1, 6)))]
    seed = int(rng.integers(0, 2 ** 63)) if rng.random() < 0.7 else int(rng.integers(0, 1000))
    FakeFile.rows = {f"f{i}.h5": n for i, n in enumerate(rows)}
    random.seed(seed)
    lib_rng, or_rng = PairIndexRNG(seed), R.MT19937(seed)
    ok = True
    for tag in [str(rng.choice(["all", "novel_train", "novel_test"])) for _ in range(int(rng.integers(1, 4)))]:
        ds = ref_gaze.GazeDataset("xgaze", "/fake", "rgb", None, keys_to_use=[f"f{i}.h5" for i in range(len(rows))],
                                  camera_tag=tag, stereo=True)
        ref = [tuple(map(int, t)) for t in ds.idx_to_kv]
        ok = ok and build_pair_index(rows, tag, lib_rng) == ref and [tuple(t) for t in R.build_pair_index(rows, tag, or_rng)] == ref
    bad += not ok
    print(("ok  " if ok else "FAIL"), rows, seed, flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
