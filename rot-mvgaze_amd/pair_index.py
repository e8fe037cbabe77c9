"""Stereo pair index of the dataset (integer-only, bit-exact with the reference).

Mirrors the ``idx_to_kv`` build of /root/reference/dataset/gaze.py:39-73 driven by Python's global
``random`` stream (seeded by utils/util.py:8).  Runs in the C library (host code: the stream is
sequential by construction) - O(rows) instead of the reference's O(rows^2) list scans.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from ._lib import check, lib

CAMERA_TAGS = {"all": 0, "novel_train": 1, "novel_test": 2}


class PairIndexRNG:
    """CPython-compatible MT19937 state (``random.seed(int)``) that persists across builds, like
    the reference's global ``random`` does between the train and test datasets (main.py:130-147)."""

    def __init__(self, seed: int = 0):
        self.state = np.zeros(625, dtype=np.uint32)
        if seed < 0 or seed >= 1 << 64:
            raise ValueError("seed must fit 64 bits")
        check(lib().mvg_mt19937_seed(self.state.ctypes.data_as(C.c_void_p), C.c_uint64(int(seed))), "mt19937_seed")


def build_pair_index(file_rows: Sequence[int], camera_tag: str, rng: PairIndexRNG) -> List[Tuple[int, int, int]]:
    """(file index, row, partner row) for every selected row that has at least one other selected
    camera in its 18-row frame."""
    rows = np.asarray(list(file_rows), dtype=np.int64)
    cap = int(rows.sum()) + 1
    out = np.empty((cap, 3), dtype=np.int64)
    n = lib().mvg_pair_index_build(rng.state.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p), len(rows),
                                   CAMERA_TAGS[camera_tag], out.ctypes.data_as(C.c_void_p), cap)
    if n < 0:
        raise RuntimeError("pair_index_build failed")
    return [tuple(int(x) for x in t) for t in out[:n]]
