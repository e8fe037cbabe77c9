"""CPU-side checks of the boundary: the C-ABI library builds, loads and exports every symbol the
header declares; the drop-in module honours the reference's checkpoint contract; the pair index
(host integer code inside the library) is bit-exact with the reference's fixtures; the product
fails loudly without a GPU instead of falling back."""
import json
import os
import re

import numpy as np
import pytest
import torch

import rot_mvgaze_amd  # noqa: F401
from rot_mvgaze_amd import arch, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as ge
    ge.build()
    from rot_mvgaze_amd import _lib
    return _lib.lib()


def test_header_symbols_exported(built_lib):
    from rot_mvgaze_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rotmvgaze.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mvg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(built_lib, name), f"{name} declared in include/rotmvgaze.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert built_lib.mvg_abi_version() == _lib.ABI_VERSION == 10


def test_pair_index_bit_exact(built_lib, golden_dir):
    from rot_mvgaze_amd.pair_index import PairIndexRNG, build_pair_index
    with open(os.path.join(golden_dir, "pair_index.json")) as f:
        cases = json.load(f)
    n = 0
    for key, c in cases.items():
        if key == "shared_stream":
            rng = PairIndexRNG(c["seed"])
            assert [list(t) for t in build_pair_index(c["train_rows"], "novel_train", rng)] == c["train"]
            assert [list(t) for t in build_pair_index(c["test_rows"], "novel_test", rng)] == c["test"]
            continue
        got = build_pair_index(c["rows"], c["tag"], PairIndexRNG(c["seed"]))
        assert [list(t) for t in got] == c["tuples"], key
        n += len(got)
    assert n > 1000
    # edge cases: empty list, file shorter than a frame, single selected camera in a partial frame
    assert build_pair_index([], "all", PairIndexRNG(0)) == []
    assert build_pair_index([1], "all", PairIndexRNG(0)) == []
    assert build_pair_index([3], "novel_test", PairIndexRNG(0)) == []       # only camera 2 is selected
    # a larger run against the oracle restatement
    from oracle import restatement as R
    rows = [18 * 700 + 5, 18 * 123, 17]
    for tag in ("all", "novel_train", "novel_test"):
        assert build_pair_index(rows, tag, PairIndexRNG(42)) == R.build_pair_index(rows, tag, R.MT19937(42))


@pytest.mark.parametrize("depth", [18, 50])
def test_state_dict_contract(depth):
    from rot_mvgaze_amd.model import FeatRotationSymm
    m = FeatRotationSymm(backbone_depth=depth, num_iter=3)
    sd = m.state_dict()
    want = {n: tuple(s) for n, s, _ in arch.state_dict_shapes(depth, 3)}
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert len(sd) == (150 if depth == 18 else 348)
    nparam = sum(p.numel() for p in m.parameters())
    assert nparam == (40_019_502 if depth == 18 else 91_640_366)
    ref = synth.make_state_dict(depth, 3, 3, perturb_bn=True)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in ref.items()}, strict=True)
    back = m.state_dict()
    for k, v in ref.items():
        assert np.array_equal(back[k].numpy(), np.asarray(v)), k
    w = dict(m.named_parameters())["_feat_extractor.0.layer1.0.conv1.weight"]
    assert w.is_contiguous(memory_format=torch.channels_last)                # KRSC in memory


def test_invalid_variants_and_cpu_inputs_fail_loudly():
    from rot_mvgaze_amd.model import FeatRotationSymm
    with pytest.raises(AssertionError):                                   # rot_mv.py:133
        FeatRotationSymm(18, 3, encode_rotmat=True, ignore_rotmat=True)
    with pytest.raises(ValueError):                                       # combinations the reference cannot run either
        FeatRotationSymm(18, 3, share_feature=True, share_weights=True)
    m = FeatRotationSymm(18, 3)
    d = {"img_0": torch.zeros(1, 3, 32, 32), "img_1": torch.zeros(1, 3, 32, 32),
         "rot_0": torch.eye(3)[None], "rot_1": torch.eye(3)[None]}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(d)
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rotation_matrix_2d(torch.zeros(2, 2))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from rot_mvgaze_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_host_angular_error_metric(golden_dir):
    from rot_mvgaze_amd.geometry import angular_error
    g = np.load(os.path.join(golden_dir, "geometry_loss.npz"))
    sel = [0, 1] + list(range(3, 32))
    got = angular_error(g["loss_pred"].astype(np.float64), g["loss_gt"].astype(np.float64))
    np.testing.assert_allclose(got[sel], g["ang_err_np"][sel], rtol=1e-9, atol=1e-9)
