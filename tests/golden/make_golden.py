#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE's own Python.

Runs only in the build container (needs /root/reference, read-only).  It imports the reference
modules unmodified (inert stubs for third-party imports that the hot path never calls, SURVEY.md
§8(c)), loads the deterministic weights of ``rot_mvgaze_amd.synth`` with ``load_state_dict``,
runs forward / loss / backward on the deterministic inputs and stores *outputs only* (inputs and
weights are re-derived from (depth, seed) by the tests).  No reference source is copied.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np
import torch

# ---- inert stubs for imports the hot path never uses (rot_mv.py:8, math.py:5-15, gaze.py) ----
for name in ["torchvision", "torchvision.models", "torchvision.transforms", "cv2", "h5py",
             "albumentations", "omegaconf", "rich", "rich.progress"]:
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
for attr in ("OmegaConf", "ListConfig", "DictConfig"):
    setattr(sys.modules["omegaconf"], attr, type(attr, (), {}))
sys.modules["rich.progress"].track = lambda it, **kw: it
sys.modules["rich"].progress = sys.modules["rich.progress"]


def _no_download(*a, **k):
    raise RuntimeError("pretrained download is forbidden offline (SURVEY.md §8(c))")


import torch.hub
import torch.utils.model_zoo
torch.hub.load_state_dict_from_url = _no_download
torch.utils.model_zoo.load_url = _no_download

import models.resnet as ref_resnet            # noqa: E402
ref_resnet.load_state_dict_from_url = _no_download
import models.rot_mv as ref_rot_mv            # noqa: E402
ref_rot_mv.load_state_dict_from_url = _no_download
_r18, _r50 = ref_resnet.resnet18, ref_resnet.resnet50
ref_rot_mv.resnet18 = lambda pretrained=False, **kw: _r18(pretrained=False, **kw)
ref_rot_mv.resnet50 = lambda pretrained=False, **kw: _r50(pretrained=False, **kw)
import utils.math as ref_math                 # noqa: E402
import losses.gaze_loss as ref_gaze_loss      # noqa: E402
import losses.stereo_loss as ref_stereo_loss  # noqa: E402

import rot_mvgaze_amd                         # noqa: E402
from rot_mvgaze_amd import synth              # noqa: E402

torch.set_num_threads(8)


def t2n(x):
    return x.detach().cpu().numpy()


def gen_geometry():
    """A1 / A8 known answers + random batches."""
    hp = torch.tensor(synth.uniform01(64, 7, "geom_hp").reshape(32, 2) - 0.5, dtype=torch.float32) * 2.0
    hp[0] = torch.tensor([0.1, 0.2])
    out = {"hp": t2n(hp),
           "R": t2n(ref_math.rotation_matrix_2d(hp)),
           "R_inv": t2n(ref_math.rotation_matrix_2d(hp, inverse=True)),
           "R_1d": t2n(ref_math.rotation_matrix_2d(hp[0])),
           "vec": t2n(ref_math.pitchyaw_to_vector(hp))}
    pred = torch.tensor(synth.uniform01(64, 8, "loss_pred").reshape(32, 2) - 0.5, dtype=torch.float32)
    gt = torch.tensor(synth.uniform01(64, 8, "loss_gt").reshape(32, 2) - 0.5, dtype=torch.float32)
    pred[0] = torch.tensor([0.1, 0.2]); gt[0] = torch.tensor([0.1, 0.25])
    pred[1] = torch.tensor([0.0, 0.0]); gt[1] = torch.tensor([0.3, -0.2])
    pred[2] = gt[2]                                     # sits ON the acos singularity
    pred = pred.requires_grad_(True)
    loss = ref_gaze_loss.gaze_angular_loss(pred, gt)
    loss.backward()
    out.update({"loss_pred": t2n(pred), "loss_gt": t2n(gt), "loss": t2n(loss), "loss_dpred": t2n(pred.grad)})
    p2 = torch.tensor([[0.1, 0.2], [0.0, 0.0]], requires_grad=True)
    g2 = torch.tensor([[0.1, 0.25], [0.3, -0.2]])
    l2 = ref_gaze_loss.gaze_angular_loss(p2, g2)
    l2.backward()
    out.update({"ka_loss": t2n(l2), "ka_dpred": t2n(p2.grad)})
    out["ang_err_np"] = ref_math.angular_error_numpy(t2n(pred).astype(np.float64), t2n(gt).astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "geometry_loss.npz"), **out)
    print("geometry_loss: loss", float(loss), "known-answer", float(l2))


def gen_lp_loss():
    """GazeLoss loss_type 'l1' / 'l2' (losses/gaze_loss.py:21-29,56-64): value and gradient wrt pred."""
    pred = torch.tensor(synth.uniform01(64, 18, "lp_pred").reshape(32, 2) - 0.5, dtype=torch.float32)
    label = torch.tensor(synth.uniform01(64, 18, "lp_label").reshape(32, 2) - 0.5, dtype=torch.float32)
    pred[3] = label[3]                                  # |x| at 0: zero gradient
    out = {"pred": t2n(pred), "label": t2n(label)}
    for lt in ("l1", "l2"):
        p = pred.clone().requires_grad_(True)
        loss = ref_gaze_loss.GazeLoss(gaze_weight=1.0, loss_type=lt)(p, label)
        loss.backward()
        out[f"{lt}_loss"] = t2n(loss)
        out[f"{lt}_dpred"] = t2n(p.grad)
    np.savez_compressed(os.path.join(HERE, "lp_loss.npz"), **out)
    print("lp_loss: l1", float(out["l1_loss"]), "l2", float(out["l2_loss"]))


def gen_vec2py():
    """utils/math.py:62-94 (imported by trainer.py:26 and losses/gaze_loss.py:6, not called on the path):
    vector_to_pitchyaw on un-normalised vectors, torch and numpy branches."""
    v = torch.tensor(synth.normal(96, 9, "vec2py").reshape(32, 3), dtype=torch.float32)
    v[0] = torch.tensor([0.0, 0.0, 2.0])
    out = {"v": t2n(v), "py_torch": t2n(ref_math.vector_to_pitchyaw(v)),
           "py_numpy": ref_math.vector_to_pitchyaw(t2n(v).astype(np.float64))}
    np.savez_compressed(os.path.join(HERE, "vector_to_pitchyaw.npz"), **out)
    print("vector_to_pitchyaw: 32 vectors")


def gen_pair_index():
    """A10: run the reference GazeDataset.__init__ against a fake in-memory h5py."""
    import dataset.gaze as ref_gaze

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
    cases = {"a": [36, 40], "b": [18, 17, 19, 1, 54], "c": [5], "d": [180, 7]}
    res = {}
    for cname, rows in cases.items():
        FakeFile.rows = {f"f{i}.h5": n for i, n in enumerate(rows)}
        for seed in (0, 123456789012):
            for tag in ("all", "novel_train", "novel_test"):
                random.seed(seed)
                ds = ref_gaze.GazeDataset("xgaze", "/fake", "rgb", None,
                                          keys_to_use=[f"f{i}.h5" for i in range(len(rows))],
                                          camera_tag=tag, stereo=True)
                res[f"{cname}|{seed}|{tag}"] = {"rows": rows, "seed": seed, "tag": tag,
                                                "tuples": [list(map(int, t)) for t in ds.idx_to_kv]}
    # one shared stream: train (novel_train) then test (novel_test), like main.py:130-147
    FakeFile.rows = {"f0.h5": 36, "f1.h5": 40}
    random.seed(5)
    tr = ref_gaze.GazeDataset("xgaze", "/fake", "rgb", None, keys_to_use=["f0.h5", "f1.h5"],
                              camera_tag="novel_train", stereo=True)
    te = ref_gaze.GazeDataset("xgaze", "/fake", "rgb", None, keys_to_use=["f1.h5"],
                              camera_tag="novel_test", stereo=True)
    res["shared_stream"] = {"seed": 5, "train_rows": [36, 40], "test_rows": [40],
                            "train": [list(map(int, t)) for t in tr.idx_to_kv],
                            "test": [list(map(int, t)) for t in te.idx_to_kv]}
    with open(os.path.join(HERE, "pair_index.json"), "w") as f:
        json.dump(res, f, separators=(",", ":"))
    print("pair_index:", len(res), "cases; a|0|all first", res["a|0|all"]["tuples"][:3])


GRAD_SAMPLES = [  # (state_dict key, number of leading flat elements kept)
    ("_feat_extractor.0.conv1.weight", 2048),
    ("_feat_extractor.0.bn1.weight", 64), ("_feat_extractor.0.bn1.bias", 64),
    ("_feat_extractor.0.layer1.0.conv1.weight", 2048),
    ("_feat_extractor.0.layer2.0.downsample.0.weight", 2048),
    ("_feat_extractor.0.layer2.0.downsample.1.weight", 128),
    ("_feat_extractor.0.layer3.1.conv2.weight", 2048),
    ("_feat_extractor.0.layer4.1.conv1.weight", 2048),
    ("_feat_extractor.0.layer4.1.bn2.weight", 512), ("_feat_extractor.0.layer4.1.bn2.bias", 512),
    ("_lifter._lifter.blocks.0.0.weight", 2048), ("_lifter._lifter.blocks.1.0.bias", 1536),
    ("_img_fusers.0._fuser.blocks.0.0.weight", 4096), ("_img_fusers.2._fuser.blocks.1.0.weight", 4096),
    ("_img_fusers.1._fuser.blocks.0.0.bias", 2048),
    ("_gaze_estimators.0.blocks.0.0.weight", 4096), ("_gaze_estimators.2.blocks.1.0.weight", 1024),
    ("_gaze_estimators.2.blocks.1.0.bias", 2),
]
STAT_SAMPLES = ["_feat_extractor.0.bn1", "_feat_extractor.0.layer1.0.bn2", "_feat_extractor.0.layer2.0.downsample.1",
                "_feat_extractor.0.layer4.1.bn1"]


def build_ref(depth, seed, perturb_bn=True, variant=None):
    from rot_mvgaze_amd.arch import DEFAULT_VARIANT
    v = variant or DEFAULT_VARIANT
    model = ref_rot_mv.FeatRotationSymm(backbone_depth=depth, num_iter=3, share_weights=v.share_weights,
                                        encode_rotmat=v.encode_rotmat, share_feature=v.share_feature,
                                        ignore_rotmat=v.ignore_rotmat)
    sd_np = synth.make_state_dict(depth, seed, 3, perturb_bn=perturb_bn, variant=v)
    ref_keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    my_keys = {k: tuple(v.shape) for k, v in sd_np.items()}
    assert ref_keys == my_keys, "state_dict contract mismatch"
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    return model


def ref_inputs(batch, seed, hw):
    inp = synth.make_inputs(batch, 2, seed, hw)
    img = torch.from_numpy(inp["img"])
    hp = torch.from_numpy(inp["head_pose"])
    gt = torch.from_numpy(inp["gt_gaze"])
    data = {"img_0": img[:, 0].contiguous(), "img_1": img[:, 1].contiguous(),
            "rot_0": ref_math.rotation_matrix_2d(hp[:, 0]), "rot_1": ref_math.rotation_matrix_2d(hp[:, 1]),
            "gt_gaze": gt[:, 0].contiguous(), "gt_gaze_1": gt[:, 1].contiguous()}
    return data


def collect_outputs(data, out):
    for k in ("img_feat_0", "img_feat_1", "initial_rot_feat_0", "initial_rot_feat_1", "pred_gaze"):
        out[k] = t2n(data[k])
    for i in range(3):
        for k in ("feat_0", "feat_1", "pred_gaze_0", "pred_gaze_1"):
            out[f"iter_{i}.{k}"] = t2n(data[f"iter_{i}"][k])


VARIANT_GRAD_SAMPLES = ["_img_fusers.0._fuser.blocks.0.0.weight", "_img_fusers.0._fuser.blocks.0.0.bias",
                        "_img_fusers.2._fuser.blocks.1.0.weight", "_gaze_estimators.1.blocks.0.0.weight",
                        "_lifter._lifter.blocks.1.0.weight", "_feat_extractor.0.layer4.1.conv2.weight",
                        "_feat_extractor.0.conv1.weight"]


def gen_variant(name, variant, depth=18, batch=3, hw=64, seed_w=0, seed_in=1234):
    """Ablation variants (rot_mv.py:136-171): eval outputs, one train step (outputs, loss, gradient
    samples + norms, IntensityBatchNorm buffers)."""
    tag = f"variant_{name}_r{depth}_b{batch}_hw{hw}"
    metrics = ref_stereo_loss.IterationLoss(
        loss=ref_stereo_loss.StereoL1Loss(rel_weight=0.01, reference_decay=1.0,
                                          distance_metric="angular_error", pred_gaze_key="pred_gaze"),
        iter_decay=0.5)
    out = {}
    model = build_ref(depth, seed_w, True, variant)
    model.eval()
    with torch.no_grad():
        data = model(ref_inputs(batch, seed_in, hw))
    ev = {}
    collect_outputs(data, ev)
    out.update({"eval." + k: v for k, v in ev.items()})
    model = build_ref(depth, seed_w, True, variant)
    model.train()
    data = model(ref_inputs(batch, seed_in, hw))
    loss = metrics(data)
    loss.backward()
    tr = {}
    collect_outputs(data, tr)
    out.update({"train." + k: v for k, v in tr.items()})
    out["train.loss"] = t2n(loss)
    params = dict(model.named_parameters(remove_duplicate=False))
    for key in VARIANT_GRAD_SAMPLES:
        g = params[key].grad
        out["grad." + key] = t2n(g).reshape(-1)[:4096].copy()
        out["gradnorm." + key] = np.array(float(g.double().norm()))
    for k, v in model.state_dict().items():
        if k.endswith("_batchnorm.running_mean"):
            out["stat." + k] = t2n(v)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, "loss", float(loss), "pred_gaze[0]", out["train.pred_gaze"][0])


def gen_model(depth, batch, hw, seed_w=0, seed_in=1234):
    tag = f"model_r{depth}_b{batch}_hw{hw}"
    metrics = ref_stereo_loss.IterationLoss(
        loss=ref_stereo_loss.StereoL1Loss(rel_weight=0.01, reference_decay=1.0,
                                          distance_metric="angular_error", pred_gaze_key="pred_gaze"),
        iter_decay=0.5)
    out = {}
    # ---- eval forward ----
    model = build_ref(depth, seed_w)
    model.eval()
    with torch.no_grad():
        data = model(ref_inputs(batch, seed_in, hw))
    ev = {}
    collect_outputs(data, ev)
    out.update({"eval." + k: v for k, v in ev.items()})
    # ---- train step: forward + loss + backward (+ BN running stats) ----
    model = build_ref(depth, seed_w)
    model.train()
    data = ref_inputs(batch, seed_in, hw)
    data["img_0"].requires_grad_(True)
    data["img_1"].requires_grad_(True)
    data = model(data)
    loss = metrics(data)
    loss.backward()
    tr = {}
    collect_outputs(data, tr)
    out.update({"train." + k: v for k, v in tr.items()})
    out["train.loss"] = t2n(loss)
    params = dict(model.named_parameters())
    for key, n in GRAD_SAMPLES:
        g = params[key].grad
        out["grad." + key] = t2n(g).reshape(-1)[:n].copy()
        out["gradnorm." + key] = np.array(float(g.double().norm()))
    assert params["_feat_extractor.0.fc.weight"].grad is None
    out["grad.img_0"] = t2n(data["img_0"].grad)[:, :, ::16, ::16].copy()
    out["grad.img_1"] = t2n(data["img_1"].grad)[:, :, ::16, ::16].copy()
    out["gradnorm.img_0"] = np.array(float(data["img_0"].grad.double().norm()))
    sd = model.state_dict()
    for p in STAT_SAMPLES:
        out["stat." + p + ".running_mean"] = t2n(sd[p + ".running_mean"])
        out["stat." + p + ".running_var"] = t2n(sd[p + ".running_var"])
        out["stat." + p + ".num_batches_tracked"] = t2n(sd[p + ".num_batches_tracked"])
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, "loss", float(loss), "pred_gaze[0]", out["train.pred_gaze"][0])


def gen_multi_erase():
    """RandomMultiErasing of the reference (utils/augment.py:10-47), seeded: which images are erased and
    the resulting images (input = seeded noise), for the RNG-order and kernel parity tests."""
    import utils.augment as ref_aug
    random.seed(7)
    np.random.seed(7)
    torch.manual_seed(7)
    aug = ref_aug.RandomMultiErasing(p=0.5, proportion=[0.5, 0.6], dot_size=[0.05, 0.3])     # main.py:48
    imgs = torch.from_numpy(synth.normal(10 * 3 * 40 * 56, 77, "erase").reshape(10, 3, 40, 56).astype(np.float32))
    out = torch.stack([aug(imgs[i].clone()) for i in range(10)])
    np.savez_compressed(os.path.join(HERE, "multi_erase.npz"), out=out.numpy())
    print("multi_erase: erased images", int((out != imgs).flatten(1).any(1).sum()), "of 10")


RESIZE_CASES = [(2, 37, 41, 24), (1, 100, 90, 64), (1, 50, 60, 96), (1, 96, 96, 48), (1, 64, 64, 64)]   # n, h, w, size


def gen_resize():
    """test_transform of main.py:50-55 on raw uint8 patches of other sizes than 224: ToTensor ->
    Resize((S, S), antialias=True) -> Normalize.  torchvision is not installed in this image; its
    tensor Resize is one call to torch.nn.functional.interpolate(mode='bilinear', antialias=True,
    align_corners=False) (torchvision/transforms/_functional_tensor.py: resize), which is what
    generates these vectors (small output sizes keep the fixture small; the op is size-generic)."""
    import torch.nn.functional as F
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)      # main.py:38-39
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
    out = {}
    for idx, (n, h, w, size) in enumerate(RESIZE_CASES):
        rng = np.random.default_rng(100 + idx)
        u8 = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
        x = torch.from_numpy(u8).permute(0, 3, 1, 2).to(torch.float32).div(255)       # ToTensor
        if (h, w) != (size, size):
            x = F.interpolate(x, size=(size, size), mode="bilinear", antialias=True, align_corners=False)
        y = (x - torch.from_numpy(mean)) / torch.from_numpy(std)                        # Normalize
        out[f"u8_{idx}"] = u8
        out[f"y_{idx}"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "resize_aa.npz"), **out)
    print("resize_aa:", len(RESIZE_CASES), "cases")


def gen_variants():
    from rot_mvgaze_amd.arch import Variant
    gen_variant("share_weights", Variant(share_weights=True))
    gen_variant("ignore_rotmat", Variant(ignore_rotmat=True))
    gen_variant("encode_rotmat", Variant(encode_rotmat=True))
    gen_variant("share_feature", Variant(share_feature=True))
    gen_variant("share_weights_encode_rotmat", Variant(share_weights=True, encode_rotmat=True))


if __name__ == "__main__":
    if "--variants" in sys.argv:
        gen_variants()
        sys.exit(0)
    if "--erase" in sys.argv:
        gen_multi_erase()
        sys.exit(0)
    if "--lp" in sys.argv:
        gen_lp_loss()
        sys.exit(0)
    if "--vec2py" in sys.argv:
        gen_vec2py()
        sys.exit(0)
    if "--resize" in sys.argv:
        gen_resize()
        sys.exit(0)
    gen_geometry()
    gen_pair_index()
    gen_model(18, 2, 224)
    gen_model(50, 2, 224)
    gen_model(18, 3, 64)
    gen_model(50, 3, 64)
    gen_variants()
    gen_multi_erase()
    gen_resize()
    gen_vec2py()
    gen_lp_loss()
