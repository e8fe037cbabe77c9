"""MI355X-native Rot-MVGaze hot path: ResNet backbone + rotation-constrained cross-view fusion +
gaze heads + angular/stereo/iteration loss, forward and backward, as hand-written HIP kernels
(gfx950) behind the reference's ``FeatRotationSymm(...).forward(dict) -> dict`` surface.

Importing the package does not load the HIP library; every compute entry point does, and raises
``RuntimeError`` if ``csrc/librotmvgaze_hip.so`` is missing (there is no CPU fallback).
"""
from . import arch, synth  # noqa: F401  (pure-python, CPU-safe)

__all__ = ["arch", "synth"]


def __getattr__(name):
    # lazy: these pull in torch + the ctypes binding
    import importlib
    lazy = {
        "FeatRotationSymm": "model", "MultiViewGaze": "model",
        "GazeLoss": "losses", "StereoL1Loss": "losses", "IterationLoss": "losses",
        "MultiViewIterationLoss": "losses",
        "rotation_matrix_2d": "geometry", "pitchyaw_to_vector": "geometry",
        "angular_error": "geometry",
        "build_pair_index": "pair_index",
        "GradAllReducer": "dp",
        "Adam": "optim",
    }
    if name in lazy:
        return getattr(importlib.import_module("." + lazy[name], __name__), name)
    if name in ("model", "losses", "geometry", "pair_index", "dp", "ops", "backbone", "heads", "_lib", "optim"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
