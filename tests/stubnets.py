"""Deterministic stand-in evaluators and weight generators shared by the golden-vector
generator, the oracle tests and the GPU parity tests.

Every stub is a pure function of the input feature planes, computed in exact integer
arithmetic and converted to float32 without rounding, so the reference search, the CPU
oracle and the device engine all see bit-identical "network outputs".
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(z: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _plane_hash(x: np.ndarray) -> np.ndarray:
    """uint64 hash per sample of a float32 [b, C, n, n] feature tensor."""
    b, c, n, _ = x.shape
    nbin = c - 2  # all planes but the two capture-count planes are 0/1
    bits = (x[:, :nbin].reshape(b, -1) > 0.5).astype(np.uint64)
    w = splitmix64(np.arange(1, bits.shape[1] + 1, dtype=np.uint64) * _GOLD)
    with np.errstate(over="ignore"):
        h = (bits * w[None, :]).sum(axis=1, dtype=np.uint64)
        caps_own = np.rint(x[:, c - 2, 0, 0].astype(np.float64) * (n * n)).astype(np.uint64)
        caps_opp = np.rint(x[:, c - 1, 0, 0].astype(np.float64) * (n * n)).astype(np.uint64)
        h = h + caps_own * np.uint64(0xD6E8FEB86659FD93) + caps_opp * np.uint64(0xA0761D6478BD642F)
    return h


def stub_eval(kind: str, x: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Evaluate a stub net. x: float32 [b, C, n, n] -> (logits f32 [b, A], value f32 [b])."""
    x = np.asarray(x, dtype=np.float32)
    b, _, n, _ = x.shape
    a = n * n + 1
    if kind == "uniform":
        return np.zeros((b, a), np.float32), np.zeros((b,), np.float32)
    scale = {"hash": 4.0, "sharp": 16.0, "flatish": 0.25, "const99": 4.0}[kind]
    h = _plane_hash(x)
    with np.errstate(over="ignore"):
        acts = (np.arange(1, a + 1, dtype=np.uint64) * _GOLD)[None, :]
        m = splitmix64(h[:, None] + acts)
        mv = splitmix64(h ^ np.uint64(0xABCDEF0123456789))
    logits = ((m >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5) * scale
    value = ((mv >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5) * 1.75
    if kind == "const99":  # forces the self-play resign branch
        value = np.full((b,), 0.99)
    return logits.astype(np.float32), value.astype(np.float32)


class TorchStub:
    """Callable with the reference model signature: x tensor -> (logits, value) tensors."""

    def __init__(self, kind: str):
        self.kind = kind
        self.calls = 0
        self.rows = 0

    def __call__(self, x):
        import torch

        xn = x.detach().cpu().numpy()
        self.calls += 1
        self.rows += xn.shape[0]
        lg, v = stub_eval(self.kind, xn)
        return torch.from_numpy(lg), torch.from_numpy(v)

    def eval(self):
        return self


# ---------------------------------------------------------------------------------------
# Seed-deterministic network weights: generated from shapes only, so the golden network
# outputs can be reproduced without committing any weight file.
# ---------------------------------------------------------------------------------------
def deterministic_state_dict(shapes: dict[str, tuple[int, ...]], seed: int) -> dict[str, np.ndarray]:
    """Fill a state_dict-shaped mapping with reproducible float32 values.

    Conv/linear weights ~ N(0, 2/fan_in); BN weight in [0.6, 1.4], running_mean in
    [-0.3, 0.3], running_var in [0.5, 1.5]; every bias in [-0.2, 0.2]. Keys are visited in
    sorted order so the stream does not depend on dict order.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: dict[str, np.ndarray] = {}
    for key in sorted(shapes):
        shp = tuple(shapes[key])
        if key.endswith("num_batches_tracked"):
            out[key] = np.array(7, dtype=np.int64)
        elif key.endswith("running_mean"):
            out[key] = rng.uniform(-0.3, 0.3, shp).astype(np.float32)
        elif key.endswith("running_var"):
            out[key] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        elif len(shp) == 1 and key.endswith("weight"):  # BN gamma
            out[key] = rng.uniform(0.6, 1.4, shp).astype(np.float32)
        elif len(shp) == 1:  # any bias
            out[key] = rng.uniform(-0.2, 0.2, shp).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:]))
            out[key] = (rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    return out


def realistic_state_dict(shapes: dict[str, tuple[int, ...]], seed: int, head_gain: float = 4.0) -> dict[str, np.ndarray]:
    """Reproducible float32 values at the scale a freshly initialised (or lightly trained) network has.

    Conv/linear weights and linear/conv biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (what torch.nn's default
    initialisers draw, model.py:9-87 uses them unchanged); BatchNorm weight in [0.8, 1.2], bias in [-0.1, 0.1],
    running_mean in [-0.2, 0.2], running_var in [0.6, 1.4]. The two output layers (policy_fc.weight, value_head.6.weight)
    are then scaled by `head_gain`, so that logits span a few units as a trained network's do instead of a few tenths.
    A bf16 evaluation of such a network is well conditioned. Keys are visited in sorted order.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: dict[str, np.ndarray] = {}
    fan = {k: int(np.prod(s[1:])) for k, s in shapes.items() if len(s) > 1}
    for key in sorted(shapes):
        shp = tuple(shapes[key])
        if key.endswith("num_batches_tracked"):
            out[key] = np.array(7, dtype=np.int64)
        elif key.endswith("running_mean"):
            out[key] = rng.uniform(-0.2, 0.2, shp).astype(np.float32)
        elif key.endswith("running_var"):
            out[key] = rng.uniform(0.6, 1.4, shp).astype(np.float32)
        elif len(shp) == 1 and key.endswith("weight"):  # BN gamma
            out[key] = rng.uniform(0.8, 1.2, shp).astype(np.float32)
        elif len(shp) == 1 and key[: -len("bias")] + "weight" in fan:   # bias of a conv / linear layer
            b = 1.0 / np.sqrt(fan[key[: -len("bias")] + "weight"])
            out[key] = rng.uniform(-b, b, shp).astype(np.float32)
        elif len(shp) == 1:  # BN beta
            out[key] = rng.uniform(-0.1, 0.1, shp).astype(np.float32)
        else:
            b = 1.0 / np.sqrt(fan[key])
            out[key] = rng.uniform(-b, b, shp).astype(np.float32)
    for key in ("policy_fc.weight", "value_head.6.weight"):
        if key in out:
            out[key] = out[key] * np.float32(head_gain)
    return out


def synthetic_planes(b: int, c: int, n: int, seed: int) -> np.ndarray:
    """Feature-plane-like float32 input: sparse 0/1 planes plus two constant fraction planes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = (rng.random((b, c, n, n)) < 0.3).astype(np.float32)
    x[:, c - 5] = (rng.random((b, 1, 1)) < 0.5).astype(np.float32)
    x[:, c - 2] = (rng.integers(0, 10, (b, 1, 1)) / float(n * n)).astype(np.float32)
    x[:, c - 1] = (rng.integers(0, 10, (b, 1, 1)) / float(n * n)).astype(np.float32)
    return x
