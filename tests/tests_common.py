"""Builders shared by CPU and GPU network tests."""
import json

import numpy as np
import torch

import golden_io
import stubnets

GOLDEN_NETS = ["n5_c32_b1", "n9_c64_b4", "n9_c64_b2_nose", "n9_c192_b12", "n19_c32_b2_h4", "n19_c128_b3", "n13_c64_b2"]


def hash_name(s: str) -> int:
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def build_golden_net(name):
    """The net with the seed-generated weights used for tests/golden/network.npz, its input, its kwargs."""
    from alphazero_amd.model import AlphaZeroNet
    z = golden_io.load_network(name)
    kw = json.loads(bytes(z[f"{name}/cfg_json"]).decode())
    net = AlphaZeroNet(**kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    gen = stubnets.deterministic_state_dict(shapes, seed=hash_name(name))
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gen.items()})
    net.eval()
    c = 2 * kw.get("history_len", 8) + 5
    x = stubnets.synthetic_planes(6, c, kw["board_size"], seed=hash_name(name) + 1)
    return net, x, kw


# Weights at the scale of a freshly initialised network (stubnets.realistic_state_dict); outputs of the upstream module in
# tests/golden/network_realistic.npz (make_golden_network_realistic.py)
REALISTIC_NETS = ["real_n9_c192_b12", "real_n9_c64_b4", "real_n9_c128_b6", "real_n5_c32_b1", "real_n13_c192_b6",
                  "real_n19_c128_b3", "real_n19_c256_b20"]


def build_realistic_net(name):
    """(net with the realistic-scale seed-generated weights, input planes f32, kwargs, upstream logits, upstream value)."""
    from alphazero_amd.model import AlphaZeroNet
    z = golden_io.load("network_realistic")
    kw = json.loads(bytes(z[f"{name}/cfg_json"]).decode())
    net = AlphaZeroNet(**kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    gen = stubnets.realistic_state_dict(shapes, seed=hash_name(name))
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gen.items()})
    net.eval()
    c = 2 * kw.get("history_len", 8) + 5
    x = stubnets.synthetic_planes(int(z[f"{name}/rows"][0]), c, kw["board_size"], seed=hash_name(name) + 1)
    return net, x, kw, z[f"{name}/logits"], z[f"{name}/value"]
