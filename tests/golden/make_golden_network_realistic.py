#!/usr/bin/env python3
"""Golden outputs of the upstream AlphaZeroNet (go_zero/model.py:43-97, read-only under /root/reference) on weights at
the scale of a freshly initialised network (tests/stubnets.realistic_state_dict: fan-in-scaled uniform weights, BatchNorm
statistics near the identity), where |logit| stays within a few units and bf16 inference is well conditioned. These are the
vectors the bf16 kernels are held to 1e-2 against (logits, probabilities, value) and the float32 kernels to 1e-5; the
large-|logit| family of network.npz / network_large.npz stays as the stress case. Only configurations, inputs' seeds and
outputs are stored: the weights are regenerated from the seed. Build container only:
    python tests/golden/make_golden_network_realistic.py
"""
from __future__ import annotations

import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_golden as mg          # imports the upstream package from /root/reference
from make_golden import AlphaZeroNet, stubnets

# name -> (constructor kwargs, rows). 37 rows leave the fused tower's last workgroup ragged.
CFGS = {
    "real_n9_c192_b12": (dict(board_size=9, history_len=8, channels=192, num_blocks=12, use_se=True, value_hidden_dim=256), 37),
    "real_n9_c64_b4": (dict(board_size=9, history_len=8, channels=64, num_blocks=4, use_se=True, value_hidden_dim=256), 37),
    "real_n9_c128_b6": (dict(board_size=9, history_len=8, channels=128, num_blocks=6, use_se=True, value_hidden_dim=256), 37),
    "real_n5_c32_b1": (dict(board_size=5, history_len=8, channels=32, num_blocks=1, use_se=True, value_hidden_dim=128), 23),
    "real_n13_c192_b6": (dict(board_size=13, history_len=8, channels=192, num_blocks=6, use_se=True, value_hidden_dim=256), 7),
    "real_n19_c128_b3": (dict(board_size=19, history_len=8, channels=128, num_blocks=3, use_se=True, value_hidden_dim=256), 7),
    "real_n19_c256_b20": (dict(board_size=19, history_len=8, channels=256, num_blocks=20, use_se=True, value_hidden_dim=256), 7),
}


def main():
    flat = {}
    for name, (kw, rows) in CFGS.items():
        net = AlphaZeroNet(**kw)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        gen = stubnets.realistic_state_dict(shapes, seed=mg.hash_name(name))
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gen.items()})
        net.eval()
        c = 2 * kw.get("history_len", 8) + 5
        x = stubnets.synthetic_planes(rows, c, kw["board_size"], seed=mg.hash_name(name) + 1)
        with torch.inference_mode():
            lg, v = net(torch.from_numpy(x))
        flat[f"{name}/logits"] = lg.numpy().astype(np.float32)
        flat[f"{name}/value"] = v.numpy().astype(np.float32)
        flat[f"{name}/cfg_json"] = np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)
        flat[f"{name}/rows"] = np.array([rows], np.int64)
        flat[f"{name}/nparams"] = np.array([sum(p.numel() for p in net.parameters())], np.int64)
        print(f"  network {name}: params={int(flat[f'{name}/nparams'][0])} rows={rows} |logit|max={float(lg.abs().max()):.3g} "
              f"logit std={float(lg.std()):.3g} |value|max={float(v.abs().max()):.3g}", flush=True)
    np.savez_compressed(os.path.join(HERE, "network_realistic.npz"), **flat)


if __name__ == "__main__":
    main()
