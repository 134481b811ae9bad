#!/usr/bin/env python3
"""Golden outputs of the upstream AlphaZeroNet (go_zero/model.py:43-97, read-only under /root/reference) for shapes the
bf16 layer-by-layer kernels serve: 19x19 with 128 channels x 3 SE blocks (history 8) and 13x13 with 64 channels x 2.
Same record layout as network.npz (make_golden.gen_network): weights come from the seed-deterministic generator in
tests/stubnets.py, only the outputs are stored. Build container only:  python tests/golden/make_golden_network_large.py
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

CFGS = {
    "n19_c128_b3": dict(board_size=19, history_len=8, channels=128, num_blocks=3, use_se=True, value_hidden_dim=256),
    "n13_c64_b2": dict(board_size=13, history_len=8, channels=64, num_blocks=2, use_se=True, value_hidden_dim=64),
}


def main():
    flat = {}
    for name, kw in CFGS.items():
        net = AlphaZeroNet(**kw)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        gen = stubnets.deterministic_state_dict(shapes, seed=mg.hash_name(name))
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gen.items()})
        net.eval()
        c = 2 * kw.get("history_len", 8) + 5
        x = stubnets.synthetic_planes(6, c, kw["board_size"], seed=mg.hash_name(name) + 1)
        with torch.inference_mode():
            lg, v = net(torch.from_numpy(x))
        flat[f"{name}/logits"] = lg.numpy().astype(np.float32)
        flat[f"{name}/value"] = v.numpy().astype(np.float32)
        flat[f"{name}/cfg_json"] = np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)
        flat[f"{name}/shapes_json"] = np.frombuffer(json.dumps([[k, list(s)] for k, s in shapes.items()]).encode(), dtype=np.uint8)
        flat[f"{name}/nparams"] = np.array([sum(p.numel() for p in net.parameters())], np.int64)
        print(f"  network {name}: params={int(flat[f'{name}/nparams'][0])} |logit|max={float(lg.abs().max()):.3g}", flush=True)
    np.savez_compressed(os.path.join(HERE, "network_large.npz"), **flat)


if __name__ == "__main__":
    main()
