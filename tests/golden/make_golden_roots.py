#!/usr/bin/env python3
"""Golden vectors for MCTS.run with root_parallelism > 1 (mcts.py:240-268): the simulations are split over several
roots searched by a thread pool, visits are summed and the root values averaged with the simulation counts as weights.
Without exploration noise the roots draw nothing from the random stream, so the upstream result is reproducible.

Run only in the build container:  python tests/golden/make_golden_roots.py
Stored: inputs (action lists, settings, seeds) and upstream's outputs; nothing of the upstream tree is copied.
"""
from __future__ import annotations

import json
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/ for stubnets
sys.path.insert(0, "/root/reference")

import numpy as np

from go_zero.go import GoState, decode_action  # noqa: E402  (upstream)
from go_zero.mcts import MCTS  # noqa: E402

import stubnets  # noqa: E402

CASES = [
    dict(name="nine_r3", n=9, plies=20, pseed=31, kind="hash", sims=50, B=4, K=0, VL=1.0, roots=3, T=1.0, rseed=5),
    dict(name="nine_r4_topk", n=9, plies=41, pseed=32, kind="sharp", sims=96, B=8, K=16, VL=1.0, roots=4, T=0.4, rseed=6),
    dict(name="five_r4_uneven", n=5, plies=6, pseed=33, kind="hash", sims=7, B=1, K=0, VL=1.0, roots=4, T=1.0, rseed=7),
    dict(name="five_more_roots_than_sims", n=5, plies=3, pseed=34, kind="flatish", sims=2, B=2, K=0, VL=0.0, roots=8, T=1e-6, rseed=8),
]


def main():
    flat, meta = {}, []
    for c in CASES:
        rng = random.Random(c["pseed"])
        st = GoState.new_game(board_size=c["n"], komi=5.5, history_len=8)
        acts = []
        while len(acts) < c["plies"]:
            a = rng.choice(st.legal_actions()[:-1])
            st = st.play(decode_action(c["n"], a))
            acts.append(a)
        m = MCTS(stubnets.TorchStub(c["kind"]), board_size=c["n"], num_simulations=c["sims"], inference_batch_size=c["B"],
                 topk_expand=c["K"], virtual_loss=c["VL"], root_parallelism=c["roots"], device="cpu")
        outs = []
        for _ in range(3):      # the thread pool must not make it vary
            np.random.seed(c["rseed"])
            outs.append(m.run(st, temperature=c["T"], add_exploration_noise=False, return_root_value=True))
        assert all(o[0].tobytes() == outs[0][0].tobytes() and o[1] == outs[0][1] and o[2] == outs[0][2] for o in outs)
        pi, action, rv = outs[0]
        k = c["name"]
        flat[f"{k}/root_actions"] = np.array(acts, np.int32)
        flat[f"{k}/pi"] = pi.astype(np.float32)
        flat[f"{k}/action"] = np.array([action], np.int32)
        flat[f"{k}/root_value"] = np.array([rv], np.float64)
        meta.append(c)
        print(f"  {k}: action={action} root_value={rv!r} nonzero pi={(pi > 0).sum()}", flush=True)
    flat["cases_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "root_parallel.npz"), **flat)


if __name__ == "__main__":
    main()
