#!/usr/bin/env python3
"""Golden vectors for the training-side callers (upstream go_zero/train.py), made by running the
upstream implementation (read-only at /root/reference) with fixed seeds in the build container:

    python tests/golden/make_golden_train.py

Only inputs (seeds, argv lists, synthetic arrays) and the outputs upstream produced are stored
(tests/golden/train.npz + train.json). The GPU box never runs this script.
"""
from __future__ import annotations

import copy
import dataclasses
import json
import os
import random
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import go_zero.train as T  # noqa: E402  (upstream)
from go_zero.go import GoState, decode_action  # noqa: E402

torch.set_num_threads(1)
ARR: dict[str, np.ndarray] = {}
J: dict = {}


def synth_samples(count, n, hist, seed):
    """Random samples shaped like self-play output: (planes f32 [C,n,n], pi f32 [A], z, difficulty)."""
    rng = np.random.RandomState(seed)
    c = 2 * hist + 5
    out = []
    for _ in range(count):
        s = (rng.rand(c, n, n) < 0.3).astype(np.float32)
        s[-2:] = rng.randint(0, 5, size=(2, 1, 1)).astype(np.float32) / (n * n)
        p = rng.dirichlet([0.5] * (n * n + 1)).astype(np.float32)
        z = float(rng.choice([-1.0, 0.0, 1.0]))
        d = float(max(0.01, 1.0 - p.max()))
        out.append((s, p, z, max(d, abs(z) * 0.25)))
    return out


def tiny_cfg(**kw):
    base = dict(board_size=5, history_len=8, channels=32, num_blocks=1, value_hidden_dim=128, device="cpu",
                mcts_simulations=8, eval_simulations=8, mcts_inference_batch=4, batch_size=16, train_steps_per_iter=5)
    base.update(kw)
    return T.TrainConfig(**base)


def tiny_model(cfg, seed):
    torch.manual_seed(seed)
    m = T.build_model(cfg)
    # non-trivial BN statistics so that eval/train mode differences show up
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
    return m.eval()


def main():
    # ---- config defaults and CLI mapping (train.py:26-102, :1026-1169)
    J["defaults"] = dataclasses.asdict(T.TrainConfig())
    argvs = [
        [],
        ["--device", "cuda", "--board-size", "5", "--no-se", "--no-arena-sprt", "--no-augment-symmetry"],
        ["--device", "cuda", "--no-amp", "--mcts-root-parallelism", "0", "--value-hidden-dim", "0",
         "--opening-book-prob", "1.7", "--opening-book-max-moves", "-3", "--policy-target-smoothing", "-0.5"],
        ["--use-amp", "--async-selfplay", "--disable-resign", "--pbt", "--save-buffer", "--lr", "3e-4",
         "--selfplay-workers", "6", "--mcts-topk-expand", "48", "--mcts-inference-batch", "16", "--resume", "x.pt"],
    ]
    J["cli"] = []
    for av in argvs:
        old = sys.argv
        sys.argv = ["train"] + av
        try:
            J["cli"].append({"argv": av, "cfg": dataclasses.asdict(T.parse_args())})
        finally:
            sys.argv = old

    # ---- create_mcts (train.py:248-263)
    cfg = T.TrainConfig()
    m = T.create_mcts(model=None, cfg=cfg, sims=cfg.mcts_simulations, add_noise=True)
    J["create_mcts"] = {k: v for k, v in m.__dict__.items() if k != "model"}

    # ---- symmetries (train.py:187-207)
    rng = np.random.RandomState(5)
    st = rng.rand(21, 5, 5).astype(np.float32)
    pol = rng.dirichlet([0.5] * 26).astype(np.float32)
    ARR["sym/state"], ARR["sym/policy"] = st, pol
    for s in range(8):
        ARR[f"sym/state_{s}"] = T.apply_symmetry_to_tensor(st, s)
        ARR[f"sym/policy_{s}"] = T.apply_symmetry_to_policy(pol, 5, s)

    # ---- replay buffer (train.py:105-176)
    samples = synth_samples(70, 5, 8, seed=11)
    buf = T.ReplayBuffer(capacity=50)
    buf.extend(samples[:30])
    buf._rebuild_sampling_cache()
    ARR["replay/probs_30"] = buf._sample_probs.copy()
    buf.extend(samples[30:])
    J["replay"] = {"len": len(buf), "counter": buf.counter, "first_counter": int(buf.data[0][4])}
    buf._rebuild_sampling_cache()
    ARR["replay/probs_50"] = buf._sample_probs.copy()
    for aug in (0, 1):
        random.seed(100 + aug)
        np.random.seed(200 + aug)
        s_t, p_t, v_t = buf.sample(batch_size=16, board_size=5, augment_symmetry=bool(aug))
        ARR[f"replay/states_{aug}"], ARR[f"replay/policies_{aug}"], ARR[f"replay/values_{aug}"] = s_t.numpy(), p_t.numpy(), v_t.numpy()
    try:
        T.ReplayBuffer(4).sample(2, 5, False)
    except RuntimeError as e:
        J["replay"]["empty_error"] = str(e)

    # ---- sample filter (train.py:210-234)
    fs = synth_samples(12, 5, 8, seed=21)
    fs[3] = (fs[1][0].copy(), fs[3][1], fs[3][2], fs[3][3])            # repeated position
    heavy = fs[5][1].copy(); heavy[:] = 0.0; heavy[-1] = 1.0            # pass-certain policy
    fs[5] = (fs[5][0], heavy, fs[5][2], fs[5][3])
    fs[9] = (fs[0][0].copy(), fs[9][1], fs[9][2], fs[9][3])
    from collections import deque
    keys, kset = deque(), set()
    kept = T.filter_selfplay_samples(fs, keys, kset, max_recent=6)
    J["filter"] = {"kept": [next(i for i, x in enumerate(fs) if x[0] is k[0] and x[1] is k[1]) for k in kept],
                   "recent_len": len(keys), "set_len": len(kset)}
    kept2 = T.filter_selfplay_samples(fs, keys, kset, max_recent=6)
    J["filter"]["kept_second_pass"] = [next(i for i, x in enumerate(fs) if x[0] is k[0] and x[1] is k[1]) for k in kept2]

    # ---- arena statistics and Elo (train.py:586-670)
    J["wilson"] = [[s, n, *T._wilson_interval(s, n)] for s, n in [(0, 0), (3, 10), (5.5, 10), (10, 10), (0, 7)]]
    J["score_interval"] = [[w, l, d, *T._score_interval_from_counts(w, l, d)] for w, l, d in
                           [(0, 0, 0), (6, 0, 0), (4, 1, 1), (1, 4, 1), (3, 3, 0), (2, 2, 6), (20, 10, 5)]]
    J["elo"] = [[ra, rb, sc, g, k, T.expected_score(ra, rb), *T.update_elo_pair(ra, rb, sc, g, k)] for ra, rb, sc, g, k in
                [(1500.0, 1500.0, 6.0, 12, 24.0), (1600.0, 1450.0, 3.5, 12, 24.0), (1400.0, 1700.0, 12.0, 12, 32.0)]]

    # ---- model pool (train.py:720-754)
    pool = {"entries": {f"m{i}": {"path": f"/nonexistent/m{i}.pt", "elo": 1500.0 + 17.0 * ((i * 7) % 5) - i, "games": 0, "iter": i}
                        for i in range(9)}, "best_name": "m8"}
    J["pool_in"] = copy.deepcopy(pool)
    J["pool_opponents"] = []
    for k in (0, 1, 2, 4, 5, 20):
        random.seed(7 + k)
        J["pool_opponents"].append([k, T.sample_pool_opponents(pool, exclude_name="m8", k=k)])
    pruned = copy.deepcopy(pool)
    T.prune_pool(pruned, keep_name="m2", max_size=4)
    J["pool_pruned"] = sorted(pruned["entries"])

    # ---- population-based mutation (train.py:673-684)
    J["pbt"] = []
    for seed in (1, 2, 3):
        c = T.TrainConfig()
        np.random.seed(seed)
        T.apply_pbt_mutation(c)
        J["pbt"].append([seed, c.lr, c.c_init, c.dirichlet_eps, c.temp_moves])

    # ---- opening book (train.py:266-311)
    J["book_lines"] = {str(n): T.opening_book_lines(n) for n in (4, 5, 9, 19)}
    J["book_play"] = []
    for seed in (1, 2, 3, 4):
        c = T.TrainConfig(board_size=9, opening_book_prob=0.7, opening_book_max_moves=4)
        random.seed(seed)
        s0 = GoState.new_game(9, 5.5, 8)
        s1, played = T.maybe_play_opening_book(s0, c)
        J["book_play"].append([seed, played, s1.board.reshape(-1).tolist(), s1.to_play])

    # ---- train_candidate (train.py:504-554), CPU float32, one torch thread
    cfg_t = tiny_cfg(augment_symmetry=True, policy_target_smoothing=0.05)
    model = tiny_model(cfg_t, 3)
    buf = T.ReplayBuffer(200)
    buf.extend(synth_samples(120, 5, 8, seed=31))
    T.set_seed(77)
    tri = T.train_candidate(model, buf, cfg_t)
    J["train_candidate"] = {"loss": tri, "param_abs_sum": float(sum(p.detach().abs().double().sum() for p in model.parameters())),
                            "bn_mean_abs_sum": float(sum(b.detach().abs().double().sum() for n_, b in model.named_buffers() if n_.endswith("running_mean")))}

    # ---- arena matches (train.py:557-660): argmax play, no noise
    cfg_a = tiny_cfg()
    ma, mb = tiny_model(cfg_a, 5), tiny_model(cfg_a, 6)
    J["play_match"] = list(T.play_match(ma, mb, cfg_a, 5))
    J["play_match_sprt"] = [list(T.play_match_sprt(ma, mb, cfg_a, 12, th, 0.05)) for th in (0.55, 0.3, 0.9)]

    # ---- self-play workers (train.py:357-392): one game per worker, seeds seed + iter*1000 + wid
    cfg_s = tiny_cfg(mcts_topk_expand=6, seed=9)
    ms = tiny_model(cfg_s, 8)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.pt")
        torch.save({"model": ms.state_dict()}, path)
        J["selfplay_workers"] = []
        for wid in range(3):
            T._WORKER_MODEL = None
            samples, bw, ww, dd = T.selfplay_worker(path, 1, dict(cfg_s.__dict__), 1, cfg_s.seed + 2 * 1000 + wid)
            ARR[f"worker{wid}/states"] = np.packbits(np.stack([s[0][:19] for s in samples]) > 0.5)
            ARR[f"worker{wid}/caps"] = np.stack([s[0][19:, 0, 0] for s in samples])
            ARR[f"worker{wid}/pi"] = np.stack([s[1] for s in samples])
            ARR[f"worker{wid}/z"] = np.array([s[2] for s in samples])
            ARR[f"worker{wid}/difficulty"] = np.array([s[3] for s in samples])
            J["selfplay_workers"].append([bw, ww, dd, len(samples)])
    # a game with the opening book on (global `random` stream, train.py:291-311)
    cfg_b = tiny_cfg(opening_book_prob=1.0, opening_book_max_moves=3, seed=4)
    T.set_seed(1234)
    data, winner = T.selfplay_game(ms, cfg_b)
    ARR["book_game/pi"] = np.stack([s[1] for s in data])
    ARR["book_game/first_planes"] = data[0][0]
    J["book_game"] = {"winner": int(winner), "moves": len(data)}
    J["tiny_seeds"] = {"train": 3, "arena": [5, 6], "selfplay": 8}

    def clean(o):
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (np.floating, np.integer, np.bool_)):
            return o.item()
        return o

    np.savez_compressed(os.path.join(HERE, "train.npz"), **ARR)
    with open(os.path.join(HERE, "train.json"), "w") as f:
        json.dump(clean(J), f, indent=1)
    print({k: (v if not isinstance(v, (list, dict)) else "...") for k, v in J.items()})


if __name__ == "__main__":
    main()
