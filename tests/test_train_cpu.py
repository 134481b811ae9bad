"""Host-side training callers against vectors made by the upstream train.py (tests/golden/train.*):
config and command line, symmetries, replay sampling, sample filter, arena statistics, Elo, model pool,
mutation, opening book, and the training step itself (CPU float32, one torch thread)."""
import copy
import dataclasses
import json
import os
import random
from collections import deque

import numpy as np
import pytest
import torch

import golden_io
from alphazero_amd import train as T

Z = golden_io.load("train")
J = json.load(open(os.path.join(golden_io.GOLDEN, "train.json")))


def synth_samples(count, n, hist, seed):
    """Same generator as tests/golden/make_golden_train.py."""
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
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
    return m.eval()


def test_config_defaults_and_field_set():
    mine = dataclasses.asdict(T.TrainConfig())
    assert set(mine) == set(J["defaults"]) | set(T.ENGINE_FIELDS)
    assert {k: mine[k] for k in J["defaults"]} == J["defaults"]
    assert T.UPSTREAM_FIELDS == tuple(J["defaults"])          # same order: positional construction agrees too


@pytest.mark.parametrize("case", J["cli"], ids=lambda c: " ".join(c["argv"])[:40] or "no-args")
def test_command_line(case):
    mine = dataclasses.asdict(T.parse_args(case["argv"]))
    assert {k: mine[k] for k in case["cfg"]} == case["cfg"]


def test_create_mcts_kwargs():
    cfg = T.TrainConfig()
    m = T.create_mcts(model=None, cfg=cfg, sims=cfg.mcts_simulations, add_noise=True)
    for k, v in J["create_mcts"].items():
        assert getattr(m, k) == v, k


@pytest.mark.parametrize("sym", range(8))
def test_symmetries(sym):
    st, pol = Z["sym/state"], Z["sym/policy"]
    out = T.apply_symmetry_to_tensor(st, sym)
    assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and np.array_equal(out, Z[f"sym/state_{sym}"])
    assert np.array_equal(T.transform_2d(st[3], sym), Z[f"sym/state_{sym}"][3])
    pout = T.apply_symmetry_to_policy(pol, 5, sym)
    assert pout.dtype == np.float32 and np.array_equal(pout, Z[f"sym/policy_{sym}"])


def test_replay_buffer_sampling():
    samples = synth_samples(70, 5, 8, seed=11)
    buf = T.ReplayBuffer(capacity=50)
    buf.extend(samples[:30])
    assert np.array_equal(buf.sampling_probabilities(), Z["replay/probs_30"])
    buf.extend(samples[30:])
    assert len(buf) == J["replay"]["len"] and buf.counter == J["replay"]["counter"] and buf.data[0][4] == J["replay"]["first_counter"]
    assert np.array_equal(buf.sampling_probabilities(), Z["replay/probs_50"])
    for aug in (0, 1):
        random.seed(100 + aug)
        np.random.seed(200 + aug)
        s, p, v = buf.sample(batch_size=16, board_size=5, augment_symmetry=bool(aug))
        assert s.dtype == p.dtype == v.dtype == torch.float32
        assert np.array_equal(s.numpy(), Z[f"replay/states_{aug}"])
        assert np.array_equal(p.numpy(), Z[f"replay/policies_{aug}"])
        assert np.array_equal(v.numpy(), Z[f"replay/values_{aug}"])
    with pytest.raises(RuntimeError, match=J["replay"]["empty_error"]):
        T.ReplayBuffer(4).sample(2, 5, False)


def test_sample_filter():
    fs = synth_samples(12, 5, 8, seed=21)
    fs[3] = (fs[1][0].copy(), fs[3][1], fs[3][2], fs[3][3])
    heavy = fs[5][1].copy(); heavy[:] = 0.0; heavy[-1] = 1.0
    fs[5] = (fs[5][0], heavy, fs[5][2], fs[5][3])
    fs[9] = (fs[0][0].copy(), fs[9][1], fs[9][2], fs[9][3])
    keys, kset = deque(), set()
    where = lambda kept: [next(i for i, x in enumerate(fs) if x[0] is k[0] and x[1] is k[1]) for k in kept]
    assert where(T.filter_selfplay_samples(fs, keys, kset, max_recent=6)) == J["filter"]["kept"]
    assert (len(keys), len(kset)) == (J["filter"]["recent_len"], J["filter"]["set_len"])
    assert where(T.filter_selfplay_samples(fs, keys, kset, max_recent=6)) == J["filter"]["kept_second_pass"]


def test_arena_statistics_and_elo():
    for w, l, d, lo, hi in J["score_interval"]:
        assert T._score_interval_from_counts(w, l, d) == (lo, hi)
    for ra, rb, sc, g, k, ex, na, nb in J["elo"]:
        assert T.expected_score(ra, rb) == ex and T.update_elo_pair(ra, rb, sc, g, k) == (na, nb)


def test_model_pool_selection_and_pruning(tmp_path):
    pool = copy.deepcopy(J["pool_in"])
    for k, want in J["pool_opponents"]:
        random.seed(7 + k)
        assert T.sample_pool_opponents(pool, exclude_name="m8", k=k) == want
    T.prune_pool(pool, keep_name="m2", max_size=4)
    assert sorted(pool["entries"]) == J["pool_pruned"]
    path = str(tmp_path / "pool.json")
    assert T.load_model_pool(path) == {"entries": {}, "best_name": ""}
    T.save_model_pool(path, pool)
    assert T.load_model_pool(path) == pool


def test_pbt_mutation():
    for seed, lr, c_init, eps, temp_moves in J["pbt"]:
        c = T.TrainConfig()
        np.random.seed(seed)
        T.apply_pbt_mutation(c)
        assert (c.lr, c.c_init, c.dirichlet_eps, c.temp_moves) == (lr, c_init, eps, temp_moves)


def test_opening_book_lines_and_draws():
    for n, lines in J["book_lines"].items():
        assert [[list(p) for p in ln] for ln in T.opening_book_lines(int(n))] == lines
    for seed, played, board, to_play in J["book_play"]:
        c = T.TrainConfig(board_size=9, opening_book_prob=0.7, opening_book_max_moves=4)
        random.seed(seed)
        line = T.choose_opening(9, c)
        assert len(line) == played          # on an empty board every book move is legal
        want = np.zeros(81, np.int8)
        for i, (r, col) in enumerate(line):
            want[r * 9 + col] = 1 if i % 2 == 0 else -1
        assert want.tolist() == board and to_play == (1 if played % 2 == 0 else -1)
        # the lockstep driver draws from random.Random(seed): the same stream as the module after random.seed(seed)
        assert T.choose_opening(9, c, random.Random(seed)) == line


def test_train_candidate_matches_upstream_losses():
    torch.set_num_threads(1)
    cfg = tiny_cfg(augment_symmetry=True, policy_target_smoothing=0.05)
    model = tiny_model(cfg, J["tiny_seeds"]["train"])
    buf = T.ReplayBuffer(200)
    buf.extend(synth_samples(120, 5, 8, seed=31))
    T.set_seed(77)
    loss = T.train_candidate(model, buf, cfg)
    want = J["train_candidate"]
    assert np.allclose(loss, want["loss"], rtol=1e-5, atol=1e-6), (loss, want["loss"])
    got = float(sum(p.detach().abs().double().sum() for p in model.parameters()))
    assert abs(got - want["param_abs_sum"]) <= 1e-5 * want["param_abs_sum"]
    bn = float(sum(b.detach().abs().double().sum() for n_, b in model.named_buffers() if n_.endswith("running_mean")))
    assert abs(bn - want["bn_mean_abs_sum"]) <= 1e-5 * want["bn_mean_abs_sum"]


def test_go_zero_train_surface():
    import go_zero.train as G
    assert G.main is T.main and G.TrainConfig is T.TrainConfig and G.ReplayBuffer is T.ReplayBuffer
    for name in ("run_selfplay", "selfplay_game", "train_candidate", "play_match", "play_match_sprt", "run_training",
                 "create_mcts", "build_model", "filter_selfplay_samples", "apply_symmetry_to_tensor", "set_seed"):
        assert callable(getattr(G, name)), name


def test_rank_share_covers_every_game_once():
    for games in (0, 1, 7, 40, 4096):
        for world in (1, 2, 3, 8):
            parts = [T.rank_share(games, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == games
            assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
