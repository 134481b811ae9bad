#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the upstream Python
implementation (mounted read-only at /root/reference) with fixed seeds.

Run only in the build container:  python tests/golden/make_golden.py
Nothing from the upstream tree is copied: only inputs (action lists, seeds, stub-net names)
and the outputs it produced are stored. The GPU box never runs this script.

Families (one compressed .npz each):
  rules_*.npz   per-ply boards, scalars, legal masks, scores and packed feature planes for
                seeded random playouts and scripted edge-case games          (go.py)
  search.npz    visit counts / root value / pi / action / Dirichlet noise for stub-net
                searches over a grid of MCTS settings                          (mcts.py)
  priors.npz    masked-softmax probability vectors (float64) for given logits  (mcts.py:72-89)
  network.npz   AlphaZeroNet outputs for seed-generated weights and inputs     (model.py)
  selfplay.npz  one full seeded 5x5 self-play game incl. a forced-resign game  (train.py:314)
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
import torch

from go_zero.go import GoState, decode_action, encode_action  # noqa: E402  (upstream)
from go_zero.mcts import MCTS  # noqa: E402
from go_zero.model import AlphaZeroNet  # noqa: E402
import go_zero.train as ref_train  # noqa: E402

import stubnets  # noqa: E402

torch.set_num_threads(4)


# ----------------------------------------------------------------------------- rules
def snapshot(st: GoState):
    n = st.board_size
    a = n * n + 1
    legal = np.zeros(a, np.uint8)
    legal[st.legal_actions()] = 1
    t = st.to_tensor()
    h = st.history_len
    planes = np.packbits((t[: 2 * h + 3] > 0.5).reshape(-1))
    capf = np.array([t[2 * h + 3, 0, 0], t[2 * h + 4, 0, 0]], np.float32)
    sb, sw = st.score()
    meta = np.array(
        [st.to_play, st.consecutive_passes, st.move_count, st.black_captures, st.white_captures,
         -1 if st.last_move is None else encode_action(n, st.last_move), int(st.is_terminal()),
         st.winner(), len(st.board_history), len(st.seen_hashes)], np.int32)
    return st.board.reshape(-1).copy(), meta, np.packbits(legal), planes, capf, np.array([sb, sw], np.float64)


def classify_illegal(st: GoState):
    """Count empty points that are illegal by suicide vs by superko (coverage statistic)."""
    n = st.board_size
    suicide = ko = 0
    for r in range(n):
        for c in range(n):
            if st.board[r, c] != 0 or st.is_legal((r, c)):
                continue
            b = st.board.copy()
            b[r, c] = st.to_play
            st._remove_dead_groups_around(b, r, c, -st.to_play)
            _, libs = st._group_and_liberties(b, r, c)
            if not libs:
                suicide += 1
            else:
                ko += 1
    return suicide, ko


def record_game(n, komi, hist, actions=None, seed=0, pass_prob=0.05, max_plies=None, stats=None):
    rng = random.Random(seed)
    st = GoState.new_game(board_size=n, komi=komi, history_len=hist)
    rows = [snapshot(st)]
    acts = []
    ply = 0
    while True:
        if actions is not None:
            if ply >= len(actions):
                break
            a = actions[ply]
        else:
            if st.is_terminal() or (max_plies is not None and ply >= max_plies):
                break
            legal = st.legal_actions()
            stones = legal[:-1]
            a = legal[-1] if (not stones or rng.random() < pass_prob) else rng.choice(stones)
        if stats is not None:
            s, k = classify_illegal(st)
            stats["suicide_points"] += s
            stats["superko_points"] += k
        before = st.black_captures + st.white_captures
        st = st.play(decode_action(n, a))
        if stats is not None and st.black_captures + st.white_captures > before:
            stats["capturing_moves"] += 1
        acts.append(a)
        rows.append(snapshot(st))
        ply += 1
    cols = list(zip(*rows))
    return {
        "cfg": np.array([n, hist], np.int32), "komi": np.array([komi], np.float64),
        "actions": np.array(acts, np.int32), "boards": np.stack(cols[0]), "meta": np.stack(cols[1]),
        "legal": np.stack(cols[2]), "planes": np.stack(cols[3]), "capf": np.stack(cols[4]),
        "scores": np.stack(cols[5]),
    }


def rc(n, *pairs):
    return [n * n if p is None else p[0] * n + p[1] for p in pairs]


def scripted_games():
    g = {}
    n = 5
    # corner single-stone capture, later a two-stone group captured in the centre
    g["capture_corner"] = (5, 5.5, 8, rc(n, (0, 1), (0, 0), (1, 0), (2, 2), (2, 1), (3, 2), (1, 2), None,
                                         (3, 1), (4, 4), (4, 2), None, (3, 3), None, (2, 3)))
    # one black move removes two separate white groups
    g["capture_two_groups"] = (5, 5.5, 8, rc(n, (1, 0), (0, 0), (1, 2), (0, 2), (0, 3), (4, 4), (0, 1)))
    # multi-stone suicide point (4,1) for White, then single-stone suicide point (0,0) for White
    g["suicide"] = (5, 5.5, 8, rc(n, (3, 0), (4, 0), (3, 1), (0, 4), (4, 2), (0, 3), (0, 1), (1, 4), (1, 0), None,
                                  (2, 2)))
    # simple ko in the centre: immediate retake forbidden, allowed after an exchange elsewhere
    g["ko"] = (5, 5.5, 8, rc(n, (1, 1), (1, 2), (2, 0), (2, 3), (3, 1), (3, 2), (2, 2), (2, 1), (4, 4), (0, 0),
                             (2, 2), (4, 0), (0, 4), (2, 1)))
    # pass-pass ending; repetition through passes (pass performs no superko check)
    g["pass_pass"] = (5, 5.5, 8, rc(n, (2, 2), None, None))
    g["pass_repeat"] = (5, 5.5, 8, rc(n, (2, 2), None, (1, 1), None))
    # komi ties -> winner 0
    g["tie_komi0"] = (5, 0.0, 8, rc(n, None, None))
    g["tie_komi_int"] = (5, 25.0, 8, rc(n, (2, 2), None, None))
    # short histories
    g["hist2"] = (5, 5.5, 2, rc(n, (2, 2), (1, 1), None, (3, 3), (0, 0)))
    g["hist1_7x7"] = (7, 6.5, 1, rc(7, (3, 3), (2, 2), None, (4, 4)))
    return g


def gen_rules(outdir):
    stats = {"suicide_points": 0, "superko_points": 0, "capturing_moves": 0, "plies": 0, "games": 0}
    packs = {"rules_5x5": {}, "rules_9x9": {}, "rules_19x19": {}, "rules_scripted": {}}
    for name, (n, komi, hist, acts) in scripted_games().items():
        try:
            rec = record_game(n, komi, hist, actions=acts, stats=stats)
        except ValueError as e:  # keep the legal prefix of a script that hit an illegal move
            raise SystemExit(f"scripted game {name} is illegal upstream: {e}")
        packs["rules_scripted"][name] = rec
    for s in range(12):
        packs["rules_5x5"][f"g{s}"] = record_game(5, 5.5, 8, seed=100 + s, pass_prob=0.02 if s % 3 else 0.15, stats=stats)
    packs["rules_5x5"]["h3"] = record_game(5, 2.5, 3, seed=150, pass_prob=0.02, stats=stats)
    packs["rules_5x5"]["six"] = record_game(6, 5.5, 4, seed=151, pass_prob=0.03, stats=stats)
    for s in range(5):
        packs["rules_9x9"][f"g{s}"] = record_game(9, 5.5, 8, seed=200 + s, pass_prob=0.01 if s % 2 else 0.06, stats=stats)
    packs["rules_9x9"]["k7"] = record_game(9, 7.0, 8, seed=260, pass_prob=0.0, stats=stats)
    packs["rules_19x19"]["g0"] = record_game(19, 7.5, 8, seed=300, pass_prob=0.0, stats=stats)
    packs["rules_19x19"]["g1_13"] = record_game(13, 5.5, 8, seed=301, pass_prob=0.004, stats=stats)
    for pack, games in packs.items():
        flat = {}
        for gname, rec in games.items():
            stats["games"] += 1
            stats["plies"] += len(rec["actions"])
            for k, v in rec.items():
                flat[f"{gname}/{k}"] = v
        np.savez_compressed(os.path.join(outdir, pack + ".npz"), **flat)
    return stats


# ----------------------------------------------------------------------------- search
def position_after(n, komi, hist, plies, seed, pass_prob=0.0):
    rng = random.Random(seed)
    st = GoState.new_game(board_size=n, komi=komi, history_len=hist)
    acts = []
    while len(acts) < plies and not st.is_terminal():
        legal = st.legal_actions()
        stones = legal[:-1]
        a = legal[-1] if (not stones or rng.random() < pass_prob) else rng.choice(stones)
        st = st.play(decode_action(n, a))
        acts.append(a)
    return st, acts


def search_cases():
    cases = []

    def add(name, n=9, plies=0, pseed=1, kind="hash", sims=16, B=1, K=0, VL=1.0, noise=False, T=1.0,
            rseed=7, extra_pass=False, hist=8, komi=5.5):
        cases.append(dict(name=name, n=n, plies=plies, pseed=pseed, kind=kind, sims=sims, B=B, K=K, VL=VL,
                          noise=noise, T=T, rseed=rseed, extra_pass=extra_pass, hist=hist, komi=komi))

    i = 0
    for kind in ("uniform", "hash", "sharp"):
        for B in (1, 4, 8, 16):
            for K in (0, 48):
                sims = (2, 16, 192, 256)[i % 4]
                VL = (1.0, 0.0)[(i // 2) % 2] if B > 1 else 1.0
                plies = (0, 30, 60)[i % 3]
                add(f"grid{i}_{kind}_B{B}_K{K}", plies=plies, pseed=10 + i, kind=kind, sims=sims, B=B, K=K, VL=VL,
                    noise=bool(i % 2), T=(1.0, 0.4, 0.05, 1e-6)[i % 4], rseed=1000 + i)
                i += 1
    # the headline configuration, opening / midgame / near the move cap
    add("cfg3_open", kind="flatish", sims=256, B=16, K=48, noise=True, rseed=42)
    add("cfg3_mid", plies=40, pseed=77, kind="hash", sims=256, B=16, K=48, noise=True, rseed=43)
    add("cfg3_mid_b8", plies=40, pseed=78, kind="flatish", sims=256, B=8, K=48, noise=True, rseed=44)
    add("cfg3_late", plies=156, pseed=79, kind="hash", sims=256, B=16, K=48, noise=True, rseed=45)
    add("late_b1", plies=158, pseed=80, kind="sharp", sims=192, B=1, K=0, noise=False, rseed=46)
    add("one_pass_root", plies=50, pseed=81, kind="hash", sims=64, B=4, K=0, noise=True, rseed=47, extra_pass=True)
    add("vl_half", plies=20, pseed=82, kind="hash", sims=96, B=8, K=16, VL=0.5, noise=True, rseed=48)
    add("deep_b1", plies=10, pseed=83, kind="sharp", sims=256, B=1, K=8, noise=False, rseed=49)
    add("cfg2_like", plies=12, pseed=84, kind="hash", sims=192, B=8, K=0, noise=True, rseed=50)
    # other board sizes
    add("five_smoke", n=5, kind="hash", sims=2, B=4, K=0, noise=True, rseed=51)
    add("five_long", n=5, plies=20, pseed=85, kind="sharp", sims=128, B=4, K=0, noise=True, rseed=52)
    add("five_late", n=5, plies=44, pseed=86, kind="hash", sims=64, B=2, K=6, noise=False, rseed=53)
    add("nineteen", n=19, plies=60, pseed=87, kind="hash", sims=48, B=8, K=48, noise=True, rseed=54, komi=7.5)
    add("nineteen_full", n=19, plies=20, pseed=88, kind="flatish", sims=24, B=4, K=0, noise=True, rseed=55, komi=7.5)
    add("thirteen_h4", n=13, plies=30, pseed=89, kind="hash", sims=40, B=4, K=20, noise=True, rseed=56, hist=4)
    return cases


def gen_search(outdir):
    flat = {}
    meta = []
    for c in search_cases():
        st, acts = position_after(c["n"], c["komi"], c["hist"], c["plies"], c["pseed"])
        if c["extra_pass"] and not st.is_terminal():
            st = st.play(None)
            acts.append(c["n"] * c["n"])
        if st.is_terminal():
            raise SystemExit(f"search case {c['name']} has a terminal root")
        model = stubnets.TorchStub(c["kind"])
        m = MCTS(model, board_size=c["n"], num_simulations=c["sims"], inference_batch_size=c["B"],
                 topk_expand=c["K"], virtual_loss=c["VL"], device="cpu")
        np.random.seed(c["rseed"])
        visits, rv = m._run_single_root_search(st, c["noise"], c["sims"])
        rows_eval = model.rows
        np.random.seed(c["rseed"])
        pi, action, rv2 = m.run(st, temperature=c["T"], add_exploration_noise=c["noise"], return_root_value=True)
        assert rv2 == rv
        np.random.seed(c["rseed"])
        nlegal = len(st.legal_actions())
        noise = np.random.dirichlet([m.dirichlet_alpha] * nlegal) if c["noise"] else np.zeros(0)
        k = c["name"]
        flat[f"{k}/root_actions"] = np.array(acts, np.int32)
        flat[f"{k}/visits"] = visits.astype(np.float64)
        flat[f"{k}/root_value"] = np.array([rv], np.float64)
        flat[f"{k}/pi"] = pi.astype(np.float32)
        flat[f"{k}/action"] = np.array([action], np.int32)
        flat[f"{k}/noise"] = noise.astype(np.float64)
        flat[f"{k}/nn_rows"] = np.array([rows_eval], np.int32)
        meta.append(c)
        print(f"  search {k}: sims={c['sims']} B={c['B']} K={c['K']} rows={rows_eval} "
              f"maxvisit={int(visits.max())} nz={(visits > 0).sum()}", flush=True)
    flat["cases_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(outdir, "search.npz"), **flat)
    return len(meta)


# ----------------------------------------------------------------------------- priors
class PresetModel:
    def __init__(self, logits, values):
        self.logits, self.values = logits, values

    def __call__(self, x):
        return torch.from_numpy(self.logits), torch.from_numpy(self.values)


def gen_priors(outdir):
    flat = {}
    rng = np.random.Generator(np.random.PCG64(2024))
    idx = 0
    for n, plies_list in ((5, (0, 12, 30)), (9, (0, 25, 70, 120)), (19, (0, 100))):
        a = n * n + 1
        states = [position_after(n, 5.5, 8, p, 900 + idx + j)[0] for j, p in enumerate(plies_list)]
        for scale in (0.0, 1.0, 6.0, 40.0):
            logits = (rng.standard_normal((len(states), a)) * scale).astype(np.float32)
            if scale == 40.0:
                logits[:, 0] = 700.0  # forces exp underflow of everything else
            values = rng.uniform(-1, 1, len(states)).astype(np.float32)
            m = MCTS(PresetModel(logits, values), board_size=n, device="cpu")
            probs, vals, legal = m._predict_batch(states)
            for j in range(len(states)):
                mask = np.zeros(a, np.uint8)
                mask[legal[j]] = 1
                flat[f"p{idx}/logits"] = logits[j]
                flat[f"p{idx}/legal"] = mask
                flat[f"p{idx}/probs"] = np.asarray(probs[j], np.float64)
                flat[f"p{idx}/value"] = np.array([vals[j]], np.float64)
                idx += 1
    flat["count"] = np.array([idx], np.int32)
    np.savez_compressed(os.path.join(outdir, "priors.npz"), **flat)
    return idx


# ----------------------------------------------------------------------------- network
def gen_network(outdir):
    flat = {}
    cfgs = {
        "n5_c32_b1": dict(board_size=5, channels=32, num_blocks=1, use_se=True, value_hidden_dim=128),
        "n9_c64_b4": dict(board_size=9, channels=64, num_blocks=4, use_se=True, value_hidden_dim=256),
        "n9_c64_b2_nose": dict(board_size=9, channels=64, num_blocks=2, use_se=False, value_hidden_dim=64),
        "n9_c192_b12": dict(board_size=9, channels=192, num_blocks=12, use_se=True, value_hidden_dim=256),
        "n19_c32_b2_h4": dict(board_size=19, history_len=4, channels=32, num_blocks=2, use_se=True, value_hidden_dim=32),
    }
    for name, kw in cfgs.items():
        net = AlphaZeroNet(**kw)
        sd = net.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items()}
        gen = stubnets.deterministic_state_dict(shapes, seed=hash_name(name))
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gen.items()})
        net.eval()
        n = kw["board_size"]
        c = 2 * kw.get("history_len", 8) + 5
        x = stubnets.synthetic_planes(6, c, n, seed=hash_name(name) + 1)
        with torch.inference_mode():
            lg, v = net(torch.from_numpy(x))
        flat[f"{name}/logits"] = lg.numpy().astype(np.float32)
        flat[f"{name}/value"] = v.numpy().astype(np.float32)
        flat[f"{name}/cfg_json"] = np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)
        flat[f"{name}/shapes_json"] = np.frombuffer(
            json.dumps([[k, list(s)] for k, s in shapes.items()]).encode(), dtype=np.uint8)
        flat[f"{name}/nparams"] = np.array([sum(p.numel() for p in net.parameters())], np.int64)
        print(f"  network {name}: params={int(flat[f'{name}/nparams'][0])}", flush=True)
    np.savez_compressed(os.path.join(outdir, "network.npz"), **flat)
    return len(cfgs)


def hash_name(s: str) -> int:
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


# ----------------------------------------------------------------------------- self-play
def gen_selfplay(outdir):
    flat = {}
    for name, kind, kw in (
        ("game_hash", "hash", dict(mcts_simulations=16, mcts_inference_batch=4)),
        ("game_topk", "sharp", dict(mcts_simulations=24, mcts_inference_batch=8, mcts_topk_expand=6, temp_moves=4,
                                    temp_mid_moves=8)),
        ("game_resign", "const99", dict(mcts_simulations=8, mcts_inference_batch=1)),
    ):
        cfg = ref_train.TrainConfig(board_size=5, **kw)
        log = []
        orig_run = MCTS.run

        def spy(self, root_state, temperature=1.0, add_exploration_noise=True, return_root_value=False, _o=orig_run):
            out = _o(self, root_state, temperature, add_exploration_noise, return_root_value)
            log.append((float(temperature), int(out[1]), float(out[2])))
            return out

        MCTS.run = spy
        try:
            ref_train.set_seed(321)
            data, winner = ref_train.selfplay_game(stubnets.TorchStub(kind), cfg)
        finally:
            MCTS.run = orig_run
        h = cfg.history_len
        flat[f"{name}/kind"] = np.frombuffer(kind.encode(), dtype=np.uint8)
        flat[f"{name}/cfg_json"] = np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)
        flat[f"{name}/temps"] = np.array([x[0] for x in log], np.float64)
        flat[f"{name}/actions"] = np.array([x[1] for x in log], np.int32)
        flat[f"{name}/root_v"] = np.array([x[2] for x in log], np.float64)
        flat[f"{name}/pi"] = np.stack([d[1] for d in data]).astype(np.float32)
        flat[f"{name}/z"] = np.array([d[2] for d in data], np.float64)
        flat[f"{name}/difficulty"] = np.array([d[3] for d in data], np.float64)
        flat[f"{name}/planes"] = np.stack([np.packbits((d[0][: 2 * h + 3] > 0.5).reshape(-1)) for d in data])
        flat[f"{name}/winner"] = np.array([winner], np.int32)
        print(f"  selfplay {name}: plies={len(data)} winner={winner}", flush=True)
    defaults = ref_train.TrainConfig()
    flat["trainconfig_defaults_json"] = np.frombuffer(json.dumps(defaults.__dict__).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(outdir, "selfplay.npz"), **flat)


def main():
    out = HERE
    # the forced-resign stub lives only here and in tests/stubnets.py
    print("rules ...", flush=True)
    stats = gen_rules(out)
    print("  coverage:", stats, flush=True)
    print("search ...", flush=True)
    ns = gen_search(out)
    print("priors ...", flush=True)
    npri = gen_priors(out)
    print("network ...", flush=True)
    gen_network(out)
    print("selfplay ...", flush=True)
    gen_selfplay(out)
    info = dict(rules_coverage=stats, search_cases=ns, prior_vectors=npri,
                numpy=np.__version__, torch=torch.__version__,
                state_hash_known_answer=int(GoState._state_hash(np.zeros((5, 5), np.int8), 1)))
    with open(os.path.join(out, "MANIFEST.json"), "w") as f:
        json.dump(info, f, indent=1)
    print(info)


if __name__ == "__main__":
    main()
