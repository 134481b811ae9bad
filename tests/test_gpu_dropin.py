"""The upstream call surface (go_zero.GoState / MCTS / self-play loop) served by the device engine,
checked against golden vectors recorded from the upstream Python code."""
import json
import os

import numpy as np
import pytest
import torch

import golden_io
import stubnets
from gpu_util import stub_evaluator

pytestmark = pytest.mark.gpu


def test_gostate_facade_replays_golden_games():
    from go_zero import GoState, decode_action
    picks = [("rules_scripted", "ko"), ("rules_scripted", "suicide"), ("rules_scripted", "tie_komi_int"),
             ("rules_scripted", "hist1_7x7"), ("rules_5x5", "g1"), ("rules_9x9", "g0")]
    for pack, name in picks:
        rec = golden_io.games(pack)[name]
        n, hist = (int(x) for x in rec["cfg"])
        st = GoState.new_game(board_size=n, komi=float(rec["komi"][0]), history_len=hist)
        states = [st]
        for ply, a in enumerate(list(rec["actions"]) + [None]):
            m = rec["meta"][ply]
            assert np.array_equal(st.board.reshape(-1), rec["boards"][ply])
            assert (st.to_play, st.consecutive_passes, st.move_count, st.black_captures, st.white_captures) == tuple(m[:5])
            assert (-1 if st.last_move is None else st.last_move[0] * n + st.last_move[1]) == m[5]
            assert st.is_terminal() == bool(m[6]) and st.winner() == m[7]
            assert len(st.board_history) == m[8] and len(st.seen_hashes) == m[9]
            legal = golden_io.unpack_legal(rec["legal"][ply], n)
            assert st.legal_actions() == [int(x) for x in np.nonzero(legal)[0]]
            assert np.array_equal(np.array(st.score()), rec["scores"][ply])
            t = st.to_tensor()
            assert t.dtype == np.float32 and t.shape == (2 * hist + 5, n, n)
            assert np.array_equal((t[: 2 * hist + 3] > 0.5).astype(np.uint8), golden_io.unpack_planes(rec["planes"][ply], hist, n))
            if a is not None:
                st = st.play(decode_action(n, int(a)))
                states.append(st)
        # value semantics: earlier states are untouched by later plays
        assert np.array_equal(states[0].board, np.zeros((n, n), np.int8)) and states[0].move_count == 0


def test_gostate_illegal_move_raises_valueerror():
    from go_zero import GoState
    st = GoState.new_game(5).play((2, 2))
    with pytest.raises(ValueError, match="Illegal move"):
        st.play((2, 2))
    assert not st.is_legal((2, 2)) and st.is_legal(None) and not st.is_legal((7, 0))


class _Stub:
    """model(x) -> (logits, value) like a torch module, on whatever device x lives."""

    def __init__(self, kind):
        self.kind = kind

    def __call__(self, x):
        lg, v = stubnets.stub_eval(self.kind, x.detach().cpu().numpy())
        return torch.from_numpy(lg).to(x.device), torch.from_numpy(v).to(x.device)


CASES = [c for c in golden_io.search_cases() if c["name"] in
         ("grid3_uniform_B4_K48", "grid10_hash_B4_K0", "grid15_hash_B16_K48", "cfg3_open", "cfg3_late", "one_pass_root",
          "five_smoke", "five_late", "thirteen_h4", "grid21_sharp_B8_K48")]


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_mcts_run_matches_upstream(c):
    from go_zero import GoState, MCTS, decode_action
    st = GoState.new_game(board_size=c["n"], komi=c["komi"], history_len=c["hist"])
    for a in c["root_actions"]:
        st = st.play(decode_action(c["n"], int(a)))
    m = MCTS(_Stub(c["kind"]), board_size=c["n"], num_simulations=c["sims"], inference_batch_size=c["B"],
             topk_expand=c["K"], virtual_loss=c["VL"], device="cpu")
    np.random.seed(c["rseed"])
    pi, action, rv = m.run(st, temperature=c["T"], add_exploration_noise=c["noise"], return_root_value=True)
    assert pi.dtype == np.float32 and pi.tobytes() == c["pi"].tobytes()
    assert action == c["action"] and rv == c["root_value"]
    np.random.seed(c["rseed"])
    out = m.run(st, temperature=c["T"], add_exploration_noise=c["noise"])
    assert len(out) == 2 and out[1] == c["action"]


def _root_parallel_cases():
    z = golden_io.load("root_parallel")
    out = []
    for c in json.loads(bytes(z["cases_json"]).decode()):
        k = c["name"]
        out.append(dict(c, root_actions=z[f"{k}/root_actions"], pi=z[f"{k}/pi"], action=int(z[f"{k}/action"][0]),
                        root_value=float(z[f"{k}/root_value"][0])))
    return out


@pytest.mark.parametrize("c", _root_parallel_cases(), ids=lambda c: c["name"])
def test_mcts_run_with_several_roots_matches_upstream(c):
    """mcts.py:240-268: simulations split over root_parallelism roots, visits summed, root values weighted by the
    simulation counts (upstream's threaded result without exploration noise, which is the reproducible case)."""
    from go_zero import GoState, MCTS, decode_action
    st = GoState.new_game(board_size=c["n"], komi=5.5, history_len=8)
    for a in c["root_actions"]:
        st = st.play(decode_action(c["n"], int(a)))
    m = MCTS(_Stub(c["kind"]), board_size=c["n"], num_simulations=c["sims"], inference_batch_size=c["B"], topk_expand=c["K"],
             virtual_loss=c["VL"], root_parallelism=c["roots"], device="cpu")
    np.random.seed(c["rseed"])
    pi, action, rv = m.run(st, temperature=c["T"], add_exploration_noise=False, return_root_value=True)
    assert pi.dtype == np.float32 and pi.tobytes() == c["pi"].tobytes()
    assert action == c["action"] and rv == c["root_value"]


def test_mcts_terminal_root():
    from go_zero import GoState, MCTS
    st = GoState.new_game(5).play(None).play(None)
    pi, a, v = MCTS(_Stub("hash"), board_size=5, num_simulations=4).run(st, return_root_value=True)
    assert a == 25 and pi[-1] == 1.0 and v == st.outcome_for_current_player()


@pytest.mark.parametrize("name", ["game_hash", "game_topk", "game_resign"])
@pytest.mark.parametrize("slots", [1, 3])
def test_lockstep_selfplay_reproduces_upstream_game(name, slots):
    """Game 0 of a lockstep run seeded like the upstream run must equal the upstream trajectory,
    whatever else shares the batch."""
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
    z = golden_io.load("selfplay")
    kw = json.loads(bytes(z[f"{name}/cfg_json"]).decode())
    kind = bytes(z[f"{name}/kind"]).decode()
    cfg = SelfPlayConfig(board_size=5, **kw)
    sp = BatchedSelfPlay(stub_evaluator(kind), cfg, games=slots, seed0=321)
    samples, bw, ww, dd = sp.run(num_games=slots)
    assert bw + ww + dd == slots
    traj, winner = sp.trajectory(0)
    assert winner == int(z[f"{name}/winner"][0])
    assert len(traj) == len(z[f"{name}/pi"])
    want_actions = z[f"{name}/actions"]
    for i, (packed, capf, pi, player, difficulty) in enumerate(traj):
        assert pi.tobytes() == z[f"{name}/pi"][i].tobytes(), f"pi at move {i}"
        assert np.array_equal(packed, z[f"{name}/planes"][i]), f"planes at move {i}"
    n0 = len(traj)
    got = samples[:n0]
    assert np.array_equal(np.array([s[2] for s in got]), z[f"{name}/z"])
    assert np.array_equal(np.array([s[3] for s in got]), z[f"{name}/difficulty"])
    assert got[0][0].shape == (21, 5, 5) and got[0][0].dtype == np.float32
    del want_actions


def _small_selfplay(lanes, use_async, games=6, num_games=9, steps=None, reuse_roots=True, graph_rounds=False):
    import torch
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
    torch.manual_seed(3)
    net = AlphaZeroNet(board_size=5, channels=32, num_blocks=2).eval().cuda()
    cfg = SelfPlayConfig(board_size=5, mcts_simulations=24, mcts_inference_batch=4, mcts_topk_expand=6)
    sp = BatchedSelfPlay(net, cfg, games=games, seed0=11, lanes=lanes, tower_workgroups=None,
                         reuse_roots=reuse_roots, graph_rounds=graph_rounds)
    assert sp.async_search
    sp.set_async(use_async)
    assert sp.async_search == use_async
    if steps is None:
        sp.run(num_games=num_games)
    else:
        sp.begin(num_games=num_games)
        for _ in range(steps):
            sp.step()
        sp.flush()
    return sp


def test_async_search_equals_synchronous_search():
    """The lockstep driver enqueues the batches of a search without host round trips when the network is
    the native bf16 one; the games must be the same as with a round trip per batch."""
    a = _small_selfplay(1, True, steps=5)
    b = _small_selfplay(1, False, steps=5)
    assert len(a.log) == len(b.log) == 5
    for ra, rb in zip(a.log, b.log):
        for xa, xb in zip(ra, rb):
            assert np.array_equal(xa, xb)
    for k in ("sims", "nn_rows", "nn_rows_nodedup", "depth_sum", "moves"):
        assert a.counters[k] == b.counters[k], k


def test_graph_replay_of_a_moves_rounds_plays_the_same_games():
    """az_mcts_run_rounds with use_graph: the launches of a move's batch rounds are captured once per lane and replayed
    for every later move; trajectories and counters equal those of the rounds enqueued one by one."""
    plain = _small_selfplay(2, True, steps=6)
    graph = _small_selfplay(2, True, steps=6, graph_rounds=True)
    assert plain.counters["graph_replays"] == 0
    assert graph.counters["graph_replays"] == 2 * 6          # every move of both lanes, the capturing call included
    assert len(plain.lanes[0].log) == len(graph.lanes[0].log) == 6
    for la, lb in zip(plain.lanes, graph.lanes):
        for ra, rb in zip(la.log, lb.log):
            for xa, xb in zip(ra, rb):
                assert np.array_equal(xa, xb)
    for k in ("sims", "nn_rows", "nn_rows_nodedup", "depth_sum", "moves"):
        assert plain.counters[k] == graph.counters[k], k


def test_run_rounds_contract():
    """az_mcts_run_rounds: refused outside asynchronous mode and before the root is expanded; az_net_capturable is false while
    launch timing is on or the scratch is not reserved, and then the rounds are enqueued one by one (return value 0)."""
    import torch
    from alphazero_amd import _native as N
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(1)
    net = AlphaZeroNet(board_size=5, channels=32, num_blocks=1).eval().cuda()
    nn = net.native("bf16", shared=False)
    eng = Engine(board_size=5, games=4, num_simulations=8, inference_batch_size=4, topk_expand=6)
    L = nn.L
    feat = torch.zeros((eng.max_rows, 25, 32), dtype=torch.bfloat16, device="cuda")
    lg = torch.zeros((eng.max_rows, 26), dtype=torch.float32, device="cuda")
    v = torch.zeros((eng.max_rows,), dtype=torch.float32, device="cuda")
    assert L.az_net_capturable(nn._h, eng.max_rows) == 0          # nothing reserved yet
    nn.reserve(eng.max_rows)
    assert L.az_net_capturable(nn._h, eng.max_rows) == 1
    nn.profile(True)
    assert L.az_net_capturable(nn._h, eng.max_rows) == 0          # launch timing records events per launch
    nn.profile(False)
    eng.search_begin()
    with pytest.raises(N.NativeError):                             # not in asynchronous mode
        eng.run_rounds(nn, feat, N.AZ_FEAT_BF16_NHWC, lg, v)
    eng.set_async(True)
    with pytest.raises(N.NativeError):                             # the root rows are still pending
        eng.run_rounds(nn, feat, N.AZ_FEAT_BF16_NHWC, lg, v)
    rows = eng.pending()
    f0 = eng.gather(rows, N.AZ_FEAT_BF16_NHWC)
    l0, v0 = nn(f0)
    assert not eng.expand(l0, v0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        s.wait_stream(torch.cuda.default_stream())
        assert eng.run_rounds(nn, feat, N.AZ_FEAT_BF16_NHWC, lg, v, graph=False) is False
        visits, _ = eng.result()
        st = eng.stats()
    assert st["sims"] == 4 * 8 and int(visits.sum()) == 4 * 8
    eng.close()


@pytest.mark.parametrize("lanes", [2, 3])
def test_pipelined_lanes_play_the_same_games(lanes):
    """Games depend on their id only: splitting the slots over lanes that run on their own streams, with
    the moves of one lane overlapping the searches of the others, must not change any trajectory."""
    one = _small_selfplay(1, True)
    many = _small_selfplay(lanes, True)
    assert sorted(one.finished) == sorted(many.finished) == list(range(9))
    assert one.results == many.results
    for gid in range(9):
        ta, wa = one.trajectory(gid)
        tb, wb = many.trajectory(gid)
        assert wa == wb and len(ta) == len(tb)
        for ma, mb in zip(ta, tb):
            for xa, xb in zip(ma, mb):
                assert np.array_equal(xa, xb)
    sa, sb = one.collect_samples(), many.collect_samples()
    assert len(sa[0]) == len(sb[0]) and sa[1:] == sb[1:]


def test_lockstep_selfplay_on_19x19_layer_path():
    """19x19 (3 points per lane on the board kernels, network layer by layer): a few lockstep moves run, every
    game's visit counts add up to the simulation count, and the batches of a search are enqueued without host round
    trips (the layer kernels read the row count on the device); one round trip per batch plays the same moves."""
    import torch
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
    torch.manual_seed(5)
    net = AlphaZeroNet(board_size=19, channels=64, num_blocks=1).eval().cuda()
    cfg = SelfPlayConfig(board_size=19, mcts_simulations=12, mcts_inference_batch=4, mcts_topk_expand=16)
    logs = []
    for use_async in (True, False):
        sp = BatchedSelfPlay(net, cfg, games=5, seed0=2)
        assert sp.async_search and not sp.net.fused
        sp.set_async(use_async)
        assert sp.async_search == use_async
        sp.begin(num_games=5)
        for _ in range(3):
            sp.step()
        assert sp.counters["sims"] == 3 * 5 * 12 and sp.counters["moves"] == 15
        for bits, caps, pi32, to_play, diff in sp.log:
            assert pi32.shape == (5, 362) and np.allclose(pi32.sum(axis=1), 1.0, atol=1e-5)
        logs.append((list(sp.log), dict(sp.counters)))
    for (ba, ca, pa, ta, da), (bb, cb, pb, tb, db) in zip(logs[0][0], logs[1][0]):
        assert np.array_equal(ba, bb) and np.array_equal(pa, pb) and np.array_equal(ta, tb)
    assert logs[0][1]["nn_rows"] == logs[1][1]["nn_rows"]


@pytest.mark.parametrize("n,channels", [(13, 192), (19, 128), (13, 128), (15, 192), (16, 128)])
def test_lockstep_selfplay_on_large_board_fused_tower(n, channels):
    """13x13 with 192 channels and 19x19 with 128: the fused tower serves the search (one board per workgroup of 176 or
    384 rows); a few lockstep moves with the batches enqueued without round trips equal the same moves with one round
    trip per batch."""
    import torch
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
    torch.manual_seed(8)
    net = AlphaZeroNet(board_size=n, channels=channels, num_blocks=2).eval().cuda()
    cfg = SelfPlayConfig(board_size=n, mcts_simulations=24, mcts_inference_batch=8, mcts_topk_expand=20)
    logs = []
    for use_async in (True, False):
        sp = BatchedSelfPlay(net, cfg, games=7, seed0=5)
        assert sp.net.fused and sp.async_search
        sp.set_async(use_async)
        sp.begin(num_games=7)
        for _ in range(3):
            sp.step()
        assert sp.counters["sims"] == 3 * 7 * 24
        logs.append([x[2].copy() for x in sp.log])
    for pa, pb in zip(*logs):
        assert pa.shape == (7, n * n + 1) and np.array_equal(pa, pb)


def test_layer_path_selfplay_row_bound_above_one_chunk_stays_asynchronous():
    """A layer-path search whose row bound (games x inference batch) is above the network's default chunk: reserving the
    bound grows the activation arrays, so the batches are still enqueued without round trips in one pass (the kernels
    walk their tiles with a capped grid: workgroups past the live rows would otherwise cost more than the live ones);
    one round trip per batch plays the same moves."""
    import torch
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
    torch.manual_seed(6)
    net = AlphaZeroNet(board_size=19, channels=64, num_blocks=1).eval().cuda()
    cfg = SelfPlayConfig(board_size=19, mcts_simulations=32, mcts_inference_batch=16, mcts_topk_expand=16)
    games = 160
    logs = []
    for use_async in (True, False):
        sp = BatchedSelfPlay(net, cfg, games=games, seed0=3, lanes=1)
        ln = sp.lanes[0]
        assert ln.max_rows == games * 16 and ln.max_rows > 2178        # 96 MiB / (361 positions x 64 channels x 2 bytes)
        assert not sp.net.fused and ln.net.indirect_passes(ln.max_rows) == 1 and sp.async_search
        sp.set_async(use_async)
        sp.begin(num_games=games)
        for _ in range(2):
            sp.step()
        assert sp.counters["sims"] == 2 * games * 32
        logs.append([x[2].copy() for x in sp.log])
    for pa, pb in zip(*logs):
        assert np.array_equal(pa, pb)


def test_bench_line_contract():
    """bench.py prints one JSON line with the fields the driver reads (small game count so it takes seconds)."""
    import json as _json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--games", "192", "--lanes", "2",
                          "--no-cpu-baseline", "--no-side-legs"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = _json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "sims/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["value"] == pytest.approx(2 * 192 * 256 / (d["ms_per_step"] * 2e-3), rel=1e-3)


def test_full_length_selfplay_config2():
    """BASELINE configs[1]: 9x9, 64ch x 4 blocks, 192 sims, 512 lockstep games, played to the end with slot refill
    (768 games through 512 slots in two lanes). Size-independent properties: every game ends, by two passes or at
    the 162-move cap or by resignation; per-game sample counts equal the number of moves searched; z is the
    winner seen from the side to move; pi rows are distributions."""
    import torch
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
    torch.manual_seed(1)
    net = AlphaZeroNet(board_size=9, channels=64, num_blocks=4).eval().cuda()
    cfg = SelfPlayConfig(board_size=9, mcts_simulations=192, mcts_inference_batch=16, mcts_topk_expand=24)
    sp = BatchedSelfPlay(net, cfg, games=512, seed0=100, lanes=2, tower_workgroups=1024)
    samples, bw, ww, dd = sp.run(num_games=768)
    assert bw + ww + dd == 768 and sorted(sp.finished) == list(range(768))
    total = 0
    for gid in (0, 1, 511, 512, 767):
        traj, winner = sp.trajectory(gid)
        assert 1 <= len(traj) <= 162 and winner in (-1, 0, 1)
        for packed, capf, pi, player, difficulty in traj:
            assert abs(float(pi.sum()) - 1.0) < 1e-5 and player in (-1, 1) and 0.01 <= difficulty <= 1.0
    for gid in sp.finished:
        total += sp.finished[gid][3] - sp.finished[gid][2]
    assert len(samples) == total == sp.counters["moves"]
    assert sp.counters["sims"] == 192 * sp.counters["moves"]
    zs = np.array([s[2] for s in samples])
    assert set(np.unique(zs)) <= {-1.0, 0.0, 1.0}
    st = sp.eng.stats()
    assert st["max_nodes"] <= 1 + 193 * 24 and st["max_depth"] <= 164


@pytest.mark.parametrize("use_async", [True, False])
def test_root_reuse_plays_the_same_games_with_fewer_evaluations(use_async):
    """The logits of the child that gets played stand in for the evaluation of the next root (the network gives a
    position the same output in any batch): same games, one network row less per move after a game's first."""
    a = _small_selfplay(1, use_async, reuse_roots=True)
    b = _small_selfplay(1, use_async, reuse_roots=False)
    assert sorted(a.finished) == sorted(b.finished) == list(range(9)) and a.results == b.results
    for gid in range(9):
        ta, wa = a.trajectory(gid)
        tb, wb = b.trajectory(gid)
        assert wa == wb and len(ta) == len(tb)
        for ma, mb in zip(ta, tb):
            for xa, xb in zip(ma, mb):
                assert np.array_equal(xa, xb)
    assert a.counters["moves"] == b.counters["moves"]
    saved = b.counters["nn_rows"] - a.counters["nn_rows"]
    assert saved == a.counters["moves"] - 9          # every move but the first of each of the 9 games


def test_mcts_run_from_worker_threads():
    """The scripts call MCTS.run from a UI worker thread (battle_ui_pyqt.py:178) and GoState methods from the main
    thread in between: one MCTS object, reused across threads and moves, gives the main-thread result."""
    import threading
    from go_zero import GoState, MCTS
    st = GoState.new_game(board_size=5).play((2, 2)).play((1, 1))
    m = MCTS(_Stub("hash"), board_size=5, num_simulations=24, inference_batch_size=4, device="cpu")
    want = m.run(st, temperature=1e-6, add_exploration_noise=False, return_root_value=True)
    got = [None, None]

    def work(i, state):
        got[i] = m.run(state, temperature=1e-6, add_exploration_noise=False, return_root_value=True)

    threads = [threading.Thread(target=work, args=(i, st)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    for g in got:
        assert g is not None and g[0].tobytes() == want[0].tobytes() and g[1] == want[1] and g[2] == want[2]
    nxt = st.play(divmod(want[1], 5))                       # the caller goes on with the state on its own thread
    assert nxt.move_count == 3 and nxt.to_play == st.to_play * -1


def test_run_search_enqueued_without_round_trips_equals_stepwise():
    """Engine.run_search with the fused bf16 network handle (what MCTS.run and the arena pass) enqueues all batches
    of a search at once; visits and root values equal the batch-by-batch loop's, also with games sitting out."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd import _native as N
    torch.manual_seed(4)
    dev = torch.device("cuda:0")
    net = AlphaZeroNet(board_size=9, channels=64, num_blocks=2).eval().to(dev).native("bf16")
    assert net.fused
    eng = Engine(board_size=9, games=5, num_simulations=72, inference_batch_size=8, topk_expand=16)
    rng = np.random.RandomState(3)
    for _ in range(6):
        legal = eng.legal_mask().cpu().numpy()[:, :-1]
        eng.play(torch.tensor([int(rng.choice(np.nonzero(row)[0])) for row in legal], dtype=torch.int32))
    for enable in (None, np.array([1, 0, 1, 1, 0], np.uint8)):
        v0, r0 = eng.run_search(net.forward, enable=enable, layout=net.layout)
        v1, r1 = eng.run_search(net.forward, enable=enable, layout=net.layout, native_net=net)
        assert torch.equal(v0, v1) and torch.equal(r0, r1)
        assert int(v1.sum()) == 72 * (5 if enable is None else int(enable.sum()))
    eng.close()


def test_bench_side_legs():
    """The legs bench.py runs after the timed window (small game count): plies per finished game -> games/hour, the
    every-leaf window, and the HBM-side census of the search / rules kernels."""
    import json as _json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--games", "192", "--lanes", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert 20 < d["mean_plies_per_game"] <= 162 and d["games_per_hour"] == pytest.approx(d["moves_per_s"] * 3600 / d["mean_plies_per_game"], rel=1e-2)
    assert 0 < d["value_no_dedup"] < d["value"]
    ks = d["roofline_other"]
    assert [k["kernel"].split()[0] for k in ks] == ["k_select", "k_leaf", "k_expand_backup"]
    for k in ks:
        assert k["bound"] == "hbm" and k["avg_launch_us"] > 0 and abs(k["frac"] - k["achieved_TBps"] / k["peak_TBps"]) < 1e-3


def test_config5_shape_lockstep_moves():
    """BASELINE configs[4] at its full network and search shape, few games: 19x19, 256 channels x 20 SE blocks, history 8,
    800 simulations per move, inference batch 16, top-k 48. Two lockstep moves of 6 games, three ways: batches enqueued
    without round trips (layer kernels read the row count on the device), one round trip per batch, and every leaf
    evaluated instead of one row per distinct leaf. Size-independent properties: each game's visit counts sum to 800,
    pi rows are distributions, the three ways play the same moves with the same pi, and the every-leaf run evaluates
    one row per simulation that does not end in a terminal leaf, plus the root."""
    import torch
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
    torch.manual_seed(19)
    net = AlphaZeroNet(board_size=19, channels=256, num_blocks=20).eval().cuda()
    cfg = SelfPlayConfig(board_size=19, mcts_simulations=800, mcts_inference_batch=16, mcts_topk_expand=48)
    runs = {}
    for tag, use_async, dedup in (("async", True, True), ("sync", False, True), ("every_leaf", True, False)):
        sp = BatchedSelfPlay(net, cfg, games=6, seed0=40, dedup_leaves=dedup, reuse_roots=False)
        assert not sp.net.fused and sp.async_search          # 96 rows at most: one chunk, no empty passes
        sp.set_async(use_async)
        sp.begin(num_games=6)
        for _ in range(2):
            sp.step()
        c = sp.counters
        assert c["sims"] == 2 * 6 * 800 and c["moves"] == 12
        assert 12 * 801 - 12 * 16 <= c["nn_rows_nodedup"] <= 12 * 801        # leaves after two passes are terminal: no row
        if not dedup:
            assert c["nn_rows"] == c["nn_rows_nodedup"]
        else:
            assert 12 * 40 <= c["nn_rows"] < 12 * 801        # about one row per batch and game (a batch may also end in a terminal leaf)
        for bits, caps, pi32, to_play, diff in sp.log:
            assert pi32.shape == (6, 362) and np.allclose(pi32.sum(axis=1), 1.0, atol=1e-5)
        board, _, sc, _ = sp.eng.export_state()
        runs[tag] = ([x[2].copy() for x in sp.log], board.cpu().numpy(), sc.cpu().numpy()[:, :6])
        st = sp.eng.stats()
        assert st["sims"] == 6 * 800 and st["max_nodes"] <= 1 + 801 * 48
    for tag in ("sync", "every_leaf"):
        for pa, pb in zip(runs["async"][0], runs[tag][0]):
            assert np.array_equal(pa, pb), tag
        assert np.array_equal(runs["async"][1], runs[tag][1]) and np.array_equal(runs["async"][2], runs[tag][2]), tag
