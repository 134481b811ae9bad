"""Device PUCT search (HIP, through the C ABI) against golden vectors from the upstream MCTS and
against the oracle, with stub evaluators so every side sees identical network outputs."""
import numpy as np
import pytest
import torch

import golden_io
from gpu_util import stub_evaluator

pytestmark = pytest.mark.gpu

CASES = golden_io.search_cases()


def _engine(**kw):
    from alphazero_amd.engine import Engine
    return Engine(**kw)


def _root(eng, acts):
    for a in acts:
        eng.play([int(a)])


@pytest.mark.parametrize("dedup", [True, False], ids=["dedup", "nodedup"])
@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_search_vs_golden(c, dedup):
    eng = _engine(board_size=c["n"], history_len=c["hist"], games=1, komi=c["komi"], num_simulations=c["sims"],
                  inference_batch_size=c["B"], topk_expand=c["K"], virtual_loss=c["VL"], dedup_leaves=dedup)
    _root(eng, c["root_actions"])
    rows = [0]
    rng = np.random.RandomState(c["rseed"])
    visits, rv = eng.run_search(stub_evaluator(c["kind"], rows), add_noise=c["noise"], rngs=rng)
    visits = visits.cpu().numpy()[0].astype(np.float64)
    assert np.array_equal(visits, c["visits"]), "visit counts differ"
    assert float(rv.cpu()[0]) == c["root_value"], "root value differs"
    st = eng.stats()
    assert st["sims"] == c["sims"]
    if not dedup:
        assert rows[0] == c["nn_rows"], "evaluated rows differ from upstream"
    else:
        assert rows[0] <= c["nn_rows"]
    eng.close()


@pytest.mark.parametrize("n,kind,sims,B,K,noise", [(9, "hash", 64, 8, 48, True), (9, "flatish", 48, 16, 0, True),
                                                    (5, "sharp", 40, 4, 0, False), (13, "hash", 24, 4, 20, True),
                                                    (8, "hash", 32, 8, 0, True), (11, "flatish", 24, 4, 30, True)])
def test_lockstep_search_vs_oracle(oracle_lib, n, kind, sims, B, K, noise):
    """Many games with different roots searched in lockstep; each game's visits, root value and root
    priors must equal the oracle's search of the same root."""
    G = 24
    prng = np.random.default_rng(5)
    eng = _engine(board_size=n, games=G, num_simulations=sims, inference_batch_size=B, topk_expand=K)
    orc = [oracle_lib.OracleState.new_game(n, 5.5, 8) for _ in range(G)]
    depth = prng.integers(0, 2 * n * n - 4, G)
    for ply in range(int(depth.max())):
        acts = np.full(G, -1, np.int32)
        for g in range(G):
            if ply < depth[g] and not orc[g].is_terminal():
                lm = orc[g].legal_mask()
                stones = np.nonzero(lm[:-1])[0]
                a = n * n if (len(stones) == 0 or prng.random() < 0.05) else int(prng.choice(stones))
                if orc[g].play(a).is_terminal():      # keep every root searchable
                    continue
                acts[g] = a
                orc[g].play_inplace(a)
        eng.play(acts)
    rngs = [np.random.RandomState(100 + g) for g in range(G)]
    pri = []
    visits, rv = eng.run_search(stub_evaluator(kind), add_noise=noise, rngs=rngs, capture_priors=pri)
    visits, rv = visits.cpu().numpy(), rv.cpu().numpy()
    import stubnets
    worst = 0.0
    for g in range(G):
        m = oracle_lib.OracleMCTS(lambda x: stubnets.stub_eval(kind, x), board_size=n, num_simulations=sims,
                                  inference_batch_size=B, topk_expand=K)
        ov, orv = m.search(orc[g], noise, np.random.RandomState(100 + g))
        assert np.array_equal(visits[g].astype(np.float64), ov), f"game {g}"
        assert rv[g] == orv, f"game {g}"
        a, p, v, s = m.root_children()
        got = pri[0][g]
        assert np.array_equal(np.nonzero(got)[0], np.sort(a)), "root child set differs"
        ulp = np.abs(got[a] - p) / np.spacing(p)
        worst = max(worst, float(ulp.max()))
    assert worst <= 4.0, worst       # device exp vs libm exp; everything else in the chain is exact
    eng.close()


def test_terminal_and_masked_roots_sit_out():
    eng = _engine(board_size=5, games=3, num_simulations=8, inference_batch_size=2)
    eng.play([25, 12, 12])
    eng.play([25, -1, -1])          # game 0: two passes -> terminal
    visits, rv = eng.run_search(stub_evaluator("hash"), enable=[1, 1, 0])
    v = visits.cpu().numpy()
    assert v[0].sum() == 0 and v[2].sum() == 0 and v[1].sum() == 8
    eng.close()


def test_full_size_search_dedup_invariance():
    """BASELINE configs[2] at full size (4096 games, 192ch x 12 net, 256 sims, top-k 48, B 16), one search from
    positions a few random plies in: evaluating a repeated leaf once or every time gives the same visit counts and
    root values, every game's visits add up to the simulation count, and the every-leaf run evaluates
    games x simulations leaves."""
    import torch
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(0)
    net = AlphaZeroNet(channels=192, num_blocks=12).eval().cuda().native("bf16")
    G, out = 4096, []
    rng = np.random.RandomState(3)
    plies = [rng.randint(0, 81, size=G).astype(np.int32) for _ in range(4)]
    for dedup in (True, False):
        eng = Engine(board_size=9, games=G, num_simulations=256, inference_batch_size=16, topk_expand=48, dedup_leaves=dedup)
        for a in plies:
            eng.play(a, raise_on_illegal=False)          # occupied points are skipped, the same way in both runs
        noise_rngs = [np.random.RandomState(100 + g) for g in range(G)]
        visits, rv = eng.run_search(net.forward, add_noise=True, rngs=noise_rngs, layout=net.layout)
        st = eng.stats()
        out.append((visits.cpu().numpy(), rv.cpu().numpy(), st))
        eng.close()
    (va, ra, sa), (vb, rb, sb) = out
    assert np.array_equal(va, vb) and np.array_equal(ra, rb)
    assert (va.sum(axis=1) == 256).all()
    assert sa["sims"] == sb["sims"] == G * 256 and sb["rows"] == sb["rows_raw"] == G * 256 - sb["terminal"]
    assert sa["rows"] < sb["rows"] and sa["rows_raw"] == sb["rows_raw"]


def test_wide_descent_kernels_agree_with_the_regular_ones(oracle_lib):
    """More games than the chip holds waves of the regular descent kernels switch the engine to the kernels that keep no
    levels below the root in registers: same visits and root values as the same roots searched in two halves by the
    regular kernels, and as the oracle's search on a sample of them. Roots late in random games, so that the trees go deep."""
    import stubnets
    n, sims, B, K, G = 9, 64, 8, 0, 2304
    rng = np.random.RandomState(11)
    plies = [rng.randint(0, n * n, size=G).astype(np.int32) for _ in range(56)]
    noise_seed = 500

    def run(lo, hi):
        eng = _engine(board_size=n, games=hi - lo, num_simulations=sims, inference_batch_size=B, topk_expand=K)
        for a in plies:
            eng.play(a[lo:hi], raise_on_illegal=False)     # occupied or illegal points are skipped, the same way everywhere
        rngs = [np.random.RandomState(noise_seed + g) for g in range(lo, hi)]
        visits, rv = eng.run_search(stub_evaluator("hash"), add_noise=True, rngs=rngs)
        st = eng.stats()
        out = visits.cpu().numpy(), rv.cpu().numpy(), st
        eng.close()
        return out

    va, ra, sa = run(0, G)
    vb0, rb0, sb0 = run(0, G // 2)
    vb1, rb1, sb1 = run(G // 2, G)
    assert np.array_equal(va, np.concatenate([vb0, vb1])) and np.array_equal(ra, np.concatenate([rb0, rb1]))
    assert sa["max_depth"] >= 3 and sa["max_depth"] == max(sb0["max_depth"], sb1["max_depth"])
    assert sa["depth_sum"] == sb0["depth_sum"] + sb1["depth_sum"] and sa["terminal"] == sb0["terminal"] + sb1["terminal"]
    for g in (0, 1, 700, 1151, 1152, 2303):
        st = oracle_lib.OracleState.new_game(n, 5.5, 8)
        for a in plies:
            if st.is_terminal():
                break
            if st.legal_mask()[a[g]]:
                st.play_inplace(int(a[g]))
        if st.is_terminal():
            continue
        m = oracle_lib.OracleMCTS(lambda x: stubnets.stub_eval("hash", x), board_size=n, num_simulations=sims,
                                  inference_batch_size=B, topk_expand=K)
        ov, orv = m.search(st, True, np.random.RandomState(noise_seed + g))
        assert np.array_equal(va[g].astype(np.float64), ov) and ra[g] == orv, f"game {g}"


def test_wide_round_kernels_agree_with_the_regular_ones():
    """The same for searches enqueued as a whole with the native network (expansion + backup + next descents in one
    launch): 2304 games in one engine against the two halves, and against the step-by-step search (one launch per
    phase), bit for bit."""
    import torch
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(1)
    net = AlphaZeroNet(channels=64, num_blocks=2).eval().cuda().native("bf16")
    n, sims, B, K, G = 9, 256, 16, 3, 2304     # three children per node and 16 batches: the trees go deep
    rng = np.random.RandomState(12)
    plies = [rng.randint(0, n * n, size=G).astype(np.int32) for _ in range(60)]

    def run(lo, hi, rounds):
        eng = Engine(board_size=n, games=hi - lo, num_simulations=sims, inference_batch_size=B, topk_expand=K)
        for a in plies:
            eng.play(a[lo:hi], raise_on_illegal=False)
        rngs = [np.random.RandomState(900 + g) for g in range(lo, hi)]
        visits, rv = eng.run_search(net.forward, add_noise=True, rngs=rngs, layout=net.layout, native_net=net if rounds else None)
        st = eng.stats()
        out = visits.cpu().numpy(), rv.cpu().numpy(), st
        eng.close()
        return out

    va, ra, sa = run(0, G, True)
    vb0, rb0, sb0 = run(0, G // 2, True)
    vb1, rb1, sb1 = run(G // 2, G, True)
    vc, rc, sc = run(0, G, False)
    assert np.array_equal(va, np.concatenate([vb0, vb1])) and np.array_equal(ra, np.concatenate([rb0, rb1]))
    assert np.array_equal(va, vc) and np.array_equal(ra, rc)
    assert (va.sum(axis=1) == sims).all()
    for k in ("sims", "rows", "rows_raw", "depth_sum", "terminal"):
        assert sa[k] == sb0[k] + sb1[k] == sc[k], k
    assert sa["max_depth"] >= 3, sa
