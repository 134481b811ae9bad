"""Device rules / feature planes (HIP, through the C ABI) against the golden vectors and the oracle."""
import numpy as np
import pytest
import torch

import golden_io
from gpu_util import device_snapshot

pytestmark = pytest.mark.gpu

GAMES = list(golden_io.all_rule_games())


def _engine(**kw):
    from alphazero_amd.engine import Engine
    return Engine(**kw)


@pytest.mark.parametrize("name,rec", GAMES, ids=[g[0] for g in GAMES])
def test_rules_trajectory_vs_golden(name, rec):
    n, hist = (int(x) for x in rec["cfg"])
    eng = _engine(board_size=n, history_len=hist, games=1, komi=float(rec["komi"][0]))
    actions = rec["actions"]
    for ply in range(len(actions) + 1):
        s = device_snapshot(eng)
        want = rec["meta"][ply]
        assert np.array_equal(s["board"][0], rec["boards"][ply]), f"board ply {ply}"
        # info: to_play, passes, move_count, bcaps, wcaps, last_move, terminal, winner
        assert np.array_equal(s["info"][0], want[:8]), f"scalars ply {ply}: {s['info'][0]} vs {want[:8]}"
        assert s["sc"][0][6] == want[8], f"history count ply {ply}"
        assert s["sc"][0][7] == want[9], f"seen-set size ply {ply}"
        assert np.array_equal(s["legal"][0], golden_io.unpack_legal(rec["legal"][ply], n)), f"legal ply {ply}"
        assert np.array_equal(s["score"][0], rec["scores"][ply]), f"score ply {ply}"
        t = s["feats"][0]
        assert np.array_equal((t[: 2 * hist + 3] > 0.5).astype(np.uint8), golden_io.unpack_planes(rec["planes"][ply], hist, n))
        assert set(np.unique(t[: 2 * hist + 3])) <= {0.0, 1.0}
        capf = np.array([t[2 * hist + 3, 0, 0], t[2 * hist + 4, 0, 0]], np.float32)
        assert capf.tobytes() == rec["capf"][ply].tobytes()
        assert (t[2 * hist + 3] == capf[0]).all() and (t[2 * hist + 4] == capf[1]).all()
        if ply < len(actions):
            eng.play([int(actions[ply])])
    eng.close()


@pytest.mark.parametrize("n,games,seed", [(5, 96, 1), (9, 128, 2), (13, 32, 3), (19, 16, 4),
                                          (8, 48, 5), (11, 32, 6), (2, 16, 7)])
def test_lockstep_random_games_vs_oracle(oracle_lib, n, games, seed):
    """G games in lockstep, each following its own random legal line; every ply every game is compared
    with the oracle (board, scalars, legal mask, feature planes, score)."""
    rng = np.random.default_rng(seed)
    eng = _engine(board_size=n, history_len=8, games=games, komi=5.5)
    orc = [oracle_lib.OracleState.new_game(n, 5.5, 8) for _ in range(games)]
    plies = 2 * n * n + 3 if n <= 9 else 120
    for ply in range(plies):
        s = device_snapshot(eng)
        acts = np.full(games, -1, np.int32)
        for g in range(games):
            ob, om = orc[g].snapshot()
            assert np.array_equal(s["board"][g], ob), (ply, g)
            assert np.array_equal(s["info"][g], om[:8]), (ply, g, s["info"][g], om)
            lm = orc[g].legal_mask()
            assert np.array_equal(s["legal"][g], lm), (ply, g)
            assert np.array_equal(s["score"][g], np.array(orc[g].score())), (ply, g)
            assert s["feats"][g].tobytes() == orc[g].to_tensor().tobytes(), (ply, g)
            if orc[g].is_terminal():
                continue
            stones = np.nonzero(lm[:-1])[0]
            a = n * n if (len(stones) == 0 or rng.random() < 0.03) else int(rng.choice(stones))
            acts[g] = a
            orc[g].play_inplace(a)
        if (acts < 0).all():
            break
        eng.play(acts)
    eng.close()


def test_illegal_move_reports_and_keeps_state():
    eng = _engine(board_size=5, games=2)
    eng.play([12, 12])
    before = device_snapshot(eng)
    with pytest.raises(ValueError, match="Illegal move"):
        eng.play([12, 3])            # occupied point in game 0
    st = eng.play([12, -1], raise_on_illegal=False).cpu().numpy()
    assert st[0] == -4 and st[1] == 0
    after = device_snapshot(eng)
    assert np.array_equal(before["board"][0], after["board"][0])
    assert np.array_equal(before["info"][0], after["info"][0])
    eng.close()


def test_bf16_feature_layout_matches_f32():
    from alphazero_amd import _native as N
    eng = _engine(board_size=9, games=8)
    rng = np.random.default_rng(0)
    for _ in range(30):
        lm = eng.legal_mask().cpu().numpy()
        acts = [int(rng.choice(np.nonzero(lm[g][:-1])[0])) for g in range(8)]
        eng.play(acts)
    f32 = eng.features().cpu()
    bf = eng.features(N.AZ_FEAT_BF16_NHWC).cpu().float()          # [G, 81, 32]
    assert bf.shape == (8, 81, 32)
    want = f32.reshape(8, 21, 81).permute(0, 2, 1)
    assert torch.equal(bf[:, :, :21], want.to(torch.bfloat16).float())
    assert (bf[:, :, 21:] == 0).all()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,hist", [(9, 16), (5, 1), (13, 8)])
def test_bf16_feature_layout_other_plane_counts(n, hist):
    """The channel-minor bf16 planes for plane counts other than 21: more than 32 planes (history 16: 37 planes padded to
    64 channels), the minimum (history 1: 7 planes), and a board with several points per lane; root features and the
    leaf gather of a search both equal the float32 planes, captures included (the capture planes are not 0/1)."""
    from alphazero_amd import _native as N
    G = 6
    eng = _engine(board_size=n, history_len=hist, games=G, num_simulations=8, inference_batch_size=4)
    rng = np.random.default_rng(n + hist)
    for _ in range(n * n - n):     # boards nearly full: captures have happened
        lm = eng.legal_mask().cpu().numpy()
        acts = [int(rng.choice(np.nonzero(lm[g][:-1])[0])) if lm[g][:-1].any() else n * n for g in range(G)]
        eng.play(acts)
    C, cpad, npts = 2 * hist + 5, (32 if 2 * hist + 5 <= 32 else 64), n * n
    for source in ("root", "leaf"):
        if source == "root":
            f32, bf = eng.features().cpu(), eng.features(N.AZ_FEAT_BF16_NHWC).cpu().float()
        else:
            eng.search_begin()
            rows = eng.pending()
            assert rows == G
            f32, bf = eng.gather(rows).cpu(), eng.gather(rows, N.AZ_FEAT_BF16_NHWC).cpu().float()
        assert bf.shape == (G, npts, cpad)
        want = f32.reshape(G, C, npts).permute(0, 2, 1)
        assert torch.equal(bf[:, :, :C], want.to(torch.bfloat16).float()), source
        assert (bf[:, :, C:] == 0).all()
        assert (want[:, :, 2 * hist + 3:].max() > 0)          # some captures happened
    eng.close()
