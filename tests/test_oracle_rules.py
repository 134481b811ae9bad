"""The CPU oracle's rules and feature planes against the golden vectors produced by the upstream
Python implementation (go_zero/go.py)."""
import numpy as np
import pytest

import golden_io

GAMES = list(golden_io.all_rule_games())


def test_state_hash_known_answer(oracle_lib):
    # blake2b-64 over board bytes + (to_play+2), go.py:53-59; value recorded from upstream
    assert oracle_lib.state_hash(np.zeros((5, 5), np.int8), 1) == 11312466115036544367


@pytest.mark.parametrize("name,rec", GAMES, ids=[g[0] for g in GAMES])
def test_rules_trajectory(oracle_lib, name, rec):
    n, hist = (int(x) for x in rec["cfg"])
    st = oracle_lib.OracleState.new_game(n, float(rec["komi"][0]), hist)
    actions = rec["actions"]
    for ply in range(len(actions) + 1):
        board, meta = st.snapshot()
        assert np.array_equal(board, rec["boards"][ply]), f"board differs at ply {ply}"
        assert np.array_equal(meta, rec["meta"][ply]), f"scalars differ at ply {ply}: {meta} vs {rec['meta'][ply]}"
        assert np.array_equal(st.legal_mask(), golden_io.unpack_legal(rec["legal"][ply], n)), f"legal mask ply {ply}"
        assert np.array_equal(np.array(st.score()), rec["scores"][ply]), f"score ply {ply}"
        t = st.to_tensor()
        assert np.array_equal((t[: 2 * hist + 3] > 0.5).astype(np.uint8),
                              golden_io.unpack_planes(rec["planes"][ply], hist, n)), f"planes ply {ply}"
        assert set(np.unique(t[: 2 * hist + 3])) <= {0.0, 1.0}
        capf = np.array([t[2 * hist + 3, 0, 0], t[2 * hist + 4, 0, 0]], np.float32)
        assert capf.tobytes() == rec["capf"][ply].tobytes(), f"capture planes ply {ply}"
        assert (t[2 * hist + 3] == capf[0]).all() and (t[2 * hist + 4] == capf[1]).all()
        if ply < len(actions):
            st = st.play(int(actions[ply]))


def test_illegal_move_raises(oracle_lib):
    st = oracle_lib.OracleState.new_game(5)
    st = st.play(12)
    with pytest.raises(ValueError, match="Illegal move"):
        st.play(12)


def test_pairwise_sum_matches_numpy(oracle_lib):
    rng = np.random.default_rng(0)
    for n in (1, 3, 7, 8, 9, 26, 82, 127, 128, 129, 170, 362, 1000):
        for _ in range(20):
            a = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)
            assert oracle_lib.pairwise_sum(a) == float(a.sum()), n
