"""The CPU oracle's search (mcts.py restatement) against golden vectors from the upstream MCTS."""
import numpy as np
import pytest

import golden_io
import stubnets

CASES = golden_io.search_cases()


def root_for(oracle, c):
    st = oracle.OracleState.new_game(c["n"], c["komi"], c["hist"])
    for a in c["root_actions"]:
        st.play_inplace(int(a))
    return st


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_search_visits_and_run(oracle_lib, c):
    rows = [0]

    def ev(x):
        rows[0] += x.shape[0]
        return stubnets.stub_eval(c["kind"], x)

    m = oracle_lib.OracleMCTS(ev, board_size=c["n"], num_simulations=c["sims"], inference_batch_size=c["B"],
                              topk_expand=c["K"], virtual_loss=c["VL"])
    root = root_for(oracle_lib, c)
    rng = np.random.RandomState(c["rseed"])
    pi, action, rv = m.run(root, temperature=c["T"], add_exploration_noise=c["noise"], rng=rng)
    a, p, v, s = m.root_children()
    visits = np.zeros(c["n"] ** 2 + 1)
    visits[a] = v
    assert np.array_equal(visits, c["visits"]), "visit counts differ"
    assert rv == c["root_value"], "root value differs"
    assert pi.tobytes() == c["pi"].tobytes()
    assert action == c["action"]
    assert rows[0] == c["nn_rows"], "number of evaluated rows differs"
    if c["noise"]:
        rng = np.random.RandomState(c["rseed"])
        assert np.array_equal(rng.dirichlet([0.3] * len(c["noise_vec"])), c["noise_vec"])


def test_priors_against_upstream(oracle_lib):
    z = golden_io.load("priors")
    worst = 0.0
    for i in range(int(z["count"][0])):
        got = oracle_lib.masked_softmax(z[f"p{i}/logits"], z[f"p{i}/legal"])
        want = z[f"p{i}/probs"]
        assert np.array_equal(got == 0.0, want == 0.0)
        nz = want > 0
        ulp = np.abs(got[nz] - want[nz]) / np.spacing(want[nz])
        worst = max(worst, float(ulp.max()))
    # libm exp vs numpy's SIMD exp may differ in the last place; everything else is exact
    assert worst <= 4.0, worst
    print("max prior drift vs upstream: %.1f ulp" % worst)
