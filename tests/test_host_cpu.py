"""CPU-only checks: the C-ABI library loads and exports every declared symbol, host-side logic
(policy from visits, action sampling, per-game random streams) matches NumPy / the upstream formulas,
and the product refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "azamd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    from alphazero_amd import _native as N
    lib = N.lib()
    names = declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f"libazamd.so lacks {name}"
    assert set(names) == set(N.SIGNATURES), "ctypes table and header disagree"


def test_zobrist_keys_are_stable():
    from alphazero_amd import _native as N
    lib = N.lib()
    keys = [lib.az_zobrist_key(i) for i in range(2 * 361 + 2)]
    assert len(set(keys)) == len(keys)
    assert lib.az_zobrist_key(0) == lib.az_zobrist_key(0)


def test_argument_errors_without_gpu():
    from alphazero_amd import _native as N
    lib = N.lib()
    bad = N.AzConfig(board_size=25, history_len=8, games=1)
    h = C.c_void_p()
    assert lib.az_engine_create(C.byref(bad), C.byref(h)) == N.AZ_ERR_ARG
    assert b"board_size" in lib.az_last_error()
    nbad = N.AzNetConfig(board_size=9, in_planes=21, channels=190, blocks=1, use_se=1, se_hidden=8, value_hidden=8, precision=0, device=0)
    assert lib.az_net_create(C.byref(nbad), C.byref(h)) == N.AZ_ERR_ARG


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU refusal")
def test_no_cpu_fallback():
    from alphazero_amd import _native as N
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet
    with pytest.raises(N.NativeError):
        Engine(board_size=5)
    with pytest.raises(N.NativeError):
        AlphaZeroNet(board_size=5, channels=32, num_blocks=1).native("bf16")


def test_host_rng_streams_match_numpy_legacy():
    from alphazero_amd.selfplay import HostRng
    G = 40
    rng = HostRng(G, threads=4)
    ref = []
    for g in range(G):
        rng.seed(g, 500 + g)
        ref.append(np.random.RandomState(500 + g))
    gen = np.random.default_rng(0)
    for _ in range(25):
        counts = gen.integers(0, 83, G).astype(np.int32)
        out = rng.dirichlet(counts, 0.3, 82)
        mask = (gen.random(G) < 0.7).astype(np.uint8)
        u = rng.random_sample(mask)
        for g in range(G):
            if counts[g]:
                assert np.array_equal(out[g, : counts[g]], ref[g].dirichlet([0.3] * counts[g]))
            if mask[g]:
                assert u[g] == ref[g].random_sample()
    # alpha >= 1 exercises the other branch of the legacy gamma sampler
    for alpha in (1.0, 2.5):
        out = rng.dirichlet(np.full(G, 7, np.int32), alpha, 82)
        for g in range(G):
            assert np.array_equal(out[g, :7], ref[g].dirichlet([alpha] * 7))


def test_policy_and_sampling_match_upstream_formulas():
    from alphazero_amd.selfplay import policy_from_visits, sample_actions
    from alphazero_amd.mcts import visits_to_pi
    gen = np.random.default_rng(1)
    G, A = 64, 82
    visits = np.floor(gen.random((G, A)) ** 6 * 300).astype(np.float64)
    visits[3] = 0
    visits[4, :] = 0
    visits[4, 17] = 256
    temps = gen.choice([1.0, 0.4, 0.05, 1e-6], G)
    pi = policy_from_visits(visits, temps)
    for g in range(G):
        want = visits_to_pi(visits[g], temps[g])          # the single-game form (mcts.py:270-279)
        assert pi[g].tobytes() == want.tobytes(), (g, temps[g])
    # np.random.choice(A, p=pi) consumes one random_sample and does a right-sided search in the cdf
    for g in range(G):
        rs = np.random.RandomState(g)
        want = int(rs.choice(A, p=pi[g]))
        u = np.random.RandomState(g).random_sample()
        got = int(sample_actions(pi[g:g + 1], np.array([u]))[0])
        assert got == want


def test_pb_c_table_matches_python_math():
    """The engine tabulates log((N + c_base + 1) / c_base) + c_init with the host libm; it must equal
    what CPython's math.log gives (mcts.py:104)."""
    import math
    lib = C.CDLL("libm.so.6")
    lib.log.restype = C.c_double
    lib.log.argtypes = [C.c_double]
    for n in range(0, 2000):
        x = (n + 19652.0 + 1.0) / 19652.0
        assert lib.log(x) == math.log(x)


def test_selfplay_config_from_train_config():
    from alphazero_amd.selfplay import SelfPlayConfig

    class T:
        board_size, komi, history_len, mcts_simulations, mcts_inference_batch = 5, 6.5, 4, 32, 4
        mcts_topk_expand, mcts_virtual_loss, resign_threshold, disable_resign = 6, 0.5, 0.9, True
    c = SelfPlayConfig.from_train_config(T)
    assert (c.board_size, c.komi, c.history_len, c.mcts_simulations, c.mcts_topk_expand, c.disable_resign) == (5, 6.5, 4, 32, 6, True)
    assert c.c_base == 19652.0 and c.temp_moves == 20
