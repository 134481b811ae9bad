"""az_lane_* through the C ABI: a whole self-play move per call (train.py:322-343 for every game of an engine). The calls'
order is enforced, one move equals the step-by-step search plus upstream's move choice, and the lockstep driver plays the
same games with the native lane as with its NumPy host half."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lane_setup(games=6, sims=24, batch=4, topk=6, collect=1, temp_start=1.0, temp_moves=30, set_async=True):
    import torch
    from alphazero_amd import _native as N
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import HostRng
    torch.manual_seed(3)
    net = AlphaZeroNet(board_size=5, channels=32, num_blocks=2).eval().cuda().native("bf16")
    eng = Engine(board_size=5, games=games, num_simulations=sims, inference_batch_size=batch, topk_expand=topk)
    net.reserve(eng.max_rows)
    if set_async:
        eng.set_async(True)
    rng = HostRng(games, 2)
    for s in range(games):
        rng.seed(s, 100 + s)
    feat = torch.zeros((eng.max_rows, 25, eng.cpad), dtype=torch.bfloat16, device="cuda")
    logits = torch.zeros((eng.max_rows, eng.A), dtype=torch.float32, device="cuda")
    values = torch.zeros((eng.max_rows,), dtype=torch.float32, device="cuda")
    plen = sims + 2
    tabs = np.zeros((3, plen), np.float64)
    for i, t in enumerate((temp_start, 0.5, 0.1)):
        tabs[i] = np.power(np.arange(plen, dtype=np.float64), 1.0 / t)
    lc = N.AzLaneConfig(temp_moves, 10, temp_start, 0.5, 0.1, -0.9, 1, 12, 0.3, collect, 2, 0)
    keep = (net, eng, rng, feat, logits, values, tabs)
    return N, lc, keep


def _create(N, lc, keep, pow_len=None):
    net, eng, rng, feat, logits, values, tabs = keep
    h = C.c_void_p()
    rc = N.lib().az_lane_create(eng._h, net._h, rng._h, C.byref(lc), tabs.ctypes.data_as(C.c_void_p), tabs.shape[1] if pow_len is None else pow_len,
                                C.c_void_p(feat.data_ptr()), int(net.layout), C.c_void_p(logits.data_ptr()), C.c_void_p(values.data_ptr()), C.byref(h))
    return rc, h


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_lane_calls_are_checked():
    N, lc, keep = _lane_setup(set_async=False)
    L = N.lib()
    rc, h = _create(N, lc, keep)
    assert rc == N.AZ_ERR_STATE and b"asynchronous" in L.az_last_error()       # the engine must run without round trips
    keep[1].set_async(True)
    rc, h = _create(N, lc, keep, pow_len=5)
    assert rc == N.AZ_ERR_ARG and b"power tables" in L.az_last_error()
    rc, h = _create(N, lc, keep)
    assert rc == 0
    r = N.AzLaneResult()
    assert L.az_lane_launch(h, 0, _stream()) == N.AZ_ERR_STATE                  # no roots prepared
    assert L.az_lane_finish(h, C.byref(r)) == N.AZ_ERR_STATE                    # no move in flight
    p_on, p_mi = C.c_void_p(), C.c_void_p()
    assert L.az_lane_slots(h, C.byref(p_on), C.byref(p_mi)) == 0
    on = np.ctypeslib.as_array(C.cast(p_on, C.POINTER(C.c_uint8)), shape=(6,))
    on[:] = [1, 1, 1, 0, 1, 1]
    assert L.az_lane_prepare(h, _stream()) == 0
    on[3] = 1                                                                   # switched on behind the lane's back
    assert L.az_lane_launch(h, 0, _stream()) == N.AZ_ERR_STATE and b"az_lane_prepare first" in L.az_last_error()
    on[3] = 0
    assert L.az_lane_launch(h, 0, _stream()) >= 0
    assert L.az_lane_prepare(h, _stream()) == N.AZ_ERR_STATE                    # a move is in flight
    assert L.az_lane_launch(h, 0, _stream()) == N.AZ_ERR_STATE
    assert L.az_lane_finish(h, C.byref(r)) == 0
    assert r.counters[0] == 5 * 24 and r.counters[7] == 5 and r.n_over == 0    # five slots searched 24 simulations each
    assert L.az_lane_destroy(h) == 0
    keep[1].close()


def test_one_move_equals_the_step_by_step_search_and_upstream_move_choice():
    """The lane's first move against the same roots searched through az_mcts_* with the same noise, and the move
    np.random.choice draws from visits ** (1 / T) with the stream position the lane's draws leave (mcts.py:265-283)."""
    import torch
    from alphazero_amd.engine import Engine
    N, lc, keep = _lane_setup(games=5, temp_start=1.0)
    net, eng, rng, feat, logits, values, tabs = keep
    L = N.lib()
    rc, h = _create(N, lc, keep)
    assert rc == 0
    p_on, p_mi = C.c_void_p(), C.c_void_p()
    L.az_lane_slots(h, C.byref(p_on), C.byref(p_mi))
    np.ctypeslib.as_array(C.cast(p_on, C.POINTER(C.c_uint8)), shape=(5,))[:] = 1
    assert L.az_lane_prepare(h, _stream()) == 0
    assert L.az_lane_launch(h, 0, _stream()) >= 0
    G, A = 5, eng.A
    r = N.AzLaneResult()
    actions, rv, to_play = np.zeros(G, np.int32), np.zeros(G, np.float64), np.zeros(G, np.int32)
    pi32 = np.zeros((G, A), np.float32)
    r.actions, r.root_value, r.to_play, r.pi32 = actions.ctypes.data, rv.ctypes.data, to_play.ctypes.data, pi32.ctypes.data
    assert L.az_lane_finish(h, C.byref(r)) == 0
    assert (to_play == 1).all()                                    # black moves first
    # the same search, step by step, with NumPy's own draws from the same seeds
    ref = Engine(board_size=5, games=G, num_simulations=24, inference_batch_size=4, topk_expand=6)
    streams = [np.random.RandomState(100 + s) for s in range(G)]
    visits, rv_ref = ref.run_search(net.forward, add_noise=True, rngs=streams, layout=net.layout)
    visits, rv_ref = visits.cpu().numpy().astype(np.float64), rv_ref.cpu().numpy()
    assert np.array_equal(rv, rv_ref)
    for g in range(G):
        p = np.power(visits[g], 1.0)
        p = p / p.sum()
        assert np.array_equal(pi32[g], p.astype(np.float32)), g
        assert actions[g] == streams[g].choice(A, p=p), g
    L.az_lane_destroy(h)
    eng.close()
    ref.close()


@pytest.mark.parametrize("kw", [dict(), dict(temp_moves=2, temp_mid_moves=5, temp_mid=0.4, temp_end=0.05),
                                dict(temp_moves=1, temp_mid_moves=2, temp_end=1e-3, resign_threshold=0.02),
                                dict(temp_start=1e-3, dirichlet_eps=0.0, resign_threshold=0.3, mcts_virtual_loss=0.0),
                                dict(opening_book_prob=1.0, opening_book_max_moves=3)],
                         ids=["defaults", "three_temperatures", "argmax_and_resign", "no_noise_no_virtual_loss", "opening_book"])
def test_native_lane_and_numpy_host_half_play_the_same_games(monkeypatch, kw):
    """Temperature phases (sampled and arg-max moves, mcts.py:265-283), resignation (train.py:333-337), searches without
    noise or virtual loss, book openings: the move's tail on the device against the NumPy host half."""
    import torch
    from alphazero_amd import selfplay
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(3)
    net = AlphaZeroNet(board_size=5, channels=32, num_blocks=2).eval().cuda()
    cfg = selfplay.SelfPlayConfig(board_size=5, mcts_simulations=24, mcts_inference_batch=4, mcts_topk_expand=6, **kw)
    runs = []
    for native in (True, False):
        monkeypatch.setattr(selfplay, "NATIVE_LANE", native)
        sp = selfplay.BatchedSelfPlay(net, cfg, games=6, seed0=11)
        assert sp.async_search
        sp.run(num_games=9)
        assert sp.lanes[0].native == native
        runs.append(sp)
    a, b = runs
    assert sorted(a.finished) == sorted(b.finished) == list(range(9)) and a.results == b.results
    for gid in range(9):
        ta, wa = a.trajectory(gid)
        tb, wb = b.trajectory(gid)
        assert wa == wb and len(ta) == len(tb)
        for ma, mb in zip(ta, tb):
            for xa, xb in zip(ma, mb):
                assert np.array_equal(xa, xb)
    sa, sb = a.collect_samples(), b.collect_samples()
    assert len(sa[0]) == len(sb[0]) and sa[1:] == sb[1:]


def test_lane_streams_overlap():
    """az_streams_overlap tells a stream apart from itself (kernels on one stream run in turn) and the streams the
    driver picks for its lanes run kernels at the same time."""
    import torch
    from alphazero_amd import _native as N
    from alphazero_amd.selfplay import lane_streams
    L = N.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    s = torch.cuda.Stream()
    flag = C.c_int32(7)
    assert L.az_streams_overlap(C.c_void_p(s.cuda_stream), C.c_void_p(s.cuda_stream), 200, dev.index, C.byref(flag)) == 0
    assert flag.value == 0
    a, b = lane_streams(dev, 2)
    assert a.cuda_stream != b.cuda_stream
    assert L.az_streams_overlap(C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), 200, dev.index, C.byref(flag)) == 0
    assert flag.value == 1
    assert L.az_streams_overlap(None, None, 200, dev.index, None) == N.AZ_ERR_ARG
    many = lane_streams(dev, 9)                     # more lanes than hardware queues: the streams found are shared
    assert len(many) == 9 and len({x.cuda_stream for x in many}) >= 2
