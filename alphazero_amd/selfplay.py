"""Lockstep self-play: G games advance one move per step on one GPU, each game reproducing what the
upstream selfplay_game (go_zero/train.py:314-354) does for it: temperature schedule, root noise,
move sampling, resignation, trajectory samples with outcome z and difficulty. Game j draws its
random numbers from its own legacy-NumPy stream seeded with seed0 + j, i.e. exactly what a
sequential upstream run of that game after np.random.seed(seed0 + j) would draw.

run(num_games) returns the tuple run_selfplay returns (train.py:395-463):
(samples, black_wins, white_wins, draws) with samples = [(planes f32 [C,n,n], pi f32 [A], z, difficulty)].
"""
from __future__ import annotations

import ctypes as C
import os
import random
import time
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as N
from .engine import GRAPH_ROUNDS, Engine

NATIVE_ROUNDS = os.environ.get("AZ_NATIVE_ROUNDS", "1") != "0"   # 0: one call per kernel from Python (A/B timing)
# 0: the per-move host half (policy targets, move sampling, resignation, play, status) in NumPy with a device round trip per
# piece, as before the native lane (az_lane_*) existed; same games either way (A/B timing and cross-check)
NATIVE_LANE = os.environ.get("AZ_NATIVE_LANE", "1") != "0"


@dataclass
class SelfPlayConfig:
    """The TrainConfig fields (train.py:26-102) that self-play reads, with upstream defaults."""
    board_size: int = 9
    komi: float = 5.5
    history_len: int = 8
    mcts_simulations: int = 192
    mcts_inference_batch: int = 8
    mcts_topk_expand: int = 0
    mcts_virtual_loss: float = 1.0
    c_base: float = 19652.0
    c_init: float = 1.25
    dirichlet_alpha: float = 0.3
    dirichlet_eps: float = 0.25
    temp_moves: int = 20
    temp_mid_moves: int = 40
    temp_start: float = 1.0
    temp_mid: float = 0.4
    temp_end: float = 0.05
    resign_threshold: float = 0.98
    disable_resign: bool = False
    opening_book_prob: float = 0.0
    opening_book_max_moves: int = 0

    @classmethod
    def from_train_config(cls, cfg) -> "SelfPlayConfig":
        return cls(**{k: getattr(cfg, k) for k in cls.__dataclass_fields__ if hasattr(cfg, k)})


def opening_book_lines(board_size: int):
    """train.py:266-288: four short centre openings, duplicates and off-board points dropped."""
    if board_size < 5:
        return []
    c = board_size // 2
    lo, hi = max(0, c - 1), min(board_size - 1, c + 1)
    raw = [[(c, c), (lo, c), (c, lo), (hi, c), (c, hi)],
           [(c, c), (c, lo), (lo, c), (c, hi), (hi, c)],
           [(lo, lo), (hi, hi), (lo, hi), (hi, lo), (c, c)],
           [(lo, c), (hi, c), (c, lo), (c, hi), (c, c)]]
    lines = []
    for line in raw:
        pts = [p for p in dict.fromkeys(line) if 0 <= p[0] < board_size and 0 <= p[1] < board_size]
        if pts:
            lines.append(pts)
    return lines


def choose_opening(board_size: int, cfg, rng=random):
    """The draws of train.py:291-302 from `rng` (the global `random` module upstream): the book moves to try."""
    if cfg.opening_book_prob <= 0.0 or cfg.opening_book_max_moves <= 0:
        return []
    if rng.random() >= cfg.opening_book_prob:
        return []
    lines = opening_book_lines(board_size)
    if not lines:
        return []
    line = rng.choice(lines)
    return line[: min(len(line), cfg.opening_book_max_moves)]


class HostRng:
    """G legacy-NumPy-compatible MT19937 streams (az_rng_* in the C ABI)."""

    def __init__(self, count: int, threads: int = 8):
        self.L = N.lib()
        self.count, self.threads = count, threads
        h = C.c_void_p()
        N.check(self.L.az_rng_create(count, C.byref(h)), "az_rng_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self.L.az_rng_destroy(self._h)
            self._h = None

    def seed(self, index: int, seed: int):
        N.check(self.L.az_rng_seed(self._h, int(index), int(seed) & 0xFFFFFFFF), "az_rng_seed")

    def dirichlet(self, counts: np.ndarray, alpha: float, stride: int) -> np.ndarray:
        counts = np.ascontiguousarray(counts, np.int32)
        out = np.zeros((self.count, stride), np.float64)
        N.check(self.L.az_rng_dirichlet(self._h, counts.ctypes.data_as(C.c_void_p), float(alpha),
                                        out.ctypes.data_as(C.c_void_p), stride, self.threads), "az_rng_dirichlet")
        return out

    def random_sample(self, mask: np.ndarray | None = None) -> np.ndarray:
        out = np.zeros(self.count, np.float64)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        N.check(self.L.az_rng_random_sample(self._h, None if m is None else m.ctypes.data_as(C.c_void_p),
                                            out.ctypes.data_as(C.c_void_p)), "az_rng_random_sample")
        return out


def policy_from_visits(visits: np.ndarray, temps: np.ndarray) -> np.ndarray:
    """mcts.py:270-279 for a batch: visits f64 [G,A], one temperature per row -> pi f64 [G,A]."""
    pi = np.zeros_like(visits)
    for t in np.unique(temps):
        rows = np.nonzero(temps == t)[0]
        v = visits[rows]
        if t <= 1e-2:
            p = np.zeros_like(v)
            p[np.arange(len(rows)), np.argmax(v, axis=1)] = 1.0
        else:
            p = np.power(v, 1.0 / t)
            s = p.sum(axis=1)      # contiguous rows: the same pairwise order as the 1-D sums upstream takes
            bad = s <= 0
            p = np.where(bad[:, None], 1.0 / v.shape[1], p / np.where(bad, 1.0, s)[:, None])
        pi[rows] = p
    return pi


def sample_actions(pi: np.ndarray, u: np.ndarray) -> np.ndarray:
    """RandomState.choice(len(p), p=p) given its uniform draw u: cdf = cumsum(p); cdf /= cdf[-1];
    index = searchsorted(cdf, u, side='right')."""
    cdf = np.cumsum(pi, axis=1)
    cdf /= cdf[:, -1:]
    return (cdf <= u[:, None]).sum(axis=1).astype(np.int32)


# Streams for the lanes of this process, per device, each pair of them checked to run kernels at the same time
# (az_streams_overlap): HIP maps streams onto a few hardware queues, and two lanes whose streams share a queue take turns
# instead of overlapping. Reused by every BatchedSelfPlay of the process, so that a later driver does not land on a worse
# pair than the first one.
_LANE_STREAMS: dict = {}


def lane_streams(dev, count: int, tries: int = 12):
    """`count` streams on `dev` for the lanes of a driver: mutually overlapping ones as far as the hardware queues allow
    (further lanes share the streams found, round-robin)."""
    have = _LANE_STREAMS.setdefault(str(dev), [])
    L = N.lib()
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    for _ in range(tries):
        if len(have) >= count:
            break
        cand = torch.cuda.Stream(device=dev)
        ok = True
        for other in have:
            flag = C.c_int32(0)
            N.check(L.az_streams_overlap(C.c_void_p(cand.cuda_stream), C.c_void_p(other.cuda_stream), 150, index, C.byref(flag)), "az_streams_overlap")
            if not flag.value:
                ok = False
                break
        if ok:
            have.append(cand)
        else:
            _LANE_STREAMS.setdefault(str(dev) + ":shared", []).append(cand)   # kept alive: torch hands out the next stream of its pool
    return [have[i % len(have)] for i in range(count)]


class _Lane:
    """A share of the lockstep games with its own engine, network handle, HIP stream and random streams.
    launch() enqueues one move's search without waiting for its batches; finish() collects it, picks the
    moves and plays them. While one lane's host work runs, the other lanes' kernels keep the GPU busy."""

    def __init__(self, owner, index, games, model, precision, dedup_leaves, rng_threads, own_stream, stream=None):
        cfg = owner.cfg
        self.o, self.index, self.G = owner, index, int(games)
        self.dev = owner.dev
        if stream is not None:
            self.stream = stream
        else:
            self.stream = torch.cuda.Stream(device=self.dev) if own_stream else torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(self.stream):
            self.eng = Engine(board_size=cfg.board_size, history_len=cfg.history_len, games=self.G, komi=cfg.komi,
                              num_simulations=cfg.mcts_simulations, c_base=cfg.c_base, c_init=cfg.c_init,
                              dirichlet_alpha=cfg.dirichlet_alpha, dirichlet_eps=cfg.dirichlet_eps,
                              inference_batch_size=cfg.mcts_inference_batch, topk_expand=cfg.mcts_topk_expand,
                              virtual_loss=cfg.mcts_virtual_loss, dedup_leaves=dedup_leaves, device=self.dev)
            if hasattr(model, "native"):
                if precision == "bf16":      # shapes the MFMA kernels do not cover run on the float32 kernels
                    self.net = model.native_auto(self.dev, shared=not own_stream)
                    precision = self.net.precision
                else:
                    self.net = model.native(precision, self.dev, shared=not own_stream)
                self.layout = self.net.layout
                self.eval = self.net.forward
            else:  # plain callable on device tensors: feats f32 NCHW -> (logits, values)
                self.net, self.layout, self.eval = None, N.AZ_FEAT_F32_NCHW, model
            self.rng = HostRng(self.G, rng_threads)
            self.A = self.eng.A
            self.max_rows = self.eng.max_rows
            if self.layout == N.AZ_FEAT_BF16_NHWC:
                self.feat_buf = torch.empty((self.max_rows, cfg.board_size ** 2, self.eng.cpad), dtype=torch.bfloat16, device=self.dev)
            else:
                self.feat_buf = torch.empty((self.max_rows, self.eng.Cp, cfg.board_size, cfg.board_size), dtype=torch.float32, device=self.dev)
            # the native bf16 network gives a position the same output in any batch, so the logits of the child that is
            # played can stand in for the evaluation of the next root
            self.root_reuse = self.net is not None and precision == "bf16" and owner.reuse_roots
            if self.root_reuse:
                self.eng.set_root_reuse(True)
            if self.net is not None and precision == "bf16":
                self.net.reserve(self.max_rows)      # no allocation inside the move loop
            # with the native bf16 network the batches of a search are enqueued without host round trips, unless the
            # layer-by-layer path cannot hold the upper bound in its activation arrays (2 GiB each at most) and would pay
            # for empty passes over it (then: one round trip per batch)
            self.can_async = self.net is not None and precision == "bf16" and self.net.indirect
            self.async_search = self.can_async and self.net.indirect_passes(self.max_rows) == 1
            if self.async_search:
                self.eng.set_async(True)
                self.rows_dev = self.eng.rows_device()
                self.logits_buf = torch.empty((self.max_rows, self.A), dtype=torch.float32, device=self.dev)
                self.values_buf = torch.empty((self.max_rows,), dtype=torch.float32, device=self.dev)
            self.visits_h = torch.empty((self.G, self.A), dtype=torch.int32, pin_memory=True)
            self.rv_h = torch.empty((self.G,), dtype=torch.float64, pin_memory=True)
            self.info_np = None
            self.step_no = 0
        self.inflight = None
        self.nl = None               # native lane (az_lane_*): created at begin() when the search runs without round trips
        self.native = False
        self.move_idx = np.zeros(self.G, np.int64)

    def __del__(self):
        self._drop_native()

    def _drop_native(self):
        if getattr(self, "nl", None):
            N.lib().az_lane_destroy(self.nl)
            self.nl = None
        self.native = False

    def _make_native(self):
        """The native lane: one enqueue and one device-to-host copy per move (az_lane_*)."""
        cfg, L = self.o.cfg, N.lib()
        # visits ** (1 / T) for every possible visit count, from NumPy's own pow: the bits upstream's np.power gives on
        # this host (mcts.py:274), whatever libm or the device would return
        plen = cfg.mcts_simulations + 2
        v = np.arange(plen, dtype=np.float64)
        tabs = np.zeros((3, plen), np.float64)
        for i, t in enumerate((cfg.temp_start, cfg.temp_mid, cfg.temp_end)):
            if t > 1e-2:
                tabs[i] = np.power(v, 1.0 / t)
        lc = N.AzLaneConfig(int(cfg.temp_moves), int(cfg.temp_mid_moves), float(cfg.temp_start), float(cfg.temp_mid), float(cfg.temp_end),
                            float(cfg.resign_threshold), 1 if cfg.disable_resign else 0, 12, float(cfg.dirichlet_alpha),
                            1 if self.o.collect else 0, int(self.rng.threads), 1 if self.o.graph_rounds else 0)
        h = C.c_void_p()
        with torch.cuda.device(self.dev):
            N.check(L.az_lane_create(self.eng._h, self.net._h, self.rng._h, C.byref(lc), tabs.ctypes.data_as(C.c_void_p), plen,
                                     C.c_void_p(self.feat_buf.data_ptr()), int(self.layout), C.c_void_p(self.logits_buf.data_ptr()),
                                     C.c_void_p(self.values_buf.data_ptr()), C.byref(h)), "az_lane_create")
        self.nl = h
        p_on, p_mi = C.c_void_p(), C.c_void_p()
        N.check(L.az_lane_slots(h, C.byref(p_on), C.byref(p_mi)), "az_lane_slots")
        self.slot_on = np.ctypeslib.as_array(C.cast(p_on, C.POINTER(C.c_uint8)), shape=(self.G,))
        self.nl_move_idx = np.ctypeslib.as_array(C.cast(p_mi, C.POINTER(C.c_int32)), shape=(self.G,))
        n, H = cfg.board_size, cfg.history_len
        self.row_bytes = ((2 * H + 3) * n * n + 7) // 8
        self.over_slot = np.zeros(self.G, np.int32)
        self.over_winner = np.zeros(self.G, np.int32)
        self.actions_np = np.zeros(self.G, np.int32)
        self.rv_np = np.zeros(self.G, np.float64)

    def _stream_ptr(self):
        return C.c_void_p(self.stream.cuda_stream)

    # ------------------------------------------------------------------ slots
    def begin(self):
        G = self.G
        self.game_id = np.full(G, -1, np.int64)
        want_native = NATIVE_LANE and self.async_search and self.net is not None
        if want_native and self.nl is None:
            self._make_native()
        elif not want_native:
            self._drop_native()
        self.native = want_native
        self.move_idx = self.nl_move_idx if self.native else np.zeros(G, np.int64)
        self.move_idx[:] = 0
        with torch.cuda.stream(self.stream):
            self.eng.reset()
        for s in range(G):
            gid = self.o._deal()
            if gid < 0:
                break
            self.game_id[s] = gid
            self.rng.seed(s, self.o.seed0 + gid)
        self.log = []                        # per step: (packed planes [G,*] u8, capf [G,2], pi f32 [G,A], to_play [G], difficulty [G])
        self.log_base = self.step_no         # step index of log[0]
        self.start_step = np.full(G, self.step_no, np.int64)   # first step of the game in each slot
        self.inflight = None
        self.info_np = None
        self._open_games(np.nonzero(self.game_id >= 0)[0])
        if self.native:
            self.slot_on[:] = self.game_id >= 0
            with torch.cuda.stream(self.stream):
                N.check(N.lib().az_lane_prepare(self.nl, self._stream_ptr()), "az_lane_prepare")

    def _open_games(self, slots):
        """Opening-book moves of the games just dealt to `slots` (train.py:291-311). A game's book draws come
        from random.Random(seed0 + id): what `random` yields after upstream's set_seed(seed0 + id)."""
        cfg = self.o.cfg
        if cfg.opening_book_prob <= 0.0 or cfg.opening_book_max_moves <= 0 or len(slots) == 0:
            return
        n = cfg.board_size
        lines = {int(s): choose_opening(n, cfg, random.Random(self.o.seed0 + int(self.game_id[s]))) for s in slots}
        live = {s for s, ln in lines.items() if ln}
        with torch.cuda.stream(self.stream):
            for k in range(max((len(ln) for ln in lines.values()), default=0)):
                actions = np.full(self.G, -1, np.int32)
                for s in list(live):
                    if k < len(lines[s]):
                        actions[s] = lines[s][k][0] * n + lines[s][k][1]
                    else:
                        live.discard(s)
                if not live:
                    break
                st = self.eng.play(torch.from_numpy(actions), raise_on_illegal=False).cpu().numpy()
                for s in list(live):
                    if st[s] != 0:
                        live.discard(s)      # an illegal book move ends the line without being played
                    else:
                        self.move_idx[s] += 1

    def busy(self) -> bool:
        return self.inflight is not None or bool((self.game_id >= 0).any())

    # ------------------------------------------------------------------ one move, first half: enqueue the search
    def launch(self):
        active = self.game_id >= 0
        if not active.any():
            return
        if self.native:      # the whole move in one native call: random draws, one upload, every kernel, one download
            refill = 1 if self.o.next_id < self.o.num_games else 0
            with torch.cuda.stream(self.stream):
                if N.check(N.lib().az_lane_launch(self.nl, refill, self._stream_ptr()), "az_lane_launch") == 1:
                    self.o.counters["graph_replays"] += 1
            self.inflight = (active,)
            return
        eng, c, cfg = self.eng, self.o.counters, self.o.cfg
        with torch.cuda.stream(self.stream):
            if self.info_np is None:       # the first move, or games were re-dealt: finish() otherwise leaves the status here
                self.info_np = eng.status()[1].cpu().numpy()
            to_play = self.info_np[:, 0].copy()
            packed = self._pack_planes(eng.features(N.AZ_FEAT_F32_NCHW)) if self.o.collect else None
            eng.search_begin(torch.from_numpy(active.astype(np.uint8)))
            if self.async_search:
                # root rows: gathered and evaluated over their upper bound, so the one round trip of the root phase (the
                # legal-move counts the noise needs) starts behind the network instead of in front of it
                eng.gather(self.max_rows, self.layout, out=self.feat_buf)
                self.net.forward_indirect(self.feat_buf, self.rows_dev, self.max_rows, self.logits_buf, self.values_buf)
                counts = eng.root_legal_counts()
                rows = eng.pending()
                logits, values = self.logits_buf, self.values_buf
            else:
                rows = eng.pending()
                if eng.gather_rows() == 0:
                    raise N.NativeError("search with no root")
                feats = eng.gather(rows, self.layout, out=self.feat_buf)
                logits, values = self.eval(feats[:rows]) if rows > 0 else (None, None)   # rows == 0: every root's logits were carried over
                counts = eng.root_legal_counts()
            counts = counts.cpu().numpy()
            t0 = time.perf_counter()
            noise = self.rng.dirichlet(counts, cfg.dirichlet_alpha, self.A)
            c["t_host_rng"] += time.perf_counter() - t0
            done = eng.expand(logits, values, torch.from_numpy(noise))
            c["nn_rows"] += rows
            c["nn_rows_nodedup"] += rows
            if not done and self.async_search and NATIVE_ROUNDS:   # every remaining round in one native call (a graph replay where possible)
                if eng.run_rounds(self.net, self.feat_buf, self.layout, self.logits_buf, self.values_buf, graph=self.o.graph_rounds):
                    c["graph_replays"] += 1
                done = True
            while not done and self.async_search:
                eng.search_select()
                eng.gather(self.max_rows, self.layout, out=self.feat_buf)
                self.net.forward_indirect(self.feat_buf, self.rows_dev, self.max_rows, self.logits_buf, self.values_buf)
                done = eng.expand(self.logits_buf, self.values_buf)
            while not done:
                eng.search_select()
                rows = eng.pending()
                if rows > 0:
                    feats = eng.gather(rows, self.layout, out=self.feat_buf)
                    logits, values = self.eval(feats[:rows])
                    done = eng.expand(logits, values)
                    c["nn_rows"] += rows
                else:
                    done = eng.expand(None, None)
            visits, rv = eng.result()
            self.visits_h.copy_(visits, non_blocking=True)      # on the host by the time finish() has waited for the search
            self.rv_h.copy_(rv, non_blocking=True)
        self.inflight = (active, to_play, packed, visits, rv)

    # ------------------------------------------------------------------ one move, second half (train.py:322-343)
    def _finish_native(self):
        o, G, A, c = self.o, self.G, self.A, self.o.counters
        t0 = time.perf_counter()
        r = N.AzLaneResult()
        r.over_slot, r.over_winner = self.over_slot.ctypes.data, self.over_winner.ctypes.data
        r.actions, r.root_value = self.actions_np.ctypes.data, self.rv_np.ctypes.data
        if o.collect:
            bits, caps = np.empty((G, self.row_bytes), np.uint8), np.empty((G, 2), np.float32)
            pi32, to_play, diff = np.empty((G, A), np.float32), np.empty(G, np.int32), np.empty(G, np.float64)
            r.bits, r.caps, r.pi32, r.to_play, r.difficulty = bits.ctypes.data, caps.ctypes.data, pi32.ctypes.data, to_play.ctypes.data, diff.ctypes.data
        N.check(N.lib().az_lane_finish(self.nl, C.byref(r)), "az_lane_finish")
        if o.collect:
            self.log.append((bits, caps, pi32, to_play, diff))
        k = r.counters
        c["sims"] += k[0]; c["nn_rows"] += k[1]; c["nn_rows_nodedup"] += k[2]; c["depth_sum"] += k[3]; c["moves"] += k[7]
        c["t_host_rng"] += r.host_seconds_rng
        if r.n_over:
            fresh = []
            for s, winner in zip(self.over_slot[: r.n_over].tolist(), self.over_winner[: r.n_over].tolist()):
                o.results[0 if winner == 1 else (1 if winner == -1 else 2)] += 1
                o.finished[int(self.game_id[s])] = (self.index, s, int(self.start_step[s]), self.step_no + 1, winner)
                self.start_step[s] = self.step_no + 1
                c["games"] += 1
                gid = o._deal()
                self.game_id[s] = gid
                if gid >= 0:
                    self.rng.seed(s, o.seed0 + gid)
                    fresh.append(s)
                else:
                    self.slot_on[s] = 0
            cfg = o.cfg
            if fresh and cfg.opening_book_prob > 0.0 and cfg.opening_book_max_moves > 0:
                # book moves are played from the host: the roots the move's tail prepared for these slots are stale
                with torch.cuda.stream(self.stream):
                    self._open_games(np.array(fresh))
                    N.check(N.lib().az_lane_prepare(self.nl, self._stream_ptr()), "az_lane_prepare")
        c["t_host_policy"] += time.perf_counter() - t0
        self.step_no += 1

    def finish(self):
        if self.inflight is None:
            return
        if self.native:
            self.inflight = None
            return self._finish_native()
        active, to_play, packed, visits, rv = self.inflight
        self.inflight = None
        o, cfg, G, eng, A = self.o, self.o.cfg, self.G, self.eng, self.A
        c = o.counters
        game_id, move_idx = self.game_id, self.move_idx
        with torch.cuda.stream(self.stream):
            st = eng.stats()
            if self.async_search:
                c["nn_rows"] += st["rows"]
            c["sims"] += st["sims"]
            c["nn_rows_nodedup"] += st["rows_raw"]
            c["depth_sum"] += st["depth_sum"]
            visits, rv = self.visits_h.numpy(), self.rv_h.numpy()   # az_mcts_stats has synchronised the stream
            t0 = time.perf_counter()
            temps = np.where(move_idx < cfg.temp_moves, cfg.temp_start,
                             np.where(move_idx < cfg.temp_mid_moves, cfg.temp_mid, cfg.temp_end)).astype(np.float64)
            act_rows = np.nonzero(active)[0]
            pi = np.zeros((G, A), np.float64)
            pi[act_rows] = policy_from_visits(visits[act_rows].astype(np.float64), temps[act_rows])
            u = self.rng.random_sample(active)
            actions = np.full(G, -1, np.int32)
            actions[act_rows] = sample_actions(pi[act_rows], u[act_rows])
            pi32 = pi.astype(np.float32)
            c["t_host_policy"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            if o.collect:
                bits, caps = packed
                diff = np.maximum(0.01, 1.0 - pi32.max(axis=1).astype(np.float64))      # train.py:336-337
                self.log.append((bits.cpu().numpy(), caps.cpu().numpy(), pi32, to_play, diff))
            c["t_samples"] += time.perf_counter() - t0
            # resignation (train.py:339-341) ends the game before the move is played
            resigned = np.zeros(G, np.int64)
            if not cfg.disable_resign:
                rs = active & (np.abs(rv) >= cfg.resign_threshold) & (move_idx >= 12)
                resigned[rs] = np.where(rv[rs] < 0, -to_play[rs], to_play[rs])
            actions[resigned != 0] = -1
            played = eng.play(torch.from_numpy(actions), raise_on_illegal=False)
            move_idx[active & (resigned == 0)] += 1
            c["moves"] += int(active.sum())
            _, info = eng.status()
            info = torch.cat((info, played.view(-1, 1)), dim=1).cpu().numpy()      # one round trip for both
            if info[:, 8].any():
                g = int(np.nonzero(info[:, 8])[0][0])
                raise ValueError(f"Illegal move: {divmod(int(actions[g]), cfg.board_size)}")
            self.info_np = info = info[:, :8]
            over = active & ((info[:, 6] != 0) | (resigned != 0))
            if over.any():
                reset = np.zeros(G, np.uint8)
                for s in np.nonzero(over)[0]:
                    winner = int(resigned[s]) if resigned[s] != 0 else int(info[s, 7])
                    o.results[0 if winner == 1 else (1 if winner == -1 else 2)] += 1
                    o.finished[int(game_id[s])] = (self.index, int(s), int(self.start_step[s]), self.step_no + 1, winner)
                    self.start_step[s] = self.step_no + 1
                    c["games"] += 1
                    move_idx[s] = 0
                    gid = o._deal()
                    game_id[s] = gid
                    if gid >= 0:
                        self.rng.seed(s, o.seed0 + gid)
                        reset[s] = 1
                self.info_np = None      # finished games keep their slot's status until re-dealt; launch() reads it again
                if reset.any():
                    eng.reset(torch.from_numpy(reset))
                    self._open_games(np.nonzero(reset)[0])
        self.step_no += 1

    def _pack_planes(self, planes):
        """Binary planes of to_tensor() (go.py:265-304: all but the two capture planes) as np.packbits would
        pack them (row-major, most significant bit first), packed on the device (az_pack_planes) before the
        copy to the host; the capture planes as one float each."""
        G, H, n = self.G, self.o.cfg.history_len, self.o.cfg.board_size
        row_bytes = ((2 * H + 3) * n * n + 7) // 8
        bits = torch.empty((G, row_bytes), dtype=torch.uint8, device=self.dev)
        caps = torch.empty((G, 2), dtype=torch.float32, device=self.dev)
        N.check(N.lib().az_pack_planes(C.c_void_p(planes.data_ptr()), G, n, H, row_bytes, C.c_void_p(bits.data_ptr()),
                                       C.c_void_p(caps.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                "az_pack_planes")
        return bits, caps

    def prune_log(self):
        running = self.game_id >= 0
        keep_from = int(self.start_step[running].min()) if running.any() else self.step_no
        drop = max(0, keep_from - self.log_base)
        del self.log[:drop]
        self.log_base += drop


class BatchedSelfPlay:
    """`games` lockstep self-play games on one GPU, split over `lanes` independent lanes (engine + network
    handle + HIP stream each). With one lane, step() plays one move of every game and returns. With more,
    step() is software-pipelined: for each lane in turn it finishes the move enqueued by the previous step()
    and enqueues the next search, so the lanes' kernels overlap each other and the host work; flush()
    completes what is in flight. A game's trajectory depends only on its id (seed0 + id seeds its random
    stream), not on the lane or slot that plays it."""

    def __init__(self, model, cfg: SelfPlayConfig, games: int, seed0: int = 0, dedup_leaves: bool = True,
                 precision: str = "bf16", collect_samples: bool = True, device=None, rng_threads: int = 8,
                 lanes: int = 1, tower_workgroups: int | None = None, reuse_roots: bool = True, graph_rounds: bool | None = None,
                 streams=None):
        self.cfg, self.G, self.seed0 = cfg, int(games), int(seed0)
        self.graph_rounds = GRAPH_ROUNDS if graph_rounds is None else bool(graph_rounds)   # replay a move's rounds from a HIP graph
        self.reuse_roots = bool(reuse_roots)
        self.collect = collect_samples
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        lanes = max(1, min(int(lanes), self.G))
        self.counters = dict(moves=0, sims=0, nn_rows=0, nn_rows_nodedup=0, games=0, steps=0, depth_sum=0, graph_replays=0,
                             t_host_rng=0.0, t_host_policy=0.0, t_samples=0.0)
        share = [self.G // lanes + (1 if i < self.G % lanes else 0) for i in range(lanes)]
        if streams is None and lanes > 1:
            streams = lane_streams(self.dev, lanes)
        self.lanes = [_Lane(self, i, g, model, precision, dedup_leaves, rng_threads, own_stream=lanes > 1, stream=streams[i] if streams else None)
                      for i, g in enumerate(share)]
        # The low-latency tower variant takes the whole device per launch (one workgroup per compute unit): right for one
        # lane (256 games: 8.7 ms per move against 11.0), wrong for three or more sharing the card (768 games in 3 lanes:
        # 19.2 ms against 15.7). Two lanes: only where a lane's rows (about one per game) can be within that variant's limit
        # of one board per compute unit; otherwise the variant's gated-off launch would still queue behind the other lane's
        # workgroups for a compute unit of its own before it can leave (1024 games in 2 lanes: +4 % per move).
        cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
        if lanes > 2 or (lanes == 2 and max(share) > cus):
            for ln in self.lanes:
                if ln.net is not None and ln.net.precision == "bf16":
                    ln.net.set_device_choice(False)
        if tower_workgroups is None and lanes > 2:
            # Workgroups per tower launch. One or two lanes: the library's default (as many as the card holds at once, each
            # walking its share of the boards; for the two-per-CU kernels that is two per compute unit). More, shorter-lived
            # workgroups per launch (1024, the setting while four lanes were the default) let the lanes' launches interleave at
            # a finer grain and cost more than that gains: 4096 games in two lanes, same box, 16.2 M simulations/s with the
            # default against 15.5 M with 1024 and 15.4 M with 192 (a count the eight XCDs do not share evenly). Three or more
            # lanes: half the default, so that two launches fit side by side (four lanes: 16.0 M against 15.4 M).
            tower_workgroups = max(1, cus // 2)
        if tower_workgroups is not None:
            for ln in self.lanes:
                if ln.net is not None and ln.net.precision == "bf16":
                    ln.net.set_tower_workgroups(tower_workgroups)
        self.A = self.lanes[0].A
        self.num_games, self.next_id = 0, 0
        self.finished, self.results = {}, [0, 0, 0]

    # single-lane conveniences
    eng = property(lambda self: self.lanes[0].eng)
    net = property(lambda self: self.lanes[0].net)
    log = property(lambda self: self.lanes[0].log)
    async_search = property(lambda self: all(ln.async_search for ln in self.lanes))

    def set_async(self, enable: bool):
        """Switch between enqueueing the batches of a search without host round trips and one round trip per batch."""
        for ln in self.lanes:
            ok = bool(enable) and ln.can_async
            if ok and not hasattr(ln, "logits_buf"):
                with torch.cuda.stream(ln.stream):
                    ln.rows_dev = ln.eng.rows_device()
                    ln.logits_buf = torch.empty((ln.max_rows, ln.A), dtype=torch.float32, device=ln.dev)
                    ln.values_buf = torch.empty((ln.max_rows,), dtype=torch.float32, device=ln.dev)
            ln.async_search = ok
            ln.eng.set_async(ok)

    def _deal(self) -> int:
        if self.next_id >= self.num_games:
            return -1
        self.next_id += 1
        return self.next_id - 1

    # ------------------------------------------------------------------ driver
    def begin(self, num_games: int | None = None):
        """Reset every slot to a new game; game ids 0..num_games-1 are dealt to slots as they free up."""
        self.num_games = self.G if num_games is None else int(num_games)
        self.next_id = 0
        self.finished = {}                   # game id -> (lane, slot, first step, last step + 1, winner)
        self.results = [0, 0, 0]             # black wins, white wins, draws
        for ln in self.lanes:
            ln.begin()

    def busy(self) -> bool:
        return any(ln.busy() for ln in self.lanes)

    def step(self):
        """One lockstep move of every running game (one selfplay_game loop iteration each, train.py:322-343)."""
        if len(self.lanes) == 1:
            self.lanes[0].launch()
            self.lanes[0].finish()
        else:
            for ln in self.lanes:
                ln.finish()
                ln.launch()
        self.counters["steps"] += 1

    def flush(self):
        for ln in self.lanes:
            ln.finish()

    def run(self, num_games: int | None = None, max_steps: int | None = None):
        """Play `num_games` games (default: one per slot), or stop after `max_steps` lockstep moves."""
        self.begin(num_games)
        steps = 0
        while self.busy() and (max_steps is None or steps < max_steps):
            self.step()
            steps += 1
        self.flush()
        return self.collect_samples()

    def profile(self, enable: bool, read: bool = False):
        """Tower-kernel timing over all lanes. read=True returns (sum of launch ms, launches, merged busy ms):
        the last is the length of the union of the launch intervals, i.e. launches that overlap on different
        streams are not counted twice."""
        nets = [ln.net for ln in self.lanes if ln.net is not None and ln.layout == N.AZ_FEAT_BF16_NHWC]
        out = None
        if read:
            spans = [n.profile_spans() for n in nets]
            spans = np.concatenate(spans) if spans else np.zeros((0, 2))
            total, launches = 0.0, 0
            for n in nets:
                ms, ln_, _ = n.profile(enable, read=True)
                total += ms
                launches += ln_
            busy, end = 0.0, -np.inf
            for a, b in spans[np.argsort(spans[:, 0])]:
                if b > end:
                    busy += b - max(a, end)
                    end = b
            out = (total, launches, busy)
        else:
            for n in nets:
                n.profile(enable)
        return out

    def trajectory(self, gid: int):
        """(list of (packed planes, capf, pi f32, to_play, difficulty) per move, winner) of a finished game."""
        lane, slot, a, b, winner = self.finished[gid]
        ln = self.lanes[lane]
        out = []
        for k in range(a - ln.log_base, b - ln.log_base):
            bits, caps, pi32, to_play, diff = ln.log[k]
            out.append((bits[slot], caps[slot], pi32[slot], int(to_play[slot]), float(diff[slot])))
        return out, winner

    def collect_samples(self):
        """Samples of the games finished so far, in game-id order (train.py:346-353)."""
        n, H = self.cfg.board_size, self.cfg.history_len
        nbin = (2 * H + 3) * n * n
        samples = []
        for gid in sorted(self.finished):
            tr, winner = self.trajectory(gid)
            for packed, capf, p, player, difficulty in tr:
                z = 0.0 if winner == 0 else (1.0 if winner == player else -1.0)
                samples.append((self._unpack(packed, capf, nbin), p, z, max(difficulty, abs(z) * 0.25)))
        bw, ww, dd = self.results
        return samples, int(bw), int(ww), int(dd)

    def drain_samples(self):
        """collect_samples(), then forget the finished games and the log entries no running game needs."""
        out = self.collect_samples()
        self.finished.clear()
        for ln in self.lanes:
            ln.prune_log()
        return out

    def _unpack(self, packed, capf, nbin):
        n, H = self.cfg.board_size, self.cfg.history_len
        t = np.zeros((2 * H + 5, n, n), np.float32)
        t[: 2 * H + 3] = np.unpackbits(packed)[:nbin].reshape(2 * H + 3, n, n)
        t[2 * H + 3], t[2 * H + 4] = capf[0], capf[1]
        return t
