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
import time
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as N
from .engine import Engine


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

    @classmethod
    def from_train_config(cls, cfg) -> "SelfPlayConfig":
        return cls(**{k: getattr(cfg, k) for k in cls.__dataclass_fields__ if hasattr(cfg, k)})


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
            s = np.array([row.sum() for row in p])      # 1-D sums: NumPy's pairwise order, as upstream
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


class BatchedSelfPlay:
    def __init__(self, model, cfg: SelfPlayConfig, games: int, seed0: int = 0, dedup_leaves: bool = True,
                 precision: str = "bf16", collect_samples: bool = True, device=None, rng_threads: int = 8):
        self.cfg, self.G, self.seed0 = cfg, int(games), int(seed0)
        self.collect = collect_samples
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.eng = Engine(board_size=cfg.board_size, history_len=cfg.history_len, games=self.G, komi=cfg.komi,
                          num_simulations=cfg.mcts_simulations, c_base=cfg.c_base, c_init=cfg.c_init,
                          dirichlet_alpha=cfg.dirichlet_alpha, dirichlet_eps=cfg.dirichlet_eps,
                          inference_batch_size=cfg.mcts_inference_batch, topk_expand=cfg.mcts_topk_expand,
                          virtual_loss=cfg.mcts_virtual_loss, dedup_leaves=dedup_leaves, device=self.dev)
        if hasattr(model, "native"):
            self.net = model.native(precision, self.dev)
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
        self.counters = dict(moves=0, sims=0, nn_rows=0, nn_rows_nodedup=0, games=0, steps=0, depth_sum=0,
                             t_host_rng=0.0, t_host_policy=0.0, t_samples=0.0)

    # ------------------------------------------------------------------ one lockstep search
    def _search(self, enable: np.ndarray):
        eng, c = self.eng, self.counters
        eng.search_begin(torch.from_numpy(enable.astype(np.uint8)))
        rows = eng.pending()
        if rows == 0:
            return None, None
        feats = eng.gather(rows, self.layout, out=self.feat_buf)
        logits, values = self.eval(feats[:rows])
        counts = eng.root_legal_counts().cpu().numpy()
        t0 = time.perf_counter()
        noise = self.rng.dirichlet(counts, self.cfg.dirichlet_alpha, self.A)
        c["t_host_rng"] += time.perf_counter() - t0
        done = eng.expand(logits, values, torch.from_numpy(noise))
        c["nn_rows"] += rows
        c["nn_rows_nodedup"] += rows
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
        st = eng.stats()
        c["sims"] += st["sims"]
        c["nn_rows_nodedup"] += st["rows_raw"]
        c["depth_sum"] += st["depth_sum"]
        visits, rv = eng.result()
        return visits.cpu().numpy(), rv.cpu().numpy()

    # ------------------------------------------------------------------ driver
    def begin(self, num_games: int | None = None):
        """Reset every slot to a new game; game ids 0..num_games-1 are dealt to slots as they free up."""
        G = self.G
        self.num_games = G if num_games is None else int(num_games)
        self.game_id = np.full(G, -1, np.int64)
        self.move_idx = np.zeros(G, np.int64)
        self.next_id = 0
        self.eng.reset()
        for s in range(G):
            if self.next_id < self.num_games:
                self.game_id[s] = self.next_id
                self.rng.seed(s, self.seed0 + self.next_id)
                self.next_id += 1
        self.traj = [[] for _ in range(G)]   # per slot: (packed planes, capf, pi f32, to_play, difficulty)
        self.finished = {}                   # game id -> (trajectory, winner)
        self.results = [0, 0, 0]             # black wins, white wins, draws

    def busy(self) -> bool:
        return bool((self.game_id >= 0).any())

    def step(self):
        """One lockstep move of every running game (one selfplay_game loop iteration each, train.py:322-343)."""
        cfg, G, eng, c = self.cfg, self.G, self.eng, self.counters
        A, H, n = self.A, cfg.history_len, cfg.board_size
        game_id, move_idx = self.game_id, self.move_idx
        active = game_id >= 0
        _, info = eng.status()
        to_play = info.cpu().numpy()[:, 0].copy()
        planes = eng.features(N.AZ_FEAT_F32_NCHW) if self.collect else None
        visits, rv = self._search(active)
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
        if self.collect:
            pl = planes.cpu().numpy()
            bits = np.packbits(pl[:, : 2 * H + 3].reshape(G, -1) > 0.5, axis=1)
            caps = pl[:, 2 * H + 3:, 0, 0].copy()
            diff = np.maximum(0.01, 1.0 - pi32.max(axis=1).astype(np.float64))      # train.py:336-337
            for s in act_rows:
                self.traj[s].append((bits[s], caps[s], pi32[s].copy(), int(to_play[s]), float(diff[s])))
        c["t_samples"] += time.perf_counter() - t0
        # resignation (train.py:339-341) ends the game before the move is played
        resigned = np.zeros(G, np.int64)
        if not cfg.disable_resign:
            rs = active & (np.abs(rv) >= cfg.resign_threshold) & (move_idx >= 12)
            resigned[rs] = np.where(rv[rs] < 0, -to_play[rs], to_play[rs])
        actions[resigned != 0] = -1
        eng.play(torch.from_numpy(actions))
        move_idx[active & (resigned == 0)] += 1
        c["moves"] += int(active.sum())
        _, info = eng.status()
        info = info.cpu().numpy()
        over = active & ((info[:, 6] != 0) | (resigned != 0))
        if over.any():
            reset = np.zeros(G, np.uint8)
            for s in np.nonzero(over)[0]:
                winner = int(resigned[s]) if resigned[s] != 0 else int(info[s, 7])
                self.results[0 if winner == 1 else (1 if winner == -1 else 2)] += 1
                self.finished[int(game_id[s])] = (self.traj[s], winner)
                self.traj[s] = []
                c["games"] += 1
                move_idx[s] = 0
                if self.next_id < self.num_games:
                    game_id[s] = self.next_id
                    self.rng.seed(s, self.seed0 + self.next_id)
                    self.next_id += 1
                    reset[s] = 1
                else:
                    game_id[s] = -1
            if reset.any():
                eng.reset(torch.from_numpy(reset))
        c["steps"] += 1

    def run(self, num_games: int | None = None, max_steps: int | None = None):
        """Play `num_games` games (default: one per slot), or stop after `max_steps` lockstep moves."""
        self.begin(num_games)
        steps = 0
        while self.busy() and (max_steps is None or steps < max_steps):
            self.step()
            steps += 1
        return self.collect_samples()

    def collect_samples(self):
        n, H = self.cfg.board_size, self.cfg.history_len
        nbin = (2 * H + 3) * n * n
        samples = []
        for gid in sorted(self.finished):
            tr, winner = self.finished[gid]
            for packed, capf, p, player, difficulty in tr:
                z = 0.0 if winner == 0 else (1.0 if winner == player else -1.0)      # train.py:346-353
                samples.append((self._unpack(packed, capf, nbin), p, z, max(difficulty, abs(z) * 0.25)))
        bw, ww, dd = self.results
        return samples, int(bw), int(ww), int(dd)

    def _unpack(self, packed, capf, nbin):
        n, H = self.cfg.board_size, self.cfg.history_len
        t = np.zeros((2 * H + 5, n, n), np.float32)
        t[: 2 * H + 3] = np.unpackbits(packed)[:nbin].reshape(2 * H + 3, n, n)
        t[2 * H + 3], t[2 * H + 4] = capf[0], capf[1]
        return t
