"""Training-side callers of the hot path with the upstream names (go_zero/train.py): TrainConfig and
its CLI, replay buffer with prioritised sampling and dihedral augmentation, self-play, the training
step, arena matches with the sequential stopping rule, Elo / model pool / checkpoints, the main loop.

What differs from upstream is where the work runs, not what is computed:
  * run_selfplay plays all games of an iteration in lockstep on the GPU (alphazero_amd.selfplay) instead
    of a process pool of CPU workers; game j draws from the random streams a one-game upstream worker
    seeded `seed + iter*1000 + j` would draw from (train.py:446).
  * ReplayBuffer keeps a device mirror of its samples (bit-packed planes) and assembles training batches
    with a HIP kernel (az_replay_gather); indices and symmetries are still drawn from np.random / random
    exactly as ReplayBuffer.sample draws them (train.py:158-166), so a batch is the same batch.
  * arena games are played with arg-max moves and no noise (train.py:572), so a game is a function of
    (network pair, colour assignment): the two distinct games are played once, in lockstep, and the
    win/loss/draw tally and the stopping rule are evaluated over the game ids in upstream order.
  * under torch.distributed (one process per GPU) self-play games are sharded over the ranks, every rank
    trains on its own replay shard with gradient all-reduce, and rank 0 runs the arena and writes files.
"""
from __future__ import annotations

import argparse
import copy
import dataclasses
import json
import os
import random
from collections import deque

import numpy as np
import torch

from . import _native as N
from .go import BLACK, WHITE, GoState, decode_action
from .mcts import MCTS
from .model import AlphaZeroNet
from .selfplay import choose_opening, opening_book_lines
from .train_step import all_reduce_, broadcast_

# ----------------------------------------------------------------------------------------------- config
# (field, default, command line): "value" = --field-name VALUE, "flag" = --field-name sets True,
# "neg:X" = --X sets False, None = not on the command line (train.py:26-102, :1026-1169).
_SPEC = [
    ("board_size", 9, "value"), ("komi", 5.5, "value"), ("history_len", 8, "value"),
    ("device", "cpu", "value"), ("seed", 42, "value"),
    ("iterations", 50, "value"), ("selfplay_games_per_iter", 40, "value"), ("selfplay_workers", 1, "value"),
    ("async_selfplay", False, "flag"), ("train_torch_threads", 0, "value"), ("worker_torch_threads", 1, "value"),
    ("mcts_simulations", 192, "value"), ("eval_simulations", 256, "value"), ("mcts_inference_batch", 8, "value"),
    ("mcts_topk_expand", 0, "value"), ("mcts_virtual_loss", 1.0, "value"), ("mcts_root_parallelism", 1, "value"),
    ("temp_moves", 20, "value"), ("temp_mid_moves", 40, "value"), ("temp_start", 1.0, "value"), ("temp_mid", 0.4, "value"),
    ("temp_end", 0.05, "value"), ("resign_threshold", 0.98, "value"), ("disable_resign", False, "flag"),
    ("replay_size", 200000, "value"), ("batch_size", 256, "value"), ("train_steps_per_iter", 800, "value"),
    ("lr", 1e-3, "value"), ("weight_decay", 1e-4, "value"), ("grad_clip_norm", 1.0, "value"), ("use_amp", True, None),
    ("value_huber_delta", 1.0, "value"), ("policy_entropy_coeff", 1e-3, "value"), ("policy_target_smoothing", 0.0, "value"),
    ("channels", 192, "value"), ("num_blocks", 12, "value"), ("use_se", True, "neg:no-se"), ("value_hidden_dim", 256, "value"),
    ("arena_games", 40, "value"), ("promote_threshold", 0.55, "value"), ("arena_sprt", True, "neg:no-arena-sprt"),
    ("arena_sprt_delta", 0.05, "value"),
    ("augment_symmetry", True, "neg:no-augment-symmetry"),
    ("c_base", 19652.0, "value"), ("c_init", 1.25, "value"), ("dirichlet_alpha", 0.3, "value"), ("dirichlet_eps", 0.25, "value"),
    ("pool_max_size", 20, "value"), ("pool_opponents", 4, "value"), ("elo_games_per_opponent", 12, "value"),
    ("benchmark_opponents", 4, "value"), ("benchmark_games_per_opponent", 8, "value"), ("elo_k", 24.0, "value"),
    ("elo_init", 1500.0, "value"),
    ("async_prefetch_batches", 2, "value"), ("pbt", False, "flag"), ("pbt_patience", 3, "value"), ("pbt_sigma", 0.15, "value"),
    ("opening_book_prob", 0.0, "value"), ("opening_book_max_moves", 0, "value"),
    ("save_dir", "checkpoints", "value"), ("resume", "", "value"), ("save_buffer", False, "flag"),
]
# Knobs of the GPU engine that upstream has no counterpart for (kept out of upstream's field set on purpose).
_ENGINE_SPEC = [
    ("lockstep_games", 0, "value"),      # device game slots per GPU (0: as many as the iteration has games, at most 4096)
    ("selfplay_lanes", 0, "value"),      # independent lanes (0: 2 from 256 game slots on, else 1)
    ("no_dedup_leaves", False, "flag"),  # evaluate a leaf reached by several descents of a batch every time, as upstream does
]
UPSTREAM_FIELDS = tuple(n for n, _, _ in _SPEC)
ENGINE_FIELDS = tuple(n for n, _, _ in _ENGINE_SPEC)

TrainConfig = dataclasses.make_dataclass(
    "TrainConfig", [(n, type(d), dataclasses.field(default=d)) for n, d, _ in _SPEC + _ENGINE_SPEC])
TrainConfig.__doc__ = "train.py:26-102 field for field, plus the engine knobs in ENGINE_FIELDS."


def _parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="AlphaZero-style Go training (MI355X engine, go_zero.train command line)")
    for name, default, cli in _SPEC + _ENGINE_SPEC:
        flag = "--" + name.replace("_", "-")
        if cli == "value":
            p.add_argument(flag, type=type(default), default=default)
        elif cli == "flag":
            p.add_argument(flag, action="store_true")
        elif cli and cli.startswith("neg:"):
            p.add_argument("--" + cli[4:], action="store_true")
    p.add_argument("--use-amp", action="store_true")
    p.add_argument("--no-amp", action="store_true")
    return p


def parse_args(argv=None) -> TrainConfig:
    a = vars(_parser().parse_args(argv))
    kw = {}
    for name, _, cli in _SPEC + _ENGINE_SPEC:
        if cli in ("value", "flag"):
            kw[name] = a[name]
        elif cli and cli.startswith("neg:"):
            kw[name] = not a[cli[4:].replace("-", "_")]
    # train.py:1123, :1137, :1140, :1144, :1164-1165
    kw["use_amp"] = (not a["no_amp"]) and (a["use_amp"] or kw["device"].startswith("cuda"))
    kw["mcts_root_parallelism"] = max(1, kw["mcts_root_parallelism"])
    kw["policy_target_smoothing"] = max(0.0, kw["policy_target_smoothing"])
    kw["value_hidden_dim"] = max(1, kw["value_hidden_dim"])
    kw["opening_book_prob"] = float(np.clip(kw["opening_book_prob"], 0.0, 1.0))
    kw["opening_book_max_moves"] = max(0, kw["opening_book_max_moves"])
    return TrainConfig(**kw)


def config_from_dict(d: dict) -> TrainConfig:
    """A TrainConfig from a checkpoint's "cfg" dict (upstream's or ours): unknown keys are ignored."""
    names = {f.name for f in dataclasses.fields(TrainConfig)}
    return TrainConfig(**{k: v for k, v in d.items() if k in names})


def set_seed(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


# ----------------------------------------------------------------------------------------------- symmetries
_PERM_CACHE: dict[int, np.ndarray] = {}


def symmetry_permutations(n: int) -> np.ndarray:
    """int16 [8, n*n]: source point of every destination point under symmetry s, numbered as
    train.py:187-192 (s < 4: s quarter turns; s >= 4: mirror left-right, then s-4 quarter turns).
    Built by transforming an index grid, so it is the upstream transform by construction."""
    t = _PERM_CACHE.get(n)
    if t is None:
        grid = np.arange(n * n, dtype=np.int16).reshape(n, n)
        t = np.stack([np.rot90(grid if s < 4 else np.fliplr(grid), k=s % 4).reshape(-1) for s in range(8)])
        _PERM_CACHE[n] = t = np.ascontiguousarray(t)
    return t


def transform_2d(board: np.ndarray, sym: int) -> np.ndarray:
    n = board.shape[0]
    return board.reshape(-1)[symmetry_permutations(n)[sym]].reshape(n, n)


def apply_symmetry_to_tensor(state_tensor: np.ndarray, sym: int) -> np.ndarray:
    c, n = state_tensor.shape[0], state_tensor.shape[1]
    out = state_tensor.reshape(c, n * n)[:, symmetry_permutations(n)[sym]].reshape(c, n, n)
    return np.ascontiguousarray(out).astype(np.float32, copy=False)


def apply_symmetry_to_policy(policy: np.ndarray, board_size: int, sym: int) -> np.ndarray:
    np_ = board_size * board_size
    out = np.zeros_like(policy, dtype=np.float32)
    out[:np_] = policy[:np_][symmetry_permutations(board_size)[sym]]
    out[np_] = policy[np_]
    return out


# ----------------------------------------------------------------------------------------------- replay
class _DeviceMirror:
    """Replay samples in HBM in compact form (see az_replay_gather): ring of `capacity` rows that follows
    the host deque (row = insertion counter mod capacity)."""

    def __init__(self, capacity, n, hist, device):
        self.cap, self.n, self.hist, self.dev = capacity, n, hist, device
        self.nbin = 2 * hist + 3
        self.row_bytes = (self.nbin * n * n + 7) // 8
        self.A = n * n + 1
        self.bits = torch.zeros((capacity, self.row_bytes), dtype=torch.uint8, device=device)
        self.caps = torch.zeros((capacity, 2), dtype=torch.float32, device=device)
        self.pi = torch.zeros((capacity, self.A), dtype=torch.float32, device=device)
        self.z = torch.zeros((capacity,), dtype=torch.float32, device=device)
        self.perm = torch.from_numpy(symmetry_permutations(n)).to(device)

    def write(self, counters, samples):
        if not samples:
            return
        st = np.stack([s[0] for s in samples])
        planes = st[:, : self.nbin].reshape(len(samples), -1)
        binary = np.logical_or(planes == 0.0, planes == 1.0).all() and (st[:, self.nbin:] == st[:, self.nbin:, :1, :1]).all()
        if not binary:
            raise ValueError("replay samples are not to_tensor() planes (binary planes + constant capture planes)")
        rows = torch.from_numpy(np.asarray(counters, np.int64) % self.cap).to(self.dev)
        self.bits[rows] = torch.from_numpy(np.packbits(planes > 0.5, axis=1)).to(self.dev)
        self.caps[rows] = torch.from_numpy(np.ascontiguousarray(st[:, self.nbin:, 0, 0])).to(self.dev)
        self.pi[rows] = torch.from_numpy(np.stack([s[1] for s in samples]).astype(np.float32)).to(self.dev)
        self.z[rows] = torch.tensor([float(s[2]) for s in samples], dtype=torch.float32, device=self.dev)

    def gather(self, counters, syms):
        count = len(counters)
        idx = torch.from_numpy((np.asarray(counters, np.int64) % self.cap).astype(np.int32)).to(self.dev)
        sym = torch.from_numpy(np.asarray(syms, np.int32)).to(self.dev)
        states = torch.empty((count, self.nbin + 2, self.n, self.n), dtype=torch.float32, device=self.dev)
        pols = torch.empty((count, self.A), dtype=torch.float32, device=self.dev)
        import ctypes as C
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.dev):
            N.check(N.lib().az_replay_gather(p(self.bits), p(self.caps), p(self.pi), p(self.perm), p(idx), p(sym), count, self.n,
                                             self.hist, self.row_bytes, p(states), p(pols),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "az_replay_gather")
        return states, pols, self.z[idx.long()]


class ReplayBuffer:
    """train.py:105-176. `data` holds (state, pi, z, difficulty, counter) like upstream; sampling weights are
    difficulty x freshness (position in the buffer) x recency (insertion counter)."""

    def __init__(self, capacity: int):
        self.data = deque(maxlen=capacity)
        self.counter = 0
        self._probs = None
        self._snapshot = []
        self._mirror = None
        self._mirror_failed = False

    def __len__(self):
        return len(self.data)

    def extend(self, items):
        added = []
        for s, p, v, difficulty in items:
            self.data.append((s, p, v, float(difficulty), self.counter))
            added.append(self.counter)
            self.counter += 1
        self._probs = None
        if self._mirror is not None and added:
            cap = self._mirror.cap
            keep = added[-cap:]
            tail = list(self.data)[-len(keep):]
            self._mirror.write(keep, tail)

    def sampling_probabilities(self) -> np.ndarray:
        """train.py:122-145 over the current contents."""
        if self._probs is None or len(self._snapshot) != len(self.data):
            self._snapshot = list(self.data)
            n = len(self._snapshot)
            if n == 0:
                self._probs = None
                return None
            pos = np.arange(n, dtype=np.float64)
            latest = max(1.0, float(self._snapshot[-1][4]))
            freshness = 0.2 + 0.8 * ((pos + 1.0) / n)
            difficulty = np.array([max(1e-3, float(x[3])) for x in self._snapshot], dtype=np.float64)
            recency = np.array([0.5 + 0.5 * (float(x[4]) / latest) for x in self._snapshot], dtype=np.float64)
            w = difficulty * freshness * recency
            total = float(w.sum())
            self._probs = np.full(n, 1.0 / n, dtype=np.float64) if total <= 0.0 else w / total
        return self._probs

    def _draw(self, batch_size, augment_symmetry):
        """The random draws of train.py:158-166 in upstream order: one np.random.choice for the indices,
        then one random.randint(0, 7) per sample when augmenting."""
        if len(self.data) == 0:
            raise RuntimeError("ReplayBuffer is empty.")
        probs = self.sampling_probabilities()
        picks = np.random.choice(len(self._snapshot), size=batch_size, replace=True, p=probs)
        syms = [random.randint(0, 7) if augment_symmetry else 0 for _ in picks]
        return picks, syms

    def sample(self, batch_size: int, board_size: int, augment_symmetry: bool):
        picks, syms = self._draw(batch_size, augment_symmetry)
        states, policies, values = [], [], []
        for i, sym in zip(picks, syms):
            s, p, v, _, _ = self._snapshot[int(i)]
            if augment_symmetry:
                s, p = apply_symmetry_to_tensor(s, sym), apply_symmetry_to_policy(p, board_size, sym)
            states.append(s)
            policies.append(p)
            values.append(v)
        return (torch.from_numpy(np.stack(states, axis=0)).to(torch.float32),
                torch.from_numpy(np.stack(policies, axis=0)).to(torch.float32),
                torch.tensor(values, dtype=torch.float32))

    def attach_device(self, device, board_size: int, history_len: int):
        """Mirror the contents (and everything added later) in HBM for sample_device()."""
        dev = torch.device(device)
        if self._mirror is None or self._mirror.dev != dev:
            self._mirror = _DeviceMirror(self.data.maxlen, board_size, history_len, dev)
            items = list(self.data)
            self._mirror.write([x[4] for x in items], items)

    def sample_device(self, batch_size: int, board_size: int, augment_symmetry: bool):
        """sample() with the batch assembled on the GPU: same random draws, same tensors, no host stacking."""
        if self._mirror is None:
            raise RuntimeError("attach_device() first")
        picks, syms = self._draw(batch_size, augment_symmetry)
        counters = [self._snapshot[int(i)][4] for i in picks]
        return self._mirror.gather(counters, syms)


def position_key(state_tensor: np.ndarray) -> bytes:
    return state_tensor.tobytes()


def filter_selfplay_samples(samples, recent_keys: deque, recent_set: set, max_recent: int = 200000):
    """train.py:214-234: drop near-certain-pass targets and positions seen among the last max_recent kept ones."""
    kept = []
    for sample in samples:
        if float(sample[1][-1]) > 0.995:
            continue
        key = position_key(sample[0])
        if key in recent_set:
            continue
        recent_keys.append(key)
        recent_set.add(key)
        kept.append(sample)
        while len(recent_keys) > max_recent:
            recent_set.discard(recent_keys.popleft())
    return kept


# ----------------------------------------------------------------------------------------------- model / search
def build_model(cfg) -> AlphaZeroNet:
    return AlphaZeroNet(board_size=cfg.board_size, history_len=cfg.history_len, channels=cfg.channels,
                        num_blocks=cfg.num_blocks, use_se=cfg.use_se, value_hidden_dim=cfg.value_hidden_dim)


def create_mcts(model, cfg, sims: int, add_noise: bool) -> MCTS:
    del add_noise  # noise is a per-run argument (train.py:249)
    return MCTS(model=model, board_size=cfg.board_size, num_simulations=sims, c_base=cfg.c_base, c_init=cfg.c_init,
                dirichlet_alpha=cfg.dirichlet_alpha, dirichlet_eps=cfg.dirichlet_eps,
                inference_batch_size=cfg.mcts_inference_batch, topk_expand=cfg.mcts_topk_expand,
                virtual_loss=cfg.mcts_virtual_loss, root_parallelism=cfg.mcts_root_parallelism, device=cfg.device)


def maybe_play_opening_book(state: GoState, cfg):
    played = 0
    for move in choose_opening(state.board_size, cfg):
        if not state.is_legal(move):
            break
        state = state.play(move)
        played += 1
        if state.is_terminal():
            break
    return state, played


def selfplay_game(model, cfg):
    """One game, move after move, through the MCTS drop-in (train.py:314-354); random numbers from the global
    np.random / random streams like upstream. run_selfplay plays many of these at once instead."""
    state = GoState.new_game(board_size=cfg.board_size, komi=cfg.komi, history_len=cfg.history_len)
    state, move_idx = maybe_play_opening_book(state, cfg)
    mcts = create_mcts(model, cfg, sims=cfg.mcts_simulations, add_noise=True)
    trajectory, resigned = [], 0
    while not state.is_terminal():
        temp = cfg.temp_start if move_idx < cfg.temp_moves else (cfg.temp_mid if move_idx < cfg.temp_mid_moves else cfg.temp_end)
        pi, action, root_v = mcts.run(state, temperature=temp, add_exploration_noise=True, return_root_value=True)
        trajectory.append((state.to_tensor(), pi, state.to_play, max(0.01, 1.0 - float(np.max(pi)))))
        if not cfg.disable_resign and abs(root_v) >= cfg.resign_threshold and move_idx >= 12:
            resigned = -state.to_play if root_v < 0 else state.to_play
            break
        state = state.play(decode_action(cfg.board_size, action))
        move_idx += 1
    winner = resigned if resigned != 0 else state.winner()
    data = []
    for s, pi, player, difficulty in trajectory:
        z = 0.0 if winner == 0 else (1.0 if winner == player else -1.0)
        data.append((s, pi, z, max(difficulty, abs(z) * 0.25)))
    return data, winner


def _host_evaluator(model):
    """Evaluate a CPU model on device feature tensors (float32 NCHW): exact upstream numerics, for parity runs."""
    def ev(feats):
        with torch.inference_mode():
            lg, v = model(feats.detach().cpu())
        return lg.to(feats.device, torch.float32), v.to(feats.device, torch.float32)
    return ev


def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def rank_share(games: int, rank: int, world: int):
    """(first game id, game count) of `rank` when `games` games are dealt to `world` ranks in contiguous blocks."""
    first = rank * games // world
    return first, (rank + 1) * games // world - first


def selfplay_seed0(cfg, iter_idx: int) -> int:
    """Seed of game 0 of an iteration; game j uses this + j. Upstream gives worker w the seed seed + iter*1000 + w
    (train.py:446) with at most a few dozen workers; here j runs over all games of the iteration, so the stride between
    iterations grows with the game count: up to 1000 games the seeds are upstream's, beyond that the iterations' seed
    ranges still do not overlap (overlapping ranges would replay games bit for bit while the best model is unchanged)."""
    return cfg.seed + iter_idx * max(1000, int(cfg.selfplay_games_per_iter))


_warned_roots = False


def _warn_root_parallelism(cfg, where: str):
    """mcts_root_parallelism > 1 splits a search over several roots upstream (mcts.py:240-268); the lockstep engine
    searches one root per game."""
    global _warned_roots
    if getattr(cfg, "mcts_root_parallelism", 1) > 1 and not _warned_roots:
        _warned_roots = True
        import warnings
        warnings.warn(f"{where}: mcts_root_parallelism={cfg.mcts_root_parallelism} is ignored by the lockstep engine (one root per game); "
                      "MCTS.run honours it", RuntimeWarning, stacklevel=3)


def run_selfplay(best_model, cfg, iter_idx: int, pool=None, model_path=None, model_version: int = 0):
    """train.py:395-463 on the GPU: returns (samples, black_wins, white_wins, draws) for this rank's share of
    cfg.selfplay_games_per_iter games (all of them without torch.distributed). `pool`, `model_path` and
    `model_version` belong to upstream's worker processes and are accepted for call compatibility."""
    del pool, model_path, model_version
    from .selfplay import BatchedSelfPlay, SelfPlayConfig
    _warn_root_parallelism(cfg, "run_selfplay")
    _, rank, world = _world()
    first, mine = rank_share(cfg.selfplay_games_per_iter, rank, world)
    if mine <= 0:
        return [], 0, 0, 0
    best_model.eval()
    on_gpu = next(best_model.parameters()).is_cuda
    slots = min(mine, cfg.lockstep_games if cfg.lockstep_games > 0 else 4096)
    # With a move's host half on the device (az_lane_*) a lane needs no partner to hide it: two lanes from 256 games on, so that
    # one lane's rules / search kernels run under the other's network, and never more (measured on the 192ch x 12 network, ms per
    # lockstep move over a whole game span, lanes 1 / 2: 256 games 10.8 / 10.3, 512 games 16.7 / 14.1, 1024 games 26.6 / 20.7, 2048
    # games 44.5 / 38.3 and 43.4 with four; 64ch x 4: 512 games 2.16 / 1.98, 1024 games 3.30 / 2.55; 4096 games of the headline
    # bench: 15.6 M sims/s with two lanes, 15.1 M with four, whose concurrent tower launches evict each other's weights)
    lanes = cfg.selfplay_lanes if cfg.selfplay_lanes > 0 else (2 if slots >= 256 else 1)
    sp = BatchedSelfPlay(best_model if on_gpu else _host_evaluator(best_model), SelfPlayConfig.from_train_config(cfg), games=slots,
                         seed0=selfplay_seed0(cfg, iter_idx) + first, dedup_leaves=not cfg.no_dedup_leaves, lanes=lanes,
                         tower_workgroups=None)
    return sp.run(num_games=mine)


# ----------------------------------------------------------------------------------------------- training step
def train_candidate(candidate, buffer: ReplayBuffer, cfg):
    """train.py:504-554: cfg.train_steps_per_iter steps of AdamW with cosine decay to 0.1*lr on batches drawn
    from the buffer; returns the means of (loss, policy loss, value loss). Batches are assembled on the GPU
    when the model lives there; gradients are averaged over the ranks under torch.distributed."""
    from .train_step import DataParallelTrainer
    dev = next(candidate.parameters()).device
    trainer = DataParallelTrainer(candidate, lr=cfg.lr, weight_decay=cfg.weight_decay, total_steps=cfg.train_steps_per_iter,
                                  grad_clip_norm=cfg.grad_clip_norm, huber_delta=cfg.value_huber_delta,
                                  entropy_coeff=cfg.policy_entropy_coeff, target_smoothing=cfg.policy_target_smoothing,
                                  use_amp=cfg.use_amp and str(cfg.device).startswith("cuda"))
    use_mirror = dev.type == "cuda"
    if use_mirror:
        try:
            buffer.attach_device(dev, cfg.board_size, cfg.history_len)
        except ValueError:   # foreign samples that are not to_tensor() planes: assemble on the host
            use_mirror = False
    candidate.train()
    log = []
    for _ in range(cfg.train_steps_per_iter):
        if use_mirror:
            states, pis, zs = buffer.sample_device(cfg.batch_size, cfg.board_size, cfg.augment_symmetry)
        else:
            states, pis, zs = buffer.sample(batch_size=cfg.batch_size, board_size=cfg.board_size, augment_symmetry=cfg.augment_symmetry)
            states, pis, zs = states.to(dev), pis.to(dev), zs.to(dev)
        log.append(trainer.step(states, pis, zs))
    arr = np.array(log, dtype=np.float64).reshape(-1, 3)
    return float(np.mean(arr[:, 0])), float(np.mean(arr[:, 1])), float(np.mean(arr[:, 2]))


# ----------------------------------------------------------------------------------------------- arena
def _arena_results(model_a, model_b, cfg, colours):
    """Winner (+1 A, -1 B, 0 draw) of one game per entry of `colours` (True: A plays black), all games in
    lockstep: each move, the games where A is to move are searched with A's network, the others with B's.
    Moves are the most visited child (first maximum), no noise (train.py:566-573)."""
    from .engine import Engine
    _warn_root_parallelism(cfg, "arena")
    G = len(colours)
    n, A = cfg.board_size, cfg.board_size * cfg.board_size + 1
    eng = Engine(board_size=n, history_len=cfg.history_len, games=G, komi=cfg.komi, num_simulations=cfg.eval_simulations,
                 c_base=cfg.c_base, c_init=cfg.c_init, dirichlet_alpha=cfg.dirichlet_alpha, dirichlet_eps=cfg.dirichlet_eps,
                 inference_batch_size=cfg.mcts_inference_batch, topk_expand=cfg.mcts_topk_expand,
                 virtual_loss=cfg.mcts_virtual_loss, dedup_leaves=not getattr(cfg, "no_dedup_leaves", False))
    evs = []
    for m in (model_a, model_b):
        m.eval()
        if next(m.parameters()).is_cuda:
            net = m.native_auto(eng.device)      # bf16 MFMA kernels, or the float32 kernels for shapes those do not cover
            evs.append((net.forward, net.layout, net))
        else:
            evs.append((_host_evaluator(m), N.AZ_FEAT_F32_NCHW, None))
    a_black = np.asarray(colours, bool)
    result = np.zeros(G, np.int64)
    live = np.ones(G, bool)
    while live.any():
        _, info = eng.status()
        info = info.cpu().numpy()
        over = live & (info[:, 6] != 0)
        for g in np.nonzero(over)[0]:
            w = int(info[g, 7])
            result[g] = 0 if w == 0 else (1 if (w == BLACK) == bool(a_black[g]) else -1)
        live &= ~over
        if not live.any():
            break
        a_turn = (info[:, 0] == BLACK) == a_black
        actions = np.full(G, -1, np.int32)
        for side, (ev, layout, net) in enumerate(evs):
            mask = live & (a_turn if side == 0 else ~a_turn)
            if mask.any():
                visits, _ = eng.run_search(ev, add_noise=False, enable=mask.astype(np.uint8), layout=layout, native_net=net)
                actions[mask] = visits.cpu().numpy()[mask].argmax(axis=1)
        eng.play(torch.from_numpy(actions))
    eng.close()
    return result


def _game_outcomes(model_a, model_b, cfg, games):
    """Outcome per game id 0..games-1 (A is black in even games, train.py:563): the even games are one and
    the same game, the odd games another, so at most two games are played."""
    if games <= 0:
        return []
    distinct = _arena_results(model_a, model_b, cfg, [True, False][: min(2, games)])
    return [int(distinct[g % 2]) for g in range(games)]


def play_match(model_a, model_b, cfg, games: int):
    """(wins, losses, draws) of model_a over `games` games (train.py:557-583)."""
    out = _game_outcomes(model_a, model_b, cfg, games)
    return out.count(1), out.count(-1), out.count(0)


# ---- host bookkeeping around the matches: the score interval below, the Elo update, the model-pool files, the PBT
# mutation and the checkpoint readers further down restate upstream's formulas, file formats and random-draw order
# (train.py:596-781), because results and files must interchange with upstream's; nothing in them runs on the GPU.
def _score_interval_from_counts(wins: int, losses: int, draws: int, z: float = 1.96):
    """Normal interval of the mean score with the empirical variance of outcomes {0, 0.5, 1} (train.py:596-610)."""
    total = wins + losses + draws
    if total <= 0:
        return 0.0, 1.0
    mean = (float(wins) + 0.5 * float(draws)) / float(total)
    var = (float(wins) * (1.0 - mean) ** 2 + float(draws) * (0.5 - mean) ** 2 + float(losses) * (0.0 - mean) ** 2) / float(total)
    margin = z * float(np.sqrt(max(0.0, var) / float(total)))
    return max(0.0, mean - margin), min(1.0, mean + margin)


def play_match_sprt(model_a, model_b, cfg, games: int, threshold: float, delta: float):
    """train.py:613-660: stop once (from the 6th game on) the score interval clears threshold +- delta;
    otherwise accept iff the mean score reaches the threshold."""
    wins = losses = draws = 0
    accepted = False
    for outcome in _game_outcomes(model_a, model_b, cfg, games):
        wins += outcome == 1
        losses += outcome == -1
        draws += outcome == 0
        if wins + losses + draws >= 6:
            low, high = _score_interval_from_counts(wins, losses, draws)
            if low > threshold + delta:
                accepted = True
                break
            if high < threshold - delta:
                break
    if not accepted:
        total = max(1, wins + losses + draws)
        accepted = (float(wins) + 0.5 * float(draws)) / float(total) >= threshold
    return int(wins), int(losses), int(draws), bool(accepted)


def expected_score(r_a: float, r_b: float) -> float:
    return 1.0 / (1.0 + 10.0 ** ((r_b - r_a) / 400.0))


def update_elo_pair(r_a: float, r_b: float, score_a: float, games: int, k: float):
    delta = k * (score_a - games * expected_score(r_a, r_b))
    return r_a + delta, r_b - delta


def apply_pbt_mutation(cfg):
    """train.py:673-684: log-normal jitter of lr and c_init, additive jitter of the noise weight and of temp_moves."""
    sigma = max(0.01, cfg.pbt_sigma)
    lr_scale = float(np.exp(np.random.normal(0.0, sigma)))
    c_scale = float(np.exp(np.random.normal(0.0, sigma)))
    eps_shift = float(np.random.normal(0.0, sigma * 0.2))
    temp_shift = int(round(np.random.normal(0.0, max(1.0, cfg.temp_moves * sigma * 0.2))))
    cfg.lr = float(np.clip(cfg.lr * lr_scale, 1e-5, 5e-3))
    cfg.c_init = float(np.clip(cfg.c_init * c_scale, 0.5, 3.5))
    cfg.dirichlet_eps = float(np.clip(cfg.dirichlet_eps + eps_shift, 0.05, 0.4))
    cfg.temp_moves = int(np.clip(cfg.temp_moves + temp_shift, 4, 40))


# ----------------------------------------------------------------------------------------------- model pool / files
def load_model_pool(pool_path: str) -> dict:
    if not os.path.exists(pool_path):
        return {"entries": {}, "best_name": ""}
    with open(pool_path, "r", encoding="utf-8") as f:
        return json.load(f)


def save_model_pool(pool_path: str, pool: dict):
    with open(pool_path, "w", encoding="utf-8") as f:
        json.dump(pool, f, ensure_ascii=True, indent=2)


def _save_model_file(path, model, cfg, it):
    torch.save({"model": model.state_dict(), "cfg": dict(cfg.__dict__), "iter": it}, path)


def init_or_load_pool(cfg, best_model, start_iter: int):
    """train.py:699-717 -> (pool, pool_dir, pool_path); a fresh pool starts with the current best network."""
    pool_dir = os.path.join(cfg.save_dir, "model_pool")
    os.makedirs(pool_dir, exist_ok=True)
    pool_path = os.path.join(cfg.save_dir, "model_pool.json")
    pool = load_model_pool(pool_path)
    entries = pool.setdefault("entries", {})
    if pool.get("best_name", "") in entries and pool.get("best_name", ""):
        return pool, pool_dir, pool_path
    name = f"model_iter_{start_iter - 1:03d}"
    path = os.path.join(pool_dir, f"{name}.pt")
    _save_model_file(path, best_model, cfg, start_iter - 1)
    entries[name] = {"path": path, "elo": cfg.elo_init, "games": 0, "iter": start_iter - 1}
    pool["best_name"] = name
    save_model_pool(pool_path, pool)
    return pool, pool_dir, pool_path


def sample_pool_opponents(pool: dict, exclude_name: str, k: int):
    """train.py:720-737: the strongest max(1, k//2) by Elo, then random others (global `random`), then fill."""
    entries = pool.get("entries", {})
    names = [n for n in entries if n != exclude_name]
    if not names or k <= 0:
        return []
    names.sort(key=lambda n: entries[n]["elo"], reverse=True)
    n_top = min(len(names), max(1, k // 2))
    chosen, rest = names[:n_top], names[n_top:]
    n_rand = min(len(rest), k - len(chosen))
    if n_rand > 0:
        chosen = chosen + random.sample(rest, n_rand)
    want = min(k, len(names))
    if len(chosen) < want:
        chosen.extend([n for n in names if n not in chosen][: want - len(chosen)])
    return chosen


def prune_pool(pool: dict, keep_name: str, max_size: int):
    """train.py:740-754: drop the oldest entries (by iteration) beyond max_size, never keep_name."""
    entries = pool.get("entries", {})
    excess = len(entries) - max_size
    if excess <= 0:
        return
    for name in sorted((n for n in entries if n != keep_name), key=lambda n: entries[n]["iter"])[:excess]:
        try:
            os.remove(entries[name]["path"])
        except OSError:
            pass
        entries.pop(name, None)


def load_model_from_file(path: str, device: str):
    ckpt = torch.load(path, map_location=device, weights_only=True)
    cfg = ckpt.get("cfg", {})
    model = AlphaZeroNet(board_size=cfg.get("board_size", 9), history_len=cfg.get("history_len", 8),
                         channels=cfg.get("channels", 192), num_blocks=cfg.get("num_blocks", 12),
                         use_se=cfg.get("use_se", True), value_hidden_dim=cfg.get("value_hidden_dim", 256)).to(device)
    model.load_state_dict(ckpt.get("model", ckpt))
    model.eval()
    return model, cfg


def runtime_model_path_for_version(save_dir: str, version: int) -> str:
    return os.path.join(save_dir, f"_runtime_best_v{version:06d}.pt")


def save_runtime_model_snapshot(model, save_dir: str, version: int) -> str:
    path = runtime_model_path_for_version(save_dir, version)
    torch.save({"model": model.state_dict()}, path)
    return path


# ----------------------------------------------------------------------------------------------- main loop
def _broadcast_flag(dist, value: bool, device) -> bool:
    if dist is None:
        return value
    t = torch.tensor([1 if value else 0], device=device)
    broadcast_(dist, t, src=0)
    return bool(int(t.item()))


def sync_batchnorm_buffers(dist, model, world: int):
    """Average the floating-point buffers (BatchNorm running_mean / running_var) over the ranks, and take rank 0's
    integer ones (num_batches_tracked): each rank trained on its own batches, so these statistics differ while the
    parameters do not. One flat all-reduce."""
    floats = [b for b in model.buffers() if b.is_floating_point()]
    if floats:
        flat = torch.cat([b.detach().reshape(-1).to(torch.float32) for b in floats])
        all_reduce_(dist, flat)
        flat.div_(world)
        off = 0
        for b in floats:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    for b in model.buffers():
        if not b.is_floating_point():
            broadcast_(dist, b, src=0)


def run_training(cfg):
    """train.py:784-1023. Files written under cfg.save_dir have upstream's names and dict keys:
    iter_XXX.pt, best.pt, model_pool.json, model_pool/model_iter_XXX.pt, _runtime_best_vNNNNNN.pt."""
    dist, rank, world = _world()
    set_seed(cfg.seed + rank)
    if cfg.train_torch_threads > 0:
        torch.set_num_threads(cfg.train_torch_threads)
    if rank == 0:
        os.makedirs(cfg.save_dir, exist_ok=True)
    dev = torch.device(cfg.device)
    best_model = build_model(cfg).to(dev)
    buffer = ReplayBuffer(cfg.replay_size)
    start_iter = 1
    if cfg.resume:
        ckpt = torch.load(cfg.resume, map_location=dev, weights_only=not cfg.save_buffer)
        best_model.load_state_dict(ckpt["best_model"] if "best_model" in ckpt else ckpt["model"])
        start_iter = int(ckpt.get("iter", 0)) + 1
        if cfg.save_buffer and "buffer" in ckpt:
            buffer.extend([x[:4] for x in ckpt["buffer"]])
        print(f"[Resume] from {cfg.resume}, next_iter={start_iter}, buffer={len(buffer)}")
    if dist is not None:   # one set of starting weights
        for t in best_model.state_dict().values():
            broadcast_(dist, t, src=0)
    pool = pool_dir = pool_path = None
    runtime_version = 0
    if rank == 0:
        pool, pool_dir, pool_path = init_or_load_pool(cfg, best_model, start_iter)
        save_runtime_model_snapshot(best_model, cfg.save_dir, runtime_version)
    stalled = 0
    recent_keys, recent_set = deque(), set()

    for it in range(start_iter, cfg.iterations + 1):
        best_model.eval()
        new_samples, black_wins, white_wins, draws = run_selfplay(best_model, cfg, it)
        buffer.extend(filter_selfplay_samples(new_samples, recent_keys, recent_set))
        if dist is not None:
            tally = torch.tensor([black_wins, white_wins, draws, len(buffer)], device=dev)
            smallest = tally[3:].clone()
            all_reduce_(dist, tally)
            all_reduce_(dist, smallest, op=dist.ReduceOp.MIN)
            black_wins, white_wins, draws = (int(x) for x in tally[:3])
            ready = int(smallest.item()) >= cfg.batch_size
        else:
            ready = len(buffer) >= cfg.batch_size
        if not ready:
            if rank == 0:
                print(f"[Iter {it}] buffer too small: {len(buffer)} < {cfg.batch_size}, skip training")
            continue

        candidate = build_model(cfg).to(dev)
        candidate.load_state_dict(copy.deepcopy(best_model.state_dict()))
        loss_mean, pol_mean, val_mean = train_candidate(candidate, buffer, cfg)
        if dist is not None:   # parameters are equal on every rank (averaged gradients); BatchNorm running statistics are
            sync_batchnorm_buffers(dist, candidate, world)   # not: average them, so every rank holds the network that is tested and saved

        promoted, arena = True, (0, 0, 0)
        if cfg.arena_games > 0 and rank == 0:
            candidate.eval()
            if cfg.arena_sprt:
                *arena, promoted = play_match_sprt(candidate, best_model, cfg, cfg.arena_games, cfg.promote_threshold, cfg.arena_sprt_delta)
            else:
                arena = play_match(candidate, best_model, cfg, cfg.arena_games)
                promoted = (float(arena[0]) + 0.5 * float(arena[2])) / float(max(1, sum(arena))) >= cfg.promote_threshold
        promoted = _broadcast_flag(dist, promoted, dev)

        elo_log, bench_log = [], []
        if promoted:
            best_model.load_state_dict(copy.deepcopy(candidate.state_dict()))
            stalled = 0
            if rank == 0:
                runtime_version += 1
                save_runtime_model_snapshot(best_model, cfg.save_dir, runtime_version)
                name = f"model_iter_{it:03d}"
                path = os.path.join(pool_dir, f"{name}.pt")
                _save_model_file(path, best_model, cfg, it)
                entries = pool["entries"]
                base = entries.get(pool.get("best_name", ""), {}).get("elo", cfg.elo_init)
                entries[name] = {"path": path, "elo": float(base), "games": 0, "iter": it}
                pool["best_name"] = name
                for opp in sample_pool_opponents(pool, exclude_name=name, k=cfg.pool_opponents):
                    opp_model, opp_cfg = load_model_from_file(entries[opp]["path"], cfg.device)
                    if int(opp_cfg.get("board_size", cfg.board_size)) != cfg.board_size:
                        continue
                    w, l, d = play_match(best_model, opp_model, cfg, cfg.elo_games_per_opponent)
                    r_new, r_opp = update_elo_pair(float(entries[name]["elo"]), float(entries[opp]["elo"]),
                                                   score_a=float(w) + 0.5 * float(d), games=w + l + d, k=cfg.elo_k)
                    entries[name]["elo"], entries[opp]["elo"] = r_new, r_opp
                    entries[name]["games"] += w + l + d
                    entries[opp]["games"] += w + l + d
                    elo_log.append((opp, w, l, d, r_new, r_opp))
                prune_pool(pool, keep_name=name, max_size=cfg.pool_max_size)
                save_model_pool(pool_path, pool)
                _save_model_file(os.path.join(cfg.save_dir, "best.pt"), best_model, cfg, it)
                for bench in sample_pool_opponents(pool, exclude_name=name, k=cfg.benchmark_opponents):
                    bench_model, bench_cfg = load_model_from_file(pool["entries"][bench]["path"], cfg.device)
                    if int(bench_cfg.get("board_size", cfg.board_size)) != cfg.board_size:
                        continue
                    bw, bl, bd = play_match(best_model, bench_model, cfg, cfg.benchmark_games_per_opponent)
                    bench_log.append((bench, bw, bl, bd, bw / max(1, bw + bl)))
        else:
            stalled += 1
            if cfg.pbt and stalled >= max(1, cfg.pbt_patience):
                old = (cfg.lr, cfg.c_init, cfg.dirichlet_eps, cfg.temp_moves)
                if rank == 0:
                    apply_pbt_mutation(cfg)
                if dist is not None:
                    knobs = torch.tensor([cfg.lr, cfg.c_init, cfg.dirichlet_eps, float(cfg.temp_moves)], dtype=torch.float64, device=dev)
                    broadcast_(dist, knobs, src=0)
                    cfg.lr, cfg.c_init, cfg.dirichlet_eps, cfg.temp_moves = (float(knobs[0]), float(knobs[1]), float(knobs[2]), int(knobs[3]))
                stalled = 0
                if rank == 0:
                    print(f"[PBT] mutate lr {old[0]:.2e}->{cfg.lr:.2e}, c_init {old[1]:.3f}->{cfg.c_init:.3f}, "
                          f"dir_eps {old[2]:.3f}->{cfg.dirichlet_eps:.3f}, temp_moves {old[3]}->{cfg.temp_moves}")

        if rank == 0:
            ckpt = {"iter": it, "model": best_model.state_dict(), "best_model": best_model.state_dict(),
                    "candidate_model": candidate.state_dict(), "cfg": dict(cfg.__dict__), "buffer_size": len(buffer),
                    "promoted": promoted, "arena": {"wins": arena[0], "losses": arena[1], "draws": arena[2]},
                    "pool_best": pool.get("best_name", "")}
            if cfg.save_buffer:
                ckpt["buffer"] = list(buffer.data)
            iter_ckpt = os.path.join(cfg.save_dir, f"iter_{it:03d}.pt")
            torch.save(ckpt, iter_ckpt)
            print(f"[Iter {it}] selfplay B/W/D={black_wins}/{white_wins}/{draws} workers={cfg.selfplay_workers} "
                  f"async={cfg.async_selfplay} buffer={len(buffer)} loss={loss_mean:.4f} pol={pol_mean:.4f} val={val_mean:.4f} "
                  f"arena={arena[0]}/{arena[1]}/{arena[2]} promoted={promoted} saved={iter_ckpt}")
            if promoted and elo_log:
                best = pool.get("best_name", "")
                print(f"[ELO] {best} rating={pool['entries'][best]['elo']:.1f}")
                for opp, w, l, d, r_new, r_opp in elo_log:
                    print(f"[ELO] vs {opp}: {w}/{l}/{d}, new={r_new:.1f}, opp={r_opp:.1f}")
            for bench, bw, bl, bd, wr in bench_log:
                print(f"[Benchmark] vs {bench}: {bw}/{bl}/{bd}, winrate={wr:.3f}")


def main(argv=None):
    """go_zero.train.main. Under `python -m torch.distributed.run --nproc-per-node N` (RANK / WORLD_SIZE /
    LOCAL_RANK in the environment) with a cuda device, each process takes GPU LOCAL_RANK and joins an RCCL group."""
    cfg = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and str(cfg.device).startswith("cuda"):
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        cfg.device = f"cuda:{local}"
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        try:
            run_training(cfg)
        finally:
            dist.destroy_process_group()
    else:
        run_training(cfg)


if __name__ == "__main__":
    main()
