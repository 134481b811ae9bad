"""ctypes front end of the CPU oracle (oracle/az_oracle.c). TEST INFRASTRUCTURE ONLY.

`OracleState` mirrors the upstream GoState methods (go_zero/go.py:37-304) and `OracleMCTS`
mirrors MCTS._run_single_root_search / MCTS.run (go_zero/mcts.py:165-284) with the evaluator
passed in as a plain numpy function  f(planes f32 [b,C,n,n]) -> (logits f32 [b,A], value f32 [b]).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libaz_oracle.so")
    src = os.path.join(_HERE, "az_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libaz_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        L.azo_state_new.restype = vp; L.azo_state_new.argtypes = [i32, f64, i32]
        L.azo_state_clone.restype = vp; L.azo_state_clone.argtypes = [vp]
        L.azo_state_free.argtypes = [vp]
        L.azo_is_legal.argtypes = [vp, i32]
        L.azo_legal_mask.argtypes = [vp, vp]
        L.azo_play.argtypes = [vp, i32]
        L.azo_is_terminal.argtypes = [vp]
        L.azo_score.argtypes = [vp, vp, vp]
        L.azo_winner.argtypes = [vp]
        L.azo_outcome.restype = f64; L.azo_outcome.argtypes = [vp]
        L.azo_to_tensor.argtypes = [vp, vp]
        L.azo_state_get.argtypes = [vp, vp, vp]
        L.azo_state_hash.restype = C.c_uint64; L.azo_state_hash.argtypes = [vp, i32, i32]
        L.azo_pairwise_sum.restype = f64; L.azo_pairwise_sum.argtypes = [vp, i32]
        L.azo_masked_softmax.argtypes = [vp, vp, i32, vp]
        L.azo_mcts_new.restype = vp; L.azo_mcts_new.argtypes = [i32, i32, f64, f64, f64, i32, i32, f64]
        L.azo_mcts_free.argtypes = [vp]
        L.azo_search_begin.argtypes = [vp, vp]
        L.azo_search_pending.argtypes = [vp, vp, i32]
        L.azo_search_feed.argtypes = [vp, vp, vp, vp]
        L.azo_search_result.argtypes = [vp, vp, vp]
        L.azo_search_root_children.argtypes = [vp, vp, vp, vp, vp]
        L.azo_search_stats.argtypes = [vp, vp]
        _LIB = L
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def state_hash(board: np.ndarray, to_play: int) -> int:
    b = np.ascontiguousarray(board, np.int8)
    return int(lib().azo_state_hash(_p(b), int(b.shape[0]), int(to_play)))


class OracleState:
    """Mutable-in-C, value-semantics-in-Python game state (play returns a new object)."""

    def __init__(self, handle, n, hist_len):
        self._h, self.n, self.hist_len = handle, n, hist_len

    @classmethod
    def new_game(cls, board_size=9, komi=5.5, history_len=8):
        h = lib().azo_state_new(board_size, float(komi), history_len)
        if not h:
            raise ValueError("unsupported board size / history length")
        return cls(h, board_size, history_len)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().azo_state_free(self._h)
            self._h = None

    def clone(self):
        return OracleState(lib().azo_state_clone(self._h), self.n, self.hist_len)

    @property
    def action_size(self):
        return self.n * self.n + 1

    def is_legal(self, action: int) -> bool:
        return bool(lib().azo_is_legal(self._h, int(action)))

    def legal_mask(self) -> np.ndarray:
        m = np.zeros(self.action_size, np.uint8)
        lib().azo_legal_mask(self._h, _p(m))
        return m

    def legal_actions(self) -> list[int]:
        return [int(a) for a in np.nonzero(self.legal_mask())[0]]

    def play(self, action: int) -> "OracleState":
        s = self.clone()
        if lib().azo_play(s._h, int(action)) != 0:
            n = self.n
            raise ValueError(f"Illegal move: {divmod(int(action), n)}")
        return s

    def play_inplace(self, action: int):
        if lib().azo_play(self._h, int(action)) != 0:
            raise ValueError(f"Illegal move: {divmod(int(action), self.n)}")

    def is_terminal(self) -> bool:
        return bool(lib().azo_is_terminal(self._h))

    def score(self):
        b, w = C.c_double(), C.c_double()
        lib().azo_score(self._h, C.byref(b), C.byref(w))
        return b.value, w.value

    def winner(self) -> int:
        return int(lib().azo_winner(self._h))

    def outcome_for_current_player(self) -> float:
        return float(lib().azo_outcome(self._h))

    def to_tensor(self) -> np.ndarray:
        t = np.zeros((2 * self.hist_len + 5, self.n, self.n), np.float32)
        lib().azo_to_tensor(self._h, _p(t))
        return t

    def snapshot(self):
        """(board int8 [n*n], meta int32 [10]) with meta laid out as tests/golden rules vectors."""
        b = np.zeros(self.n * self.n, np.int8)
        m = np.zeros(10, np.int32)
        lib().azo_state_get(self._h, _p(b), _p(m))
        return b, m

    @property
    def to_play(self) -> int:
        return int(self.snapshot()[1][0])


class OracleMCTS:
    def __init__(self, evaluator, board_size=9, num_simulations=128, c_base=19652.0, c_init=1.25,
                 dirichlet_alpha=0.3, dirichlet_eps=0.25, inference_batch_size=1, topk_expand=0,
                 virtual_loss=1.0):
        self.evaluator = evaluator
        self.n = board_size
        self.A = board_size * board_size + 1
        self.sims = num_simulations
        self.alpha, self.eps = dirichlet_alpha, dirichlet_eps
        self.B = max(1, inference_batch_size)
        self._h = lib().azo_mcts_new(board_size, num_simulations, c_base, c_init, dirichlet_eps,
                                     self.B, max(0, topk_expand), float(max(0.0, virtual_loss)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().azo_mcts_free(self._h)
            self._h = None

    def search(self, root: OracleState, add_noise: bool, rng=np.random):
        """mcts.py:165-225. Returns (visits f64 [A], root_value). Noise is drawn from `rng` exactly
        where upstream draws it (one dirichlet over the legal root actions)."""
        L = lib()
        L.azo_search_begin(self._h, root._h)
        cfeat = 2 * root.hist_len + 5
        feats = np.zeros((self.B, cfeat, self.n, self.n), np.float32)
        first = True
        while True:
            rows = L.azo_search_pending(self._h, _p(feats), self.B)
            if rows <= 0:
                break
            logits, values = self.evaluator(feats[:rows])
            logits = np.ascontiguousarray(logits, np.float32)
            values = np.ascontiguousarray(values, np.float32)
            noise = None
            if first and add_noise:
                nlegal = int(root.legal_mask().sum())
                noise = np.ascontiguousarray(rng.dirichlet([self.alpha] * nlegal), np.float64)
            L.azo_search_feed(self._h, _p(logits), _p(values), _p(noise) if noise is not None else None)
            first = False
        visits = np.zeros(self.A, np.float64)
        rv = C.c_double()
        L.azo_search_result(self._h, _p(visits), C.byref(rv))
        return visits, rv.value

    def root_children(self):
        a = np.zeros(self.A, np.int32); p = np.zeros(self.A, np.float64)
        v = np.zeros(self.A, np.int32); s = np.zeros(self.A, np.float64)
        k = lib().azo_search_root_children(self._h, _p(a), _p(p), _p(v), _p(s))
        return a[:k], p[:k], v[:k], s[:k]

    def stats(self):
        out = np.zeros(6, np.int64)
        lib().azo_search_stats(self._h, _p(out))
        return dict(rows=int(out[0]), unique_rows=int(out[1]), depth_sum=int(out[2]), sims=int(out[3]),
                    max_depth=int(out[4]), nodes=int(out[5]))

    def run(self, root: OracleState, temperature=1.0, add_exploration_noise=True, rng=np.random):
        """mcts.py:227-284 with root_parallelism=1. Returns (pi f32 [A], action, root_value)."""
        if root.is_terminal():
            pi = np.zeros(self.A, np.float32)
            pi[-1] = 1.0
            return pi, int(pi.argmax()), float(root.outcome_for_current_player())
        visits, root_value = self.search(root, add_exploration_noise, rng)
        pi = visits_to_pi(visits, temperature)
        action = int(rng.choice(len(pi), p=pi))
        return pi.astype(np.float32), action, root_value


def visits_to_pi(visits: np.ndarray, temperature: float) -> np.ndarray:
    """mcts.py:270-279 (float64 result)."""
    if temperature <= 1e-2:
        pi = np.zeros_like(visits)
        pi[np.argmax(visits)] = 1.0
        return pi
    v = np.power(visits, 1.0 / temperature)
    s = v.sum()
    return np.ones_like(v) / len(v) if s <= 0 else v / s


def masked_softmax(logits: np.ndarray, legal: np.ndarray) -> np.ndarray:
    lg = np.ascontiguousarray(logits, np.float32)
    m = np.ascontiguousarray(legal, np.uint8)
    out = np.zeros(lg.shape[0], np.float64)
    lib().azo_masked_softmax(_p(lg), _p(m), int(lg.shape[0]), _p(out))
    return out


def pairwise_sum(a: np.ndarray) -> float:
    a = np.ascontiguousarray(a, np.float64)
    return float(lib().azo_pairwise_sum(_p(a), int(a.shape[0])))
