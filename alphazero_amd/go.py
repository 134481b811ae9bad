"""Host-side GoState with the upstream call surface (go_zero/go.py:37-304). The object is an immutable
value holder; every rule computation (legality, captures, superko, scoring, feature planes) runs in
the HIP engine through the C ABI -- a one-game engine per (board size, history, komi) is kept for it.
There is no CPU implementation of the rules in this package.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

BLACK, WHITE, EMPTY = 1, -1, 0


def action_size(board_size: int) -> int:
    return board_size * board_size + 1  # every point plus pass


def encode_action(board_size: int, move) -> int:
    return board_size * board_size if move is None else move[0] * board_size + move[1]


def decode_action(board_size: int, action: int):
    return None if action == board_size * board_size else divmod(action, board_size)


def tensor_channels(history_len: int) -> int:
    return 2 * history_len + 5


_ENGINES: dict = {}
# States are values and may be queried from any thread (the scripts search on a worker thread and keep using states on
# the main one); the one-game engine behind the queries holds one state at a time, so upload + query is one critical section.
_ENGINE_LOCK = threading.RLock()


def _engine(n: int, hist: int, komi: float):
    """The one-game device engine used for single-state queries."""
    import torch
    from .engine import Engine
    key = (n, hist, float(komi), torch.cuda.current_device() if torch.cuda.is_available() else -1)
    eng = _ENGINES.get(key)
    if eng is None:
        eng = _ENGINES[key] = Engine(board_size=n, history_len=hist, games=1, komi=komi, num_simulations=0)
    return eng


def _side_key(to_play: int) -> int:
    from . import _native as N
    return int(N.lib().az_zobrist_key(2 * 361 + (1 if to_play == WHITE else 0)))


@dataclass(frozen=True)
class GoState:
    board: np.ndarray
    to_play: int = BLACK
    consecutive_passes: int = 0
    move_count: int = 0
    komi: float = 5.5
    history_len: int = 8
    board_history: tuple = ()
    seen_hashes: frozenset = frozenset()   # 64-bit Zobrist keys of (board, side to move), not blake2b digests
    black_captures: int = 0
    white_captures: int = 0
    last_move: Optional[tuple] = None
    _cache: dict = field(default_factory=dict, init=False, repr=False, compare=False)

    # ---------------------------------------------------------------- construction
    @staticmethod
    def new_game(board_size: int = 9, komi: float = 5.5, history_len: int = 8) -> "GoState":
        board = np.zeros((board_size, board_size), dtype=np.int8)
        return GoState(board=board, komi=komi, history_len=history_len, board_history=(board.copy(),),
                       seen_hashes=frozenset({_side_key(BLACK)}))

    @property
    def board_size(self) -> int:
        return self.board.shape[0]

    def in_bounds(self, r: int, c: int) -> bool:
        return 0 <= r < self.board_size and 0 <= c < self.board_size

    def neighbors(self, r: int, c: int):
        for dr, dc in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            if self.in_bounds(r + dr, c + dc):
                yield r + dr, c + dc

    # ---------------------------------------------------------------- device round trips
    def _packed(self, seen_cap: int):
        """(board, history, scalars, seen) host tensors in the layout az_rules_import takes."""
        import torch
        n, h = self.board_size, self.history_len
        hist = np.zeros((1, h, n * n), np.int8)
        for i, b in enumerate(self.board_history[:h]):
            hist[0, i] = np.asarray(b, np.int8).reshape(-1)
        keys = sorted(self.seen_hashes)
        if len(keys) > seen_cap:
            raise ValueError("seen_hashes larger than any legal game can produce")
        seen = np.zeros((1, seen_cap), np.uint64)
        seen[0, : len(keys)] = np.array(keys, np.uint64)
        sc = np.array([[self.to_play, self.consecutive_passes, self.move_count, self.black_captures,
                        self.white_captures, -1 if self.last_move is None else encode_action(n, self.last_move),
                        min(len(self.board_history), h), len(keys)]], np.int32)
        return (torch.from_numpy(np.ascontiguousarray(self.board, np.int8).reshape(1, -1)), torch.from_numpy(hist),
                torch.from_numpy(sc), torch.from_numpy(seen.view(np.int64)))

    def _upload(self):
        eng = _engine(self.board_size, self.history_len, self.komi)
        eng.import_state(*self._packed(eng.seen_cap))
        return eng

    def _evaluated(self) -> dict:
        c = self._cache
        if "winner" not in c:
            with _ENGINE_LOCK:
                eng = self._upload()
                score, info = eng.status()
                c["legal"] = eng.legal_mask().cpu().numpy()[0].astype(bool)
                c["tensor"] = eng.features().cpu().numpy()[0]
                c["score"] = tuple(float(x) for x in score.cpu().numpy()[0])
                c["winner"] = int(info.cpu().numpy()[0, 7])   # written last: the key the check above looks for
        return c

    # ---------------------------------------------------------------- upstream methods
    def is_legal(self, move) -> bool:                       # go.py:122-143
        if move is None:
            return True
        r, c = move
        if not self.in_bounds(r, c) or self.board[r, c] != EMPTY:
            return False
        return bool(self._evaluated()["legal"][r * self.board_size + c])

    def legal_actions(self) -> list:                        # go.py:145-157 (ascending, pass last)
        return [int(a) for a in np.nonzero(self._evaluated()["legal"])[0]]

    def play(self, move) -> "GoState":                      # go.py:159-203
        """One upload per call: the move is checked by the device (its status stands in for go.py:179-180's is_legal
        call unless this state's legal mask is already cached), and while the new position is still resident its legal
        mask, feature planes and score are read back into the new state's cache, so the caller's next is_legal /
        legal_actions / to_tensor / winner on it costs nothing."""
        n, h = self.board_size, self.history_len
        if move is not None:
            r, c = move
            if not self.in_bounds(r, c) or self.board[r, c] != EMPTY or \
                    ("legal" in self._cache and not self._cache["legal"][r * n + c]):
                raise ValueError(f"Illegal move: {move}")
        with _ENGINE_LOCK:
            eng = self._upload()
            status = eng.play([encode_action(n, move)], raise_on_illegal=False)
            if int(status.cpu()[0]) != 0:
                raise ValueError(f"Illegal move: {move}")
            board, hist, sc, seen = (t.cpu().numpy() for t in eng.export_state())
            score, info = eng.status()
            cache = {"legal": eng.legal_mask().cpu().numpy()[0].astype(bool), "tensor": eng.features().cpu().numpy()[0],
                     "score": tuple(float(x) for x in score.cpu().numpy()[0])}
            cache["winner"] = int(info.cpu().numpy()[0, 7])
        sc = sc[0]
        nh = int(sc[6])
        new = GoState(board=board[0].reshape(n, n).copy(), to_play=int(sc[0]), consecutive_passes=int(sc[1]),
                      move_count=int(sc[2]), komi=self.komi, history_len=h,
                      board_history=tuple(hist[0, i].reshape(n, n).copy() for i in range(nh)),
                      seen_hashes=frozenset(int(x) for x in seen[0, : int(sc[7])].view(np.uint64)),
                      black_captures=int(sc[3]), white_captures=int(sc[4]),
                      last_move=None if sc[5] < 0 else divmod(int(sc[5]), n))
        new._cache.update(cache)
        return new

    def is_terminal(self) -> bool:                          # go.py:205-206
        return self.consecutive_passes >= 2 or self.move_count >= 2 * self.board_size * self.board_size

    def score(self):                                        # go.py:242-249
        return self._evaluated()["score"]

    def winner(self) -> int:                                # go.py:251-257
        return self._evaluated()["winner"]

    def outcome_for_current_player(self) -> float:          # go.py:259-263
        w = self.winner()
        return 0.0 if w == 0 else (1.0 if w == self.to_play else -1.0)

    def to_tensor(self) -> np.ndarray:                      # go.py:265-304 (cached; do not mutate)
        return self._evaluated()["tensor"]
