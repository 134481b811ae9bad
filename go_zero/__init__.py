"""Drop-in `go_zero` package: the upstream import surface (go_zero/__init__.py:1-13) served by the
MI355X engine in alphazero_amd. `from go_zero import GoState, MCTS, AlphaZeroNet` keeps working for
scripts written against the upstream project."""
from .go import GoState, encode_action, decode_action, action_size, tensor_channels
from .model import AlphaZeroNet
from .mcts import MCTS

__all__ = ["GoState", "encode_action", "decode_action", "action_size", "tensor_channels", "AlphaZeroNet", "MCTS"]
