"""go_zero.mcts surface -> alphazero_amd.mcts."""
from alphazero_amd.mcts import MCTS  # noqa: F401
