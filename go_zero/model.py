"""go_zero.model surface -> alphazero_amd.model."""
from alphazero_amd.model import AlphaZeroNet, PolicyValueNet, ResidualBlock, SqueezeExcite  # noqa: F401
