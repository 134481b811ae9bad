"""go_zero.go surface (BLACK/WHITE/EMPTY, action helpers, GoState) -> alphazero_amd.go."""
from alphazero_amd.go import (BLACK, WHITE, EMPTY, GoState, action_size, encode_action, decode_action,  # noqa: F401
                              tensor_channels)
