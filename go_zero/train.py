"""go_zero.train surface -> alphazero_amd.train (so that `from go_zero.train import main` keeps working)."""
from alphazero_amd.train import *  # noqa: F401,F403
from alphazero_amd.train import (TrainConfig, ReplayBuffer, _score_interval_from_counts, main,  # noqa: F401
                                 parse_args, run_training)

if __name__ == "__main__":
    main()
