"""Two-lane drivers built one after another in one process: each must run as fast as the first (their lanes get streams
that were checked to overlap, alphazero_amd.selfplay.lane_streams). AZ_AB_POOL=1: lanes on consecutive streams of torch's
pool instead, as before that check: python tools/ab_stream_pool.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig, lane_streams

torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=64, num_blocks=4).eval().cuda()
cfg = SelfPlayConfig(board_size=9, mcts_simulations=192, mcts_inference_batch=16, mcts_topk_expand=48)
raw = os.environ.get("AZ_AB_POOL") == "1"
held = []
for k in range(8):
    streams = [torch.cuda.Stream(), torch.cuda.Stream()] if raw else None
    held += streams or []
    sp = BatchedSelfPlay(net, cfg, games=512, seed0=1, lanes=2, tower_workgroups=1024, streams=streams)
    sp.begin(num_games=10 ** 9)
    for _ in range(6):
        sp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        sp.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"driver {k}: {dt * 1e3:.3f} ms/move")
    del sp
if not raw:
    dev = torch.device("cuda", torch.cuda.current_device())
    got = lane_streams(dev, 8)
    print("distinct overlapping streams found for 8 lanes:", len({s.cuda_stream for s in got}))
