"""A few lockstep moves of the headline configuration on one lane, for kernel traces: python tools/sel_probe.py [sync] [noreuse]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=192, num_blocks=12).eval().to("cuda:0")
cfg = SelfPlayConfig(board_size=9, mcts_simulations=256, mcts_inference_batch=16, mcts_topk_expand=48, mcts_virtual_loss=1.0)
sp = BatchedSelfPlay(net, cfg, games=4096, seed0=1, lanes=1, reuse_roots="noreuse" not in sys.argv)
if "sync" in sys.argv:
    sp.set_async(False)
sp.begin(num_games=10 ** 9)
for _ in range(4):
    sp.step()
torch.cuda.synchronize()
print("async" if sp.async_search else "sync", sp.counters["sims"])
