"""Where the host time of a lockstep move goes with the native lane (cProfile over 60 moves of one lane):
python tools/host_profile_lane.py [games] [channels] [blocks] [sims]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig

games, ch, blocks, sims = (int(x) for x in (sys.argv[1:] + [512, 64, 4, 192][len(sys.argv) - 1:])[:4])
torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=ch, num_blocks=blocks).eval().cuda()
cfg = SelfPlayConfig(board_size=9, mcts_simulations=sims, mcts_inference_batch=16, mcts_topk_expand=48)
sp = BatchedSelfPlay(net, cfg, games=games, seed0=1, lanes=1)
sp.begin(num_games=10 ** 9)
for _ in range(30):
    sp.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(60):
    sp.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
