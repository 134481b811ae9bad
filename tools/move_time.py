"""Wall time per lockstep move without the tower's launch timing (which keeps the rounds out of a HIP graph):
python tools/move_time.py <games> <channels> <blocks> <sims> <topk> <lanes> [graph]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig

games, ch, blocks, sims, topk, lanes = (int(x) for x in sys.argv[1:7])
graph = "graph" in sys.argv[7:]
torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=ch, num_blocks=blocks).eval().cuda()
cfg = SelfPlayConfig(board_size=9, mcts_simulations=sims, mcts_inference_batch=16, mcts_topk_expand=topk)
sp = BatchedSelfPlay(net, cfg, games=games, seed0=1, lanes=lanes, graph_rounds=graph, tower_workgroups=None)
sp.begin(num_games=10 ** 9)
for _ in range(6):
    sp.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(40):
        sp.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"games {games} {ch}x{blocks} sims {sims} lanes {lanes} graph {graph}: {dt * 1e3:.3f} ms/move, {games * sims / dt / 1e6:.1f} M sims/s, graph replays {sp.counters['graph_replays']}")
