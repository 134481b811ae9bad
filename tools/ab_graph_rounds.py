"""Rounds enqueued launch by launch against rounds replayed from a HIP graph, alternating windows in one process (the card's
clock state drifts between processes): python tools/ab_graph_rounds.py [games] [channels] [blocks] [sims] [lanes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig

games, ch, blocks, sims, lanes = (int(x) for x in (sys.argv[1:] + [512, 64, 4, 192, 2][len(sys.argv) - 1:])[:5])
torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=ch, num_blocks=blocks).eval().cuda()
cfg = SelfPlayConfig(board_size=9, mcts_simulations=sims, mcts_inference_batch=16, mcts_topk_expand=48)
sps = {}
order = (True, False) if os.environ.get("AZ_AB_GRAPH_FIRST") == "1" else (False, True)
for graph in order:
    sp = BatchedSelfPlay(net, cfg, games=games, seed0=1, lanes=lanes, graph_rounds=graph, tower_workgroups=None if lanes == 1 else 1024)
    sp.begin(num_games=10 ** 9)
    for _ in range(6):
        sp.step()
    sps[graph] = sp
torch.cuda.synchronize()
for rep in range(4):
    for graph in order:
        sp = sps[graph]
        t0 = time.perf_counter()
        for _ in range(20):
            sp.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"window {rep} graph {graph}: {dt * 1e3:.3f} ms/move (moves {6 + rep * 20}..{26 + rep * 20})")
