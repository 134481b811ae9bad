"""Quick end-to-end lockstep self-play timing (development aid).

  python tools/e2e.py GAMES STEPS [DEDUP] [B] [BOARD] [CHANNELS] [BLOCKS] [SIMS] [TOPK] [LANES]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig

arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
G, steps, dedup, B = arg(1, 512), arg(2, 3), arg(3, 1) != 0, arg(4, 16)
n, ch, blocks, sims, topk, lanes = arg(5, 9), arg(6, 192), arg(7, 12), arg(8, 256), arg(9, 48), arg(10, 1)
torch.manual_seed(0)
net = AlphaZeroNet(board_size=n, channels=ch, num_blocks=blocks).eval().to("cuda:0")
cfg = SelfPlayConfig(board_size=n, mcts_simulations=sims, mcts_inference_batch=B, mcts_topk_expand=topk)
sp = BatchedSelfPlay(net, cfg, games=G, seed0=1, dedup_leaves=dedup, lanes=lanes, tower_workgroups=None)
sp.begin(num_games=10 ** 9)
for _ in range(2):
    sp.step()
torch.cuda.synchronize()
for k in sp.counters:
    sp.counters[k] = 0
t = time.time()
for _ in range(steps):
    sp.step()
torch.cuda.synchronize()
dt = time.time() - t
c = sp.counters
print(f"board={n} net={ch}x{blocks} G={G} sims={sims} B={B} K={topk} lanes={lanes} dedup={dedup} fused={sp.net.fused} async={sp.async_search}: "
      f"{dt / steps * 1e3:.1f} ms/step, sims/s={c['sims'] / dt:.0f}, nn_rows/s={c['nn_rows'] / dt:.0f}, "
      f"mean_depth={c['depth_sum'] / max(1, c['sims']):.2f}, host rng/policy/samples={c['t_host_rng']:.3f}/{c['t_host_policy']:.3f}/{c['t_samples']:.3f}s")
