"""Quick end-to-end lockstep self-play timing (development aid)."""
import sys, time
import torch
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig

G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dedup = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
torch.manual_seed(0)
net = AlphaZeroNet().eval().to("cuda:0")
cfg = SelfPlayConfig(mcts_simulations=256, mcts_inference_batch=B, mcts_topk_expand=48)
sp = BatchedSelfPlay(net, cfg, games=G, seed0=1, dedup_leaves=dedup)
sp.run(max_steps=1)
torch.cuda.synchronize()
for k in sp.counters: sp.counters[k] = 0
sp.net.profile(True)
t = time.time()
sp.run(max_steps=steps)
torch.cuda.synchronize()
dt = time.time() - t
c = sp.counters
ms, ln, rw = sp.net.profile(False, read=True)
print(f"G={G} steps={steps} dedup={dedup} B={B}: {dt:.2f}s  sims/s={c['sims']/dt:.0f} nn_rows/s={c['nn_rows']/dt:.0f} "
      f"rows_nodedup={c['nn_rows_nodedup']} rows={c['nn_rows']} tower_ms={ms:.1f} ({ms/dt/10:.1f}% of wall) launches={ln} "
      f"host_rng={c['t_host_rng']:.3f}s host_policy={c['t_host_policy']:.3f}s samples={c['t_samples']:.3f}s "
      f"mean_depth={c['depth_sum']/max(1,c['sims']):.2f}")
