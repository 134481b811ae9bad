import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
torch.manual_seed(0)
net = AlphaZeroNet(channels=192, num_blocks=12).eval().cuda()
cfg = SelfPlayConfig(mcts_simulations=256, mcts_inference_batch=16, mcts_topk_expand=48, mcts_virtual_loss=1.0)
sp = BatchedSelfPlay(net, cfg, games=4096, seed0=1, lanes=3, tower_workgroups=1024)
sp.begin(num_games=10**9)
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sp.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(i, round(dt*1e3, 1), {k: round(v, 3) for k, v in sp.counters.items() if k.startswith("t_")}, flush=True)
