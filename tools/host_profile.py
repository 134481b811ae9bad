"""cProfile of the host side of lockstep self-play (where the Python time of a move goes):
python tools/host_profile.py GAMES STEPS LANES [CHANNELS BLOCKS SIMS]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
G, steps, lanes, ch, blocks, sims = arg(1, 512), arg(2, 30), arg(3, 4), arg(4, 64), arg(5, 4), arg(6, 192)
torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=ch, num_blocks=blocks).eval().to("cuda:0")
cfg = SelfPlayConfig(board_size=9, mcts_simulations=sims, mcts_inference_batch=16, mcts_topk_expand=48)
sp = BatchedSelfPlay(net, cfg, games=G, seed0=1, lanes=lanes, tower_workgroups=None)
sp.begin(num_games=10 ** 9)
for _ in range(3):
    sp.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    sp.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
