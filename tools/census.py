"""Un-overlapped timings of the search / rules kernels (bench.py's kernel census) on their own:
python tools/census.py [games]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alphazero_amd.model import AlphaZeroNet
games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=bench.CH, num_blocks=bench.BLOCKS).eval().to("cuda:0")
plies = int(sys.argv[2]) if len(sys.argv) > 2 else 24
out, setup = bench.kernel_census(net, 16, games, plies)
print(" | ".join(f"{o['kernel']}: {o['avg_launch_us']} us" for o in out), setup)
