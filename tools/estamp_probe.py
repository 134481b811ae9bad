"""Where a wave of k_expand_select spends its time: python tools/estamp_probe.py [games] [channels] [blocks] [topk] [batch]
with AZAMD_LIB pointing at a -DAZ_ESTAMP build (sh tools/build_variant_engine.sh estamp -DAZ_ESTAMP). Prints the mean and the
maximum over the games of each phase of the last launch, in microseconds (100 MHz real-time counter)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alphazero_amd import _native as N
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig

games, ch, blocks, topk, batch = (int(x) for x in (sys.argv[1:] + [512, 64, 4, 0, 16][len(sys.argv) - 1:])[:5])
torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=ch, num_blocks=blocks).eval().cuda()
cfg = SelfPlayConfig(board_size=9, mcts_simulations=192, mcts_inference_batch=batch, mcts_topk_expand=topk)
sp = BatchedSelfPlay(net, cfg, games=games, seed0=1, lanes=1)
sp.begin(num_games=10 ** 9)
names = ["record", "values + root loads", "expand", "backup", "descents", "record out", "rows + stats", "store"]
for move in (1, 20, 40):
    while sp.counters["steps"] < move:
        sp.step()
    n = min(games, 4096)
    buf = np.zeros(n * 16, np.uint64)
    L = C.CDLL(N.LIB_PATH)
    assert L.az_debug_estamps(buf.ctypes.data_as(C.c_void_p), n * 16) == 0
    t = buf.reshape(n, 16)[:, :8].astype(np.int64)
    dt = np.diff(t, axis=1) / 100.0
    tot = (t[:, 7] - t[:, 0]) / 100.0
    print(f"move {move}: total mean {tot.mean():.1f} us, max {tot.max():.1f} us")
    for i, nm in enumerate(names[:7]):
        print(f"   {nm:22s} mean {dt[:, i].mean():6.2f}  max {dt[:, i].max():6.2f}")
    f = buf.reshape(n, 16)[:, 8:14].astype(np.int64)
    if f[:, 0].any():     # the second descent of the batch in detail, and the shader clock over it
        d2 = np.diff(f[:, :4], axis=1) / 100.0
        clk = (f[:, 5] - f[:, 4]) / np.maximum(1, f[:, 3] - f[:, 0]) * 100.0
        print(f"   descent 2: child choice {d2[:, 0].mean():5.2f}  leaf bookkeeping {d2[:, 1].mean():5.2f}  virtual loss {d2[:, 2].mean():5.2f} us; "
              f"shader clock {np.median(clk):.0f} MHz")
