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
for move in (1, 40, 80, 110):
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
    ok = t[:, 7] > 0
    print(f"   waves above 1.5x the median total: {int((tot[ok] > 1.5 * np.median(tot[ok])).sum())} of {int(ok.sum())}; p50 {np.median(tot[ok]):.1f} p90 {np.percentile(tot[ok], 90):.1f} p99 {np.percentile(tot[ok], 99):.1f}")
    for i, nm in enumerate(names[:7]):
        print(f"   {nm:22s} mean {dt[ok, i].mean():6.2f}  p90 {np.percentile(dt[ok, i], 90):6.2f}  max {dt[ok, i].max():6.2f}")
    f = buf.reshape(n, 16)[:, 8:16].astype(np.int64)
    okf = (f > 0).all(axis=1)
    if okf.any():     # shader cycles from the start of one descent to the start of the next, first eight descents of the batch
        d2 = np.diff(f[okf], axis=1)
        print("   descents 1-7, median shader cycles each: " + " ".join(str(int(np.median(d2[:, i]))) for i in range(7))
              + "; p90: " + " ".join(str(int(np.percentile(d2[:, i], 90))) for i in range(7)))
        print(f"   start of the descents after the kernel's first stamp: {np.median((t[okf, 4] - t[okf, 0]) / 100.0):.2f} us (real time)")
    if hasattr(L, "az_debug_einfo"):     # the slowest waves: what each of their first 16 descents did
        info = np.zeros(n * 16, np.uint32)
        assert L.az_debug_einfo(info.ctypes.data_as(C.c_void_p), n * 16) == 0
        info = info.reshape(n, 16)
        for gi in np.argsort(-tot)[:4]:
            desc = " ".join(f"{int(v & 255)}{'T' if v & 256 else ''}{'R' if v & 512 else ''}{'+%dL' % ((v >> 12) & 15) if (v >> 12) & 15 else ''}{'+%dm' % ((v >> 16) & 255) if (v >> 16) & 255 else ''}{'/g%d' % ((v >> 24) & 15) if (v >> 24) & 15 else ''}{'/d%d' % (v >> 28) if v >> 28 else ''}" for v in info[gi])
            if os.environ.get("AZ_FINE") == "1":    # -DAZ_ESTAMP_FINE build: shader cycles inside the fourth descent
                ff = f[gi]
                print(f"   wave {gi}: fourth descent: levels {ff[4] - ff[3]}, leaf kind {ff[5] - ff[4]}, record {ff[6] - ff[5]}, virtual loss {ff[7] - ff[6]} cycles")
            print(f"   wave {gi}: {tot[gi]:.1f} us, descents {dt[gi, 4]:.1f} us, expand {dt[gi, 2]:.1f}, backup {dt[gi, 3]:.1f}; path lengths (T terminal, R replayed, +nL levels loaded, +nm nodes from memory, /gn picks the guess did not settle, /dn picks in float64): {desc}")
