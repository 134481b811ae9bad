"""Do the fused towers read LDS or scratch memory they did not write? python tools/poison_check.py save|compare <file>
`save` with the regular library, `compare` with a -DAZ_POISON build (sh tools/build_variant.sh poison -DAZ_POISON; AZAMD_LIB):
that build fills a workgroup's LDS allocation and the global scratch with NaN patterns before the tower runs, so any output
that depended on leftovers (whose content depends on which workgroup ran there before: timing, with two workgroups per
compute unit) changes. Three kernel variants x three shapes, outputs compared bit for bit; the poisoned build also counts
residual-scratch slots that two workgroups held at once."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import stubnets
from alphazero_amd import _native as N
from alphazero_amd.model import AlphaZeroNet

mode, path = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
out = {}
clashes = []
L = C.CDLL(N.LIB_PATH)
for name, kw, rows in (("n9_c192", dict(board_size=9, channels=192, num_blocks=4), 1100), ("n9_c128", dict(board_size=9, channels=128, num_blocks=3), 900),
                       ("n5_c64", dict(board_size=5, channels=64, num_blocks=2), 700), ("n13_c192", dict(board_size=13, channels=192, num_blocks=2), 300)):
    torch.manual_seed(4)
    net = AlphaZeroNet(**kw).eval().to(dev)
    n = kw["board_size"]
    x = stubnets.synthetic_planes(rows, 21, n, seed=9)
    f = torch.zeros((rows, n * n, 32), dtype=torch.bfloat16)
    f[:, :, :21] = torch.from_numpy(x).reshape(rows, 21, n * n).permute(0, 2, 1).to(torch.bfloat16)
    f = f.contiguous().to(dev)
    for variant, env in (("default", {}), ("256_row", {"AZ_TOWER_DUO": "0"}), ("small", {"AZ_TOWER_SMALL_ROUNDS": "64"})):
        for k, v in env.items():
            os.environ[k] = v
        nn = net.native("bf16", shared=False)
        for k in env:
            del os.environ[k]
        for rep in range(3):
            lg, v = nn(f)
            out[f"{name}/{variant}/{rep}/lg"] = lg.cpu().numpy()
            out[f"{name}/{variant}/{rep}/v"] = v.cpu().numpy()
            if mode == "compare":
                clashes.append(L.az_debug_slot_clashes())
if mode == "save":
    np.savez(path, **out)
    print("saved", len(out), "arrays")
else:
    ref = np.load(path)
    bad = [k for k in out if out[k].tobytes() != ref[k].tobytes()]
    print(f"poisoned LDS + scratch: {len(out) - len(bad)} of {len(out)} outputs bit-equal to the regular build; mismatches: {bad[:6]}; "
          f"slot clashes over {len(clashes)} launches: {sum(c for c in clashes if c > 0)} (query failures: {sum(1 for c in clashes if c < 0)})")
    sys.exit(1 if bad or any(c != 0 for c in clashes) else 0)
