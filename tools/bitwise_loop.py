"""Repeats the variant-agreement check of the fused tower (determinism probe): python tools/bitwise_loop.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_network as T
import stubnets
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kw = dict(board_size=9, channels=192, num_blocks=3)
net = T._default_init(kw, seed=8)
x = stubnets.synthetic_planes(1400, 21, 9, seed=4)
dev = torch.device("cuda:0")
nn = net.to(dev).native("bf16", shared=False)
os.environ["AZ_TOWER_DUO"] = "0"
nn256 = net.native("bf16", shared=False)
del os.environ["AZ_TOWER_DUO"]
f = T._to_nhwc_bf16(x).to(dev)
if os.environ.get("AZ_LOOP_GRID"):   # limit the resident workgroups (one per compute unit: 128 for the two-per-CU kernel)
    nn.set_tower_workgroups(int(os.environ["AZ_LOOP_GRID"]))
ref_lg, ref_v = nn(f); ref_lg, ref_v = ref_lg.clone(), ref_v.clone()
bad = 0
for r in range(reps):
    for name, fn, sl in (("duo", nn, slice(0, 1400)), ("256", nn256, slice(0, 1400)), ("small", nn, slice(700, 1100)), ("small1", nn, slice(0, 1))):
        lg, v = fn(f[sl].contiguous())
        if not (torch.equal(lg, ref_lg[sl]) and torch.equal(v, ref_v[sl])):
            d = (lg - ref_lg[sl]).abs()
            rows = torch.nonzero(d.amax(1) > 0).flatten()
            print(f"rep {r} {name}: {len(rows)} rows differ, first {rows[:8].tolist()}, max {float(d.max()):.3g}")
            bad += 1
print("mismatches:", bad, "of", reps * 4)
