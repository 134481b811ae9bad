"""k_select timing across consecutive searches on one engine with a stand-in evaluator (random logits):
python tools/sel_probe2.py [noise]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphazero_amd import _native as N
from alphazero_amd.engine import Engine
G = 4096
eng = Engine(board_size=9, games=G, num_simulations=256, inference_batch_size=16, topk_expand=48, virtual_loss=1.0)
g = torch.Generator(device="cuda").manual_seed(0)
rng = np.random.default_rng(0)
ev = lambda: torch.cuda.Event(enable_timing=True)
for move in range(4):
    eng.search_begin()
    rows = eng.pending()
    eng.gather(rows, N.AZ_FEAT_BF16_NHWC)
    lg = torch.randn((rows, 82), device="cuda", generator=g) * 0.05
    v = torch.zeros(rows, device="cuda")
    noise = None
    if "noise" in sys.argv:
        noise = torch.from_numpy(rng.dirichlet([0.3] * 82, size=G))
    done = eng.expand(lg, v, noise)
    ts = []
    while not done:
        a, b = ev(), ev()
        a.record(); eng.search_select(); b.record()
        rows = eng.pending()
        eng.gather(rows, N.AZ_FEAT_BF16_NHWC)
        lg = torch.randn((rows, 82), device="cuda", generator=g) * 0.05
        v = torch.randn(rows, device="cuda", generator=g) * (0.0 if "zerov" in sys.argv else 0.05)
        done = eng.expand(lg, v)
        ts.append(int(a.elapsed_time(b) * 1e3))
    visits, rv = eng.result()
    st = eng.stats()
    print("move", move, "select us:", ts, "depth", st["depth_sum"] / st["sims"])
    acts = visits.argmax(1).to(torch.int32)
    eng.play(acts)
