import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from alphazero_amd.engine import Engine
from alphazero_amd.model import AlphaZeroNet
torch.manual_seed(0)
net = AlphaZeroNet().eval().to("cuda:0").native("bf16")
for G, B in ((1, 8), (2, 8), (40, 8)):
    eng = Engine(board_size=9, games=G, num_simulations=256, inference_batch_size=B, topk_expand=0)
    for mode in ("stepwise", "enqueued"):
        nn = None if mode == "stepwise" else net
        eng.run_search(net.forward, layout=net.layout, native_net=nn)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            v, r = eng.run_search(net.forward, layout=net.layout, native_net=nn)
            v.cpu()
        dt = (time.perf_counter() - t) / 5
        print(f"G={G} B={B} sims=256 {mode}: {dt*1e3:.2f} ms per search")
    eng.close()
