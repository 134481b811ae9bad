"""Direct forward of `live` boards against az_net_forward_indirect over an upper bound `bound` with `live` on the device
(layer path, 19x19 256 channels x 20 blocks): python tools/indirect_probe.py [live] [bound]; run under tools/kstats.sh for
per-kernel times."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet

live = int(sys.argv[1]) if len(sys.argv) > 1 else 508
bound = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n, ch, blocks = 19, 256, 20
torch.manual_seed(0)
dev = torch.device("cuda:0")
net = AlphaZeroNet(board_size=n, channels=ch, num_blocks=blocks).eval().to(dev)
nn = net.native("bf16", shared=False)
nn.reserve(bound)
f = (torch.rand((bound, n * n, 32), device=dev) < 0.3).to(torch.bfloat16)
f[:, :, 21:] = 0
rows_dev = torch.tensor([live], dtype=torch.int32, device=dev)
lb = torch.zeros((bound, n * n + 1), dtype=torch.float32, device=dev)
vb = torch.zeros((bound,), dtype=torch.float32, device=dev)
mode = sys.argv[3] if len(sys.argv) > 3 else "both"
for name in (("direct", "indirect") if mode == "both" else (mode,)):
    def run():
        if name == "direct":
            return nn(f[:live])
        nn.forward_indirect(f, rows_dev.data_ptr(), bound, lb, vb)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: live {live}, bound {bound}: {e0.elapsed_time(e1) / 10:.3f} ms per forward (passes {nn.indirect_passes(bound)})")
