"""Micro-benchmark of the bf16 network forward (rows positions) with HIP events."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet

def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ch = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 9
    torch.manual_seed(0)
    use_se = not (len(sys.argv) > 5 and sys.argv[5] == "nose")      # fifth argument "nose": no squeeze-excite
    net = AlphaZeroNet(board_size=n, channels=ch, num_blocks=blocks, use_se=use_se).eval().to("cuda:0")
    nn = net.native("bf16")
    if __import__("os").environ.get("AZ_BENCH_GRID"):   # limit the resident workgroups (diagnostics)
        nn.set_tower_workgroups(int(__import__("os").environ["AZ_BENCH_GRID"]))
    f = (torch.rand((rows, n * n, 32), device="cuda:0") < 0.3).to(torch.bfloat16)
    f[:, :, 21:] = 0
    for _ in range(3):
        nn(f)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 10
    e0.record()
    for _ in range(iters):
        nn(f)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    macs = {(192, 12, 9): 650.5e6, (64, 4, 9): 25.4e6}.get((ch, blocks, n))
    if macs is None:
        macs = n * n * 9 * 21 * ch + blocks * 2 * n * n * 9 * ch * ch
    fl = 2 * macs * rows
    print(f"rows={rows} ch={ch} blocks={blocks} board={n} fused={nn.fused}: {ms:.3f} ms/forward, {rows/ms*1e3:.0f} pos/s, {fl/ms/1e9:.1f} TFLOP/s")

main()
