"""Sustained large bf16 GEMM of the vendor library (torch.matmul) for a few seconds while amd-smi samples socket power and
XCD clocks: whether the library's best plain GEMM runs into the same power limit as the fused tower.
  python tools/gemm_power.py [seconds]"""
import subprocess, sys, threading, time
import torch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
samples, stop = [], False
def watch():
    while not stop:
        try:
            out = subprocess.run(["amd-smi", "metric", "-p", "-c"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            break
        p = [l.split()[1] for l in out.splitlines() if "SOCKET_POWER" in l]
        clk, g = [], False
        for l in out.splitlines():
            if "GFX_" in l and ":" in l: g = True
            elif g and " CLK:" in l: clk.append(int(l.split()[1])); g = False
        if p: samples.append((float(p[0]), clk[:8]))
        time.sleep(0.3)
m = n = 16384; k = 8192
x = torch.randn((m, k), device="cuda", dtype=torch.bfloat16); w = torch.randn((k, n), device="cuda", dtype=torch.bfloat16)
out = torch.empty((m, n), device="cuda", dtype=torch.bfloat16)
for _ in range(3): torch.matmul(x, w, out=out)
torch.cuda.synchronize()
t = threading.Thread(target=watch); t.start()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time(); iters = 0
a.record()
while time.time() - t0 < secs:
    for _ in range(20): torch.matmul(x, w, out=out)
    iters += 20
    torch.cuda.synchronize()
b.record(); torch.cuda.synchronize()
stop = True; t.join()
dt = a.elapsed_time(b) * 1e-3 / iters
busy = [s for s in samples if s[0] > 600]
print(f"gemm 16384x16384x8192 bf16: {2.0 * m * n * k / dt / 1e12:.1f} TFLOP/s sustained over {secs:.0f} s")
if busy:
    print(f"socket power {min(s[0] for s in busy):.0f}-{max(s[0] for s in busy):.0f} W over {len(busy)} samples; XCD clocks "
          f"{min(min(s[1]) for s in busy if s[1])}-{max(max(s[1]) for s in busy if s[1])} MHz")
