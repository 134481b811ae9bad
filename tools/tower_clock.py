"""In-kernel clock of the fused tower under sustained load: python tools/tower_clock.py [rows] [seconds]
with AZAMD_LIB pointing at a -DAZ_CLOCK build (sh tools/build_variant.sh clock -DAZ_CLOCK). Runs back-to-back forwards of
`rows` positions (9x9, 192 channels x 12 blocks) for `seconds`, then reads the per-workgroup ratio of shader cycles
(s_memtime) to 100 MHz real-time ticks (s_memrealtime) of the last launch, and the launch time from HIP events over the last
50 launches. Prints one JSON object."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alphazero_amd import _native as N
from alphazero_amd.model import AlphaZeroNet

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
torch.manual_seed(0)
net = AlphaZeroNet(board_size=9, channels=192, num_blocks=12).eval().to("cuda:0")
nn = net.native("bf16")
f = (torch.rand((rows, 81, 32), device="cuda:0") < 0.3).to(torch.bfloat16)
f[:, :, 21:] = 0
for _ in range(3):
    nn(f)
torch.cuda.synchronize()
t0 = time.perf_counter()
launches = 0
while time.perf_counter() - t0 < seconds:
    for _ in range(20):
        nn(f)
    launches += 20
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    nn(f)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
L = C.CDLL(N.LIB_PATH)
buf = np.zeros(11, np.float64)
n = L.az_debug_tower_clock(buf.ctypes.data_as(C.c_void_p), 11)
flops = 2 * 650.5e6 * rows
print(json.dumps({"rows": rows, "warm_seconds": round(seconds, 1), "launches_before": launches, "ms_per_forward": round(ms, 4),
                  "achieved_TFLOPs": round(flops / ms / 1e9, 1), "workgroups": n,
                  "in_kernel_clock_MHz": {"median": round(float(buf[5]), 1), "p10": round(float(buf[1]), 1), "p90": round(float(buf[9]), 1),
                                          "min": round(float(buf[0]), 1)},
                  "method": "per workgroup: (s_memtime at exit - at entry) / (s_memrealtime at exit - at entry) x 100 MHz, last launch of a "
                            "back-to-back sequence; forward time from HIP events over the 50 launches before it (tower + tails)"}))
