#!/usr/bin/env python3
"""What this MI355X sustains on the two textbook kernels the roofline peaks stand for: a large bf16 GEMM
(rocBLAS / hipBLASLt through torch.matmul) against the 2.5 PFLOP/s dense MFMA figure, and a device-to-device
copy against the 8 TB/s HBM3E figure. Prints one JSON line; numbers are context for roofline.frac, the
denominators stay the vendor peaks.

  python tools/peaks.py [--out profiles/peaks.json]
"""
import argparse
import json

import torch


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = {"device": torch.cuda.get_device_name(0), "gemm_bf16": [], "copy": []}
    for m, n, k in ((4096, 4096, 4096), (8192, 8192, 8192), (16384, 16384, 8192), (65536, 192, 1728), (1048576, 192, 1728)):
        x = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
        w = torch.randn((k, n), device=dev, dtype=torch.bfloat16)
        out = torch.empty((m, n), device=dev, dtype=torch.bfloat16)
        dt = timed(lambda: torch.matmul(x, w, out=out), 20)
        res["gemm_bf16"].append({"m": m, "n": n, "k": k, "ms": round(dt * 1e3, 4), "tflops": round(2.0 * m * n * k / dt / 1e12, 1)})
        del x, w, out
    for gib in (1, 4):
        src = torch.empty(gib << 30, device=dev, dtype=torch.uint8).fill_(1)
        dst = torch.empty_like(src)
        dt = timed(lambda: dst.copy_(src), 20)
        res["copy"].append({"GiB": gib, "ms": round(dt * 1e3, 4), "read_plus_write_TBps": round(2.0 * (gib << 30) / dt / 1e12, 3)})
        del src, dst
    res["note"] = ("the last two GEMM shapes are one residual-block convolution of the 9x9 192-channel tower as an explicit "
                   "im2col GEMM (positions x 192 x 9*192) for 809 and 12945 boards")
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
