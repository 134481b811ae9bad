#!/usr/bin/env python3
"""Wider sweep of the device-vs-oracle parity tests than the collected suite runs (more seeds, board sizes and
search settings). Not collected by pytest; run by hand on a GPU box:  python -u tests/soak_gpu.py
Prints a line per case, exits non-zero at the first difference."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle  # noqa: E402  (test infrastructure)
import test_gpu_rules as R  # noqa: E402
import test_gpu_search as S  # noqa: E402


def main():
    oracle.build()
    t0 = time.time()
    for n, games, seed in [(5, 256, 11), (5, 256, 12), (7, 128, 13), (9, 256, 14), (9, 256, 15), (11, 64, 16), (13, 48, 17), (19, 24, 18)]:
        R.test_lockstep_random_games_vs_oracle(oracle, n, games, seed)
        print(f"rules n={n} games={games} seed={seed}: ok ({time.time() - t0:.0f}s)", flush=True)
    for n, kind, sims, B, K, noise in [(9, "sharp", 96, 8, 16, True), (7, "hash", 50, 2, 0, True), (19, "hash", 16, 4, 48, True),
                                       (5, "uniform", 33, 16, 6, False), (9, "uniform", 256, 16, 48, True), (9, "hash", 192, 8, 0, True),
                                       (11, "flatish", 40, 1, 0, True), (13, "sharp", 64, 16, 48, False)]:
        S.test_lockstep_search_vs_oracle(oracle, n, kind, sims, B, K, noise)
        print(f"search n={n} {kind} sims={sims} B={B} K={K} noise={noise}: ok ({time.time() - t0:.0f}s)", flush=True)
    print("soak ok")


if __name__ == "__main__":
    main()
