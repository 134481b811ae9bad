#!/usr/bin/env python3
"""Headline benchmark: lockstep self-play on MI355X, 9x9, 192ch x 12 SE blocks, 256 sims/move,
top-k 48, virtual loss 1.0, 4096 games per GPU, random-init weights (synthetic).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step is one lockstep move of every game on the rank: G x 256 simulations through rules, search and
network. Rank 0 prints ONE JSON line. `value` = simulations of all ranks / max-over-ranks wall time.
The roofline entry is for the dominant kernel (the fused bf16 MFMA tower), timed with HIP events on
its launch stream inside the timed region. The cpu_baseline entry times the CPU oracle (C rules +
search, PyTorch-CPU network, one process per core with one torch thread each, like the upstream
multi-worker self-play) on a bounded sample of the same workload, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIMS, TOPK, VLOSS, CH, BLOCKS = 256, 48, 1.0, 192, 12
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X


def tower_flops_per_position(n=9, cin=21, ch=CH, blocks=BLOCKS):
    """Algorithmic FLOPs (2 per MAC) of what the tower kernel computes for one position: stem 3x3,
    2 x blocks 3x3 convolutions, SE (2 small FCs + pool), the two 1x1 head convolutions."""
    np_ = n * n
    macs = np_ * 9 * cin * ch + blocks * 2 * np_ * 9 * ch * ch
    hid = max(8, ch // 8)
    macs += blocks * (2 * ch * hid + np_ * ch)          # SE fcs + pooling adds (counted as MAC-equivalents)
    macs += 2 * np_ * ch * (ch // 4)                     # policy_head.0 and value_head.0
    return 2.0 * macs


def tower_traffic_per_launch(rows_per_launch):
    """HBM-side bytes per tower launch from the committed PMC summary (separate FETCH_SIZE / WRITE_SIZE passes of
    this benchmark, collected offline: counters cannot be read from inside the run), scaled by rows."""
    path = os.path.join(ROOT, "profiles", "r01h_tower_hbm_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return round(d["bytes_per_launch"] / d.get("rows_per_launch", 1364.3) * rows_per_launch)
    except (OSError, KeyError, ValueError):
        return None


def cpu_worker(args):
    """One CPU self-play worker: oracle search + torch CPU network, 1 thread (train.py:357-392 topology)."""
    wid, moves, batch, seed = args
    import numpy as np
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(0)
    net = AlphaZeroNet(channels=CH, num_blocks=BLOCKS).eval()

    def ev(x):
        with torch.inference_mode():
            lg, v = net(torch.from_numpy(x))
        return lg.numpy(), v.numpy()

    rng = np.random.RandomState(seed + wid)
    st = oracle.OracleState.new_game(9, 5.5, 8)
    m = oracle.OracleMCTS(ev, board_size=9, num_simulations=SIMS, inference_batch_size=batch, topk_expand=TOPK,
                          virtual_loss=VLOSS)
    ev(st.to_tensor()[None])          # warm up
    t0 = time.perf_counter()
    sims = 0
    for _ in range(moves):
        pi, a, _ = m.run(st, temperature=1.0, add_exploration_noise=True, rng=rng)
        st = st.play(a)
        sims += SIMS
    return sims, time.perf_counter() - t0, m.stats()["rows"]


def cpu_baseline(batch, moves=2):
    import multiprocessing as mp
    from oracle import oracle
    oracle.build()
    cores = min(16, os.cpu_count() or 1)
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(cpu_worker, [(w, moves, batch, 1234) for w in range(cores)])
    wall = time.perf_counter() - t0
    sims = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": round(sims / busy, 1), "unit": "sims/s", "cores": cores, "kind": "port",
            "sample": f"{cores} workers x {moves} moves x {SIMS} sims (inference batch {batch}, every leaf evaluated, "
                      f"{sum(r[2] for r in res)} network rows), opening position, torch CPU 1 thread/worker; "
                      f"{busy:.1f}s search time, {wall:.1f}s incl. worker start-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=4096, help="lockstep games per GPU")
    ap.add_argument("--batch", type=int, default=16, help="mcts inference batch (README value 16; upstream default 8)")
    ap.add_argument("--no-dedup", action="store_true", help="evaluate repeated leaves of a batch again, as upstream does")
    ap.add_argument("--lanes", type=int, default=3, help="independent game lanes (engine + stream each) whose kernels and host work overlap")
    ap.add_argument("--tower-workgroups", type=int, default=0, help="workgroup limit per tower launch (0: one per CU with 1 lane, 1024 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-moves", type=int, default=8)
    a = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig

    torch.manual_seed(0)
    net = AlphaZeroNet(channels=CH, num_blocks=BLOCKS).eval().to(torch.device("cuda", local))
    cfg = SelfPlayConfig(mcts_simulations=SIMS, mcts_inference_batch=a.batch, mcts_topk_expand=TOPK, mcts_virtual_loss=VLOSS)
    twg = a.tower_workgroups if a.tower_workgroups > 0 else (None if a.lanes == 1 else 1024)
    sp = BatchedSelfPlay(net, cfg, games=a.games, seed0=1 + rank * a.games, dedup_leaves=not a.no_dedup, lanes=a.lanes,
                         tower_workgroups=twg)
    sp.begin(num_games=10 ** 9)          # slots refill for as long as the run lasts
    for _ in range(2):                   # set-up, before the W warm-up steps: first-use allocations (head-activation scratch
        sp.step()                        # for the largest batch, torch's allocator pools) and filling the lane pipeline
    for _ in range(a.warmup):
        sp.step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in sp.counters:
        sp.counters[k] = 0
    sp.profile(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sp.step()
    barrier()
    dt = time.perf_counter() - t0
    tower_ms_sum, launches, tower_ms = sp.profile(False, read=True)   # tower_ms: merged (overlap counted once)

    c = sp.counters
    mine = torch.tensor([dt, c["sims"], c["nn_rows"], c["nn_rows_nodedup"], c["moves"], c["games"], c["depth_sum"]],
                        dtype=torch.float64, device=torch.device("cuda", local))
    if dist is not None:
        tmax = mine[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = mine.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max, tot = float(tmax[0]), tot.cpu().tolist()
    else:
        dt_max, tot = dt, mine.cpu().tolist()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    sims, rows, rows_raw, moves, games, depth_sum = tot[1:]
    fl_pos = tower_flops_per_position()
    per_launch_ms = tower_ms / max(1, launches)
    tower_rows = c["nn_rows"]      # this rank's rows: every network row of the window went through a timed tower launch
    achieved = (tower_rows * fl_pos) / (tower_ms * 1e-3) / 1e12 if tower_ms > 0 else 0.0
    out = {
        "metric": "MCTS sims/sec (node) + self-play games/hour, 9x9 192ch×12blk @256 sims, 1/2/4/8 GPU",
        "value": round(sims / dt_max, 1), "unit": "sims/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt_max / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, games from the empty board)",
        "config": {"workload": "9x9 full net 192ch x 12 SE blocks, 256 sims/move, top-k 48, virtual loss 1.0, "
                               f"{a.games} lockstep games per GPU (BASELINE.json configs[2])",
                   "games_per_gpu": a.games, "lanes": a.lanes, "inference_batch": a.batch, "dedup_repeated_leaves": not a.no_dedup, "root_evaluation_reuse": True,
                   "parallelism": f"games sharded over {world} GPU(s), no data-path collective"},
        "sims_per_s": round(sims / dt_max, 1),
        "nn_rows_per_s": round(rows / dt_max, 1),
        "nn_rows_per_s_if_every_leaf_evaluated": round(rows_raw / dt_max, 1),
        "moves_per_s": round(moves / dt_max, 1),
        "games_per_hour_est": round(moves / dt_max * 3600 / 162, 1),
        "games_per_hour_note": "moves/s x 3600 / 162 plies (the 9x9 move cap; random-init self-play runs to the cap or two passes)",
        "games_finished_in_window": int(games),
        "games_per_hour_measured": round(games / dt_max * 3600, 1),   # meaningful for windows of a few hundred steps (--steps 330)
        "mean_leaf_depth": round(depth_sum / max(1.0, sims), 3),
        "roofline": {"bound": "mfma", "kernel": "k_tower<192,2,4,96,2>", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                     "traffic": tower_traffic_per_launch(tower_rows / max(1, launches)),
                     "traffic_note": "bytes per launch on the L2's memory side (Infinity Cache hits included), PMC passes in profiles/r01h_tower_hbm_traffic.json; "
                                     "mostly the bf16 residual stream parked per workgroup between the two convolutions of a block",
                     "launches": int(launches), "avg_launch_ms": round(per_launch_ms, 4),
                     "avg_launch_ms_start_to_end": round(tower_ms_sum / max(1, launches), 4),
                     "achieved_start_to_end": round((tower_rows * fl_pos) / (tower_ms_sum * 1e-3) / 1e12, 1) if tower_ms_sum > 0 else 0.0,
                     "timing": "HIP events around every tower launch on its stream; launches of different lanes overlap, so "
                               "achieved = FLOPs of all launches / length of the union of their intervals and avg_launch_ms = "
                               "that length / launches; avg_launch_ms_start_to_end is the plain mean of (end - start) and "
                               "achieved_start_to_end the rate it gives (a launch shares the CUs with the other lanes' launches for "
                               "most of that time; --lanes 1 makes the two readings coincide)",
                     "rows_per_launch": round(tower_rows / max(1, launches), 1), "flop_per_position": fl_pos,
                     "tower_share_of_wall": round(tower_ms * 1e-3 / dt, 4)},
        "host_seconds": {"rng": round(c["t_host_rng"], 4), "policy": round(c["t_host_policy"], 4), "samples": round(c["t_samples"], 4)},
    }
    if dist is not None:
        dist.destroy_process_group()
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(a.batch, a.cpu_moves)
        except Exception as e:  # the GPU number stands even if the CPU leg cannot run
            out["cpu_baseline"] = {"value": None, "unit": "sims/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
