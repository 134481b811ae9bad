#!/usr/bin/env python3
"""Headline benchmark: lockstep self-play on MI355X, 9x9, 192ch x 12 SE blocks, 256 sims/move,
top-k 48, virtual loss 1.0, 4096 games per GPU, random-init weights (synthetic).

  python bench.py --gpus N --steps K --warmup W

N > 1: unless it already runs under torch.distributed.run (RANK in the environment), the script starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a child process
before it touches the GPU and relays rank 0's line; one rank per GPU, RCCL ("nccl") for the barriers and the
counter sums, no collective on the data path.

A step is one lockstep move of every game on the rank: G x 256 simulations through rules, search and
network. Rank 0 prints ONE JSON line. `value` = simulations of all ranks / max-over-ranks wall time.
`roofline` is for the dominant kernel (the fused bf16 MFMA tower), timed with HIP events on its launch
stream inside the timed region; `roofline_other` gives the HBM-side figures of the search / rules kernels
from an un-overlapped pass after the timed region. `cpu_baseline` times the CPU oracle (C rules + search,
PyTorch-CPU network, one process per core with one torch thread each, like the upstream multi-worker
self-play) on a bounded sample of the same workload, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIMS, TOPK, VLOSS, CH, BLOCKS = 256, 48, 1.0, 192, 12       # the headline workload (BASELINE.json configs[2])
# --config N selects one of BASELINE.json's other GPU configurations (1-based, as the survey numbers them); they are
# parity-test shapes whose lines are kept under profiles/, not the headline
WORKLOADS = {
    3: dict(board=9, ch=192, blocks=12, sims=256, topk=48, games=4096, lanes=2, batch=16,
            text="9x9 full net 192ch x 12 SE blocks, 256 sims/move, top-k 48, virtual loss 1.0, {g} lockstep games per GPU (BASELINE.json configs[2])"),
    2: dict(board=9, ch=64, blocks=4, sims=192, topk=0, games=512, lanes=2, batch=16,
            text="9x9 small net 64ch x 4 blocks, 192 sims/move, {g} lockstep games per GPU (BASELINE.json configs[1])"),
    5: dict(board=19, ch=256, blocks=20, sims=800, topk=48, games=512, lanes=2, batch=16,
            text="19x19 256ch x 20 SE blocks, 800 sims/move, history 8, top-k 48, {g} lockstep games per GPU (per-GPU share of BASELINE.json configs[4])"),
}
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X
PEAK_HBM_TBPS = 8.0         # HBM3E peak


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
    for name in ("r03f_tower_hbm_traffic.json", "r02_tower_hbm_traffic.json", "r01h_tower_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            return round(d["bytes_per_launch"] / d.get("rows_per_launch", 1364.3) * rows_per_launch), name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


# ------------------------------------------------------------------------------------------------ CPU leg
def cpu_worker(args):
    """One CPU self-play worker: oracle search + torch CPU network, 1 thread (train.py:357-392 topology).
    dedup: rows of a batch with identical feature planes (the same leaf reached by several descents) are
    evaluated once, the policy the GPU headline uses; otherwise every leaf is evaluated, as upstream does."""
    wid, budget_s, batch, seed, dedup, plies = args
    import numpy as np
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(0)
    net = AlphaZeroNet(channels=CH, num_blocks=BLOCKS).eval()
    rows_eval = [0]

    def ev(x):
        if dedup and x.shape[0] > 1:
            flat = x.reshape(x.shape[0], -1)
            _, first, inv = np.unique(flat, axis=0, return_index=True, return_inverse=True)
            with torch.inference_mode():
                lg, v = net(torch.from_numpy(x[first]))
            rows_eval[0] += len(first)
            return lg.numpy()[inv.reshape(-1)], v.numpy()[inv.reshape(-1)]
        with torch.inference_mode():
            lg, v = net(torch.from_numpy(x))
        rows_eval[0] += x.shape[0]
        return lg.numpy(), v.numpy()

    rng = np.random.RandomState(seed + wid)
    st = oracle.OracleState.new_game(9, 5.5, 8)
    for _ in range(plies):                # a mid-game root: uniformly random legal stones
        stones = np.nonzero(st.legal_mask()[:-1])[0]
        if len(stones) == 0 or st.is_terminal():
            break
        st = st.play(int(rng.choice(stones)))
    m = oracle.OracleMCTS(ev, board_size=9, num_simulations=SIMS, inference_batch_size=batch, topk_expand=TOPK,
                          virtual_loss=VLOSS)
    ev(st.to_tensor()[None])          # warm up
    rows_eval[0] = 0
    t0 = time.perf_counter()
    sims = moves = 0
    while time.perf_counter() - t0 < budget_s and not st.is_terminal():
        pi, a, _ = m.run(st, temperature=1.0, add_exploration_noise=True, rng=rng)
        st = st.play(a)
        sims += SIMS
        moves += 1
    return sims, time.perf_counter() - t0, rows_eval[0], moves


def cpu_baseline(batch, budget_s=10.0):
    import multiprocessing as mp
    from oracle import oracle
    oracle.build()
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)      # the GPU box exports its per-GPU CPU share here
    cores = max(1, min(visible, share) if share > 0 else min(visible, 16))
    ctx = mp.get_context("spawn")
    legs = {}
    with ctx.Pool(cores) as pool:
        for name, dedup in (("every_leaf", False), ("dedup", True)):
            res = pool.map(cpu_worker, [(w, budget_s, batch, 1234, dedup, (w * 7) % 64) for w in range(cores)])
            sims, busy, rows, moves = sum(r[0] for r in res), max(r[1] for r in res), sum(r[2] for r in res), sum(r[3] for r in res)
            legs[name] = {"sims_per_s": round(sims / busy, 1), "nn_rows": rows, "moves": moves, "seconds": round(busy, 1)}
    return {"value": legs["every_leaf"]["sims_per_s"], "unit": "sims/s", "cores": cores, "kind": "port",
            "value_dedup_policy": legs["dedup"]["sims_per_s"],
            "host_cpus_visible": visible,
            "sample": f"{cores} workers (one process per core, 1 torch thread each) x {budget_s:.0f} s of whole moves x {SIMS} sims, inference batch "
                      f"{batch}, roots after 0..63 random plies; leg 1 evaluates every leaf as upstream does "
                      f"({legs['every_leaf']['moves']} moves, {legs['every_leaf']['nn_rows']} network rows) -> value; leg 2 evaluates a leaf "
                      f"reached by several descents of a batch once, the GPU headline's policy ({legs['dedup']['moves']} moves, "
                      f"{legs['dedup']['nn_rows']} rows) -> value_dedup_policy; cores = min(CPUs this process may use, the box's per-GPU CPU share)",
            "reference_python_measured_in_build_container": "the upstream Python itself, timed where it can be imported (it cannot travel to the GPU box): "
                                                            "59.7 sims/s for one 1-thread worker alone; 270 sims/s and 55 games/hour for its own 8-worker "
                                                            "run_selfplay on 8 cores (8 whole games, 549 moves, 520.6 s; profiles/r03b_upstream_python_8_workers_build_container.json)"}


# ------------------------------------------------------------------------------------------------ GPU side legs
def mean_plies_per_game(net, cfg, games=256, seed0=900001):
    """Whole games of the same network and search settings on a small lockstep batch: plies per finished game."""
    from alphazero_amd.selfplay import BatchedSelfPlay
    sp = BatchedSelfPlay(net, cfg, games=games, seed0=seed0, lanes=1, collect_samples=False)
    sp.begin(num_games=games)
    while sp.busy():
        sp.step()
    sp.flush()
    return sp.counters["moves"] / max(1, sp.counters["games"]), sp.counters["games"]


def kernel_census(net, batch, games, plies=24):
    """Un-overlapped timing (events on the launch stream) of the search / rules kernels of one batch round on `games`
    mid-game positions: k_select (+ row scan), k_leaf, k_expand_backup, against their algorithmic HBM bytes."""
    import numpy as np
    import torch
    from alphazero_amd import _native as N
    from alphazero_amd.engine import Engine
    eng = Engine(board_size=9, games=games, num_simulations=SIMS, inference_batch_size=batch, topk_expand=TOPK, virtual_loss=VLOSS)
    rng = np.random.default_rng(7)
    for _ in range(plies):
        lm = eng.legal_mask().cpu().numpy()[:, :-1].astype(np.float64)
        lm[lm.sum(1) == 0] = 1.0
        cdf = np.cumsum(lm / lm.sum(1, keepdims=True), axis=1)
        acts = (cdf < rng.random((games, 1))).sum(1).astype(np.int32)
        eng.play(torch.from_numpy(np.minimum(acts, 80)), raise_on_illegal=False)
    nn = net.native("bf16", shared=False)
    nn.reserve(eng.max_rows)
    eng.search_begin()
    rows = eng.pending()
    feats = eng.gather(rows, N.AZ_FEAT_BF16_NHWC)
    lg, v = nn(feats)
    done = eng.expand(lg, v)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    t_sel = t_leaf = t_exp = 0.0
    n_rounds = leaves = 0
    while not done:
        e = [ev() for _ in range(5)]
        e[0].record(); eng.search_select(); e[1].record()
        rows = eng.pending()
        e[2].record(); feats = eng.gather(rows, N.AZ_FEAT_BF16_NHWC); e[3].record()
        lg, v = nn(feats)
        e[3].synchronize()
        t0 = ev(); t1 = ev()
        t0.record(); done = eng.expand(lg, v); t1.record(); t1.synchronize()
        t_sel += e[0].elapsed_time(e[1]); t_leaf += e[2].elapsed_time(e[3]); t_exp += t0.elapsed_time(t1)
        if os.environ.get("AZ_CENSUS_VERBOSE"):
            print(f"round {n_rounds}: select {e[0].elapsed_time(e[1]) * 1e3:.0f} us, leaf {e[2].elapsed_time(e[3]) * 1e3:.0f} us, expand {t0.elapsed_time(t1) * 1e3:.0f} us, rows {rows}", file=sys.stderr)
        n_rounds += 1
        leaves += rows
    st = eng.stats()
    eng.close()
    sims = st["sims"]
    depth = st["depth_sum"] / max(1, sims)
    kids = TOPK
    b_select = sims * (depth * (kids * 26 + 12))                                  # children records read per level + path write
    b_leaf = leaves * (81 + 8 * 81 + 8 * (plies + 1 + depth) + 81 * 32 * 2 + 82)  # board + history + seen keys in, planes + mask out
    b_expand = leaves * (82 * 4 + 82 + kids * 26) + sims * (depth + 1) * 2 * 12   # logits + mask in, child records out; path read-modify-write
    out = []
    for name, ms, by, unit in (("k_select (+ k_scan_rows)", t_sel, b_select, f"{sims // n_rounds} sims/launch"),
                               ("k_leaf", t_leaf, b_leaf, f"{leaves // n_rounds} leaves/launch"),
                               ("k_expand_backup", t_exp, b_expand, f"{leaves // n_rounds} leaves + {sims // n_rounds} backups/launch")):
        tb = by / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        out.append({"kernel": name, "bound": "hbm", "avg_launch_us": round(ms / n_rounds * 1e3, 1), "algorithmic_bytes_per_launch": int(by / n_rounds),
                    "achieved_TBps": round(tb, 4), "peak_TBps": PEAK_HBM_TBPS, "frac": round(tb / PEAK_HBM_TBPS, 4), "units": unit})
    return out, {"games": games, "root_plies": plies, "mean_leaf_depth": round(depth, 3), "rounds": n_rounds}


def train_step_leg(net, dev, dist, world, host_staged, batch=256, steps=20, warmup=3):
    """The data-parallel training step of BASELINE.json configs[3] (go_zero/train.py:504-554: AdamW, cosine LR, AMP, clip 1.0)
    under the benchmark's own process group: batch 256 per GPU of the 192ch x 12 network on synthetic replay samples assembled
    by az_replay_gather, gradients averaged by one flat all-reduce. Returns ms per step (slowest rank), the all-reduce alone
    and samples/s over all ranks. Every rank must call it (it runs collectives)."""
    import copy
    import numpy as np
    import torch
    from alphazero_amd import train as T
    from alphazero_amd.train_step import DataParallelTrainer, all_reduce_
    model = copy.deepcopy(net).to(dev)
    rng = np.random.RandomState(17)
    samples = []
    for _ in range(2048):
        st = (rng.rand(21, 9, 9) < 0.3).astype(np.float32)
        st[-2:] = 0.0
        samples.append((st, rng.dirichlet([0.5] * 82).astype(np.float32), float(rng.choice([-1.0, 1.0])), 0.5))
    buf = T.ReplayBuffer(8192)
    buf.extend(samples)
    buf.attach_device(dev, 9, 8)
    trainer = DataParallelTrainer(model, lr=1e-3, total_steps=steps + warmup, use_amp=True)
    np.random.seed(5); import random as _r; _r.seed(5)

    def one():
        states, pis, zs = buf.sample_device(batch, 9, True)
        return trainer.step(states, pis, zs)

    for _ in range(warmup):
        one()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ar_ms = None
    if dist is not None and trainer._flat is not None:     # the collective alone, on the gradient buffer of the last step
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            all_reduce_(dist, trainer._flat)
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t1) / 10 * 1e3
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if host_staged else dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0])
    params = sum(p.numel() for p in model.parameters())
    del trainer, model, buf
    torch.cuda.empty_cache()
    return {"ms_per_step": round(dt / steps * 1e3, 3), "allreduce_ms": None if ar_ms is None else round(ar_ms, 3),
            "samples_per_s": round(world * batch * steps / dt, 1), "batch_per_gpu": batch, "steps": steps, "amp": True,
            "gradient_bytes": params * 4, "last_loss": round(float(loss[0]), 4),
            "what": "DataParallelTrainer.step on the 192ch x 12 network (upstream train_candidate's step, train.py:504-554), batches "
                    "assembled on the device by az_replay_gather, one flat all-reduce of the gradients per step over the process group "
                    "(RCCL; host-staged gloo in the one-GPU rehearsal); the forward/backward is PyTorch's"}


class PowerWatch:
    """Socket power and XCD clocks of the card (amd-smi, a sample every ~0.25 s from a side thread) while the timed window runs;
    silent when the tool is missing or refuses. The figures say whether the run sits at the card's power limit."""

    def __init__(self, gpu):
        import threading
        self.gpu, self.samples, self.stop = int(gpu), [], False
        self.t0 = self.t1 = None            # the timed steps (set by the window)
        self.thread = threading.Thread(target=self._run, daemon=True)
        # not under a profiler (its preloaded library would ride along into every amd-smi child process)
        self.enabled = not any(k.startswith("ROCPROF") for k in os.environ) and "rocprof" not in os.environ.get("LD_PRELOAD", "")

    def _run(self):
        while not self.stop:
            try:
                out = subprocess.run(["amd-smi", "metric", "-g", str(self.gpu), "-p", "-c"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                return
            power = [ln.split()[1] for ln in out.splitlines() if "SOCKET_POWER" in ln]
            clocks, in_gfx = [], False
            for ln in out.splitlines():
                if "GFX_" in ln and ln.strip().endswith(":"):
                    in_gfx = True
                elif in_gfx and " CLK:" in ln:
                    try:
                        clocks.append(int(ln.split()[1]))
                    except ValueError:
                        pass
                    in_gfx = False
            try:
                if power:
                    self.samples.append((float(power[0]), clocks[:8], time.perf_counter()))
            except ValueError:
                pass
            time.sleep(0.25)

    def __enter__(self):
        if self.enabled:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        if self.enabled:
            self.thread.join(timeout=6)

    def summary(self):
        s = [x for x in self.samples if x[0] > 0 and (self.t0 is None or self.t0 <= x[2] <= (self.t1 or x[2]))]
        if not s:
            return None
        clk = [c for _, cs, _ in s for c in cs]
        return {"samples": len(s), "socket_power_W": [round(min(x[0] for x in s)), round(max(x[0] for x in s))],
                "xcd_clock_MHz": [min(clk), max(clk)] if clk else None,
                "note": "amd-smi metric -p -c sampled during the timed steps; the card's power cap is 1400 W"}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start one rank per GPU as fresh child processes (this process has not touched the GPU) and relay their output."""
    import torch
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < n and "--rehearse-on-one-gpu" not in argv:
        raise SystemExit(f"bench.py --gpus {n}: this host shows {have} GPU(s); refusing to report a {n}-GPU figure from fewer ranks")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=sorted(WORKLOADS), help="BASELINE.json configuration (3 = headline)")
    ap.add_argument("--games", type=int, default=0, help="lockstep games per GPU (0: the configuration's)")
    ap.add_argument("--batch", type=int, default=16, help="mcts inference batch (README value 16; upstream default 8)")
    ap.add_argument("--no-dedup", action="store_true", help="evaluate repeated leaves of a batch again, as upstream does")
    ap.add_argument("--lanes", type=int, default=0, help="independent game lanes (engine + stream each) whose kernels and host work overlap (0: the configuration's)")
    ap.add_argument("--tower-workgroups", type=int, default=0, help="workgroup limit per tower launch (0: the driver's choice: all resident workgroups with one lane, one per CU and launch with two)")
    ap.add_argument("--sync-search", action="store_true", help="one host round trip per inference batch instead of enqueueing a move's batches back to back (A/B timing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="search time per CPU-baseline leg")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the no-dedup window, the kernel census, the plies-per-game run and the training-step leg")
    ap.add_argument("--no-train-step", action="store_true", help="skip the data-parallel training-step leg")
    ap.add_argument("--force-dist", action="store_true", help="N=1: still go through a one-rank RCCL process group")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="test only: all ranks share GPU 0 and synchronise over gloo with host tensors (RCCL refuses two ranks on one "
                         "device); exercises the N-rank control flow on a one-GPU box, its figures mean nothing")
    a = ap.parse_args()
    wl = WORKLOADS[a.config]
    a.games = a.games or wl["games"]
    a.lanes = a.lanes or wl["lanes"]
    headline = a.config == 3

    in_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not in_launcher:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1")) if in_launcher else 1
    rank = int(os.environ.get("RANK", "0")) if in_launcher else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if in_launcher else 0
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was started with WORLD_SIZE={world}: the two must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if a.rehearse_on_one_gpu:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU {local} on this host ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        if not in_launcher:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        world = dist.get_world_size()

    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig

    torch.manual_seed(0)
    dev = torch.device("cuda", local)
    net = AlphaZeroNet(board_size=wl["board"], channels=wl["ch"], num_blocks=wl["blocks"]).eval().to(dev)
    cfg = SelfPlayConfig(board_size=wl["board"], mcts_simulations=wl["sims"], mcts_inference_batch=a.batch, mcts_topk_expand=wl["topk"],
                         mcts_virtual_loss=VLOSS)
    twg = a.tower_workgroups if a.tower_workgroups > 0 else None     # None: the driver's own choice for its lane count

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def window(sp, steps, warmup, setup=2, watch=None):
        sp.begin(num_games=10 ** 9)          # slots refill for as long as the run lasts
        for _ in range(setup + warmup):      # `setup` steps before the W warm-up steps: first-use allocations of torch's
            sp.step()                        # allocator pools and filling the lane pipeline
        for k in sp.counters:
            sp.counters[k] = 0
        sp.profile(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            sp.step()
        barrier()
        dt = time.perf_counter() - t0
        if watch is not None:
            watch.t0, watch.t1 = t0, t0 + dt
        return dt, sp.profile(False, read=True)

    sp = BatchedSelfPlay(net, cfg, games=a.games, seed0=1 + rank * a.games, dedup_leaves=not a.no_dedup, lanes=a.lanes,
                         tower_workgroups=twg)
    if a.sync_search:
        sp.set_async(False)
    # device identity check: N ranks must sit on N different GPUs (a launcher that hands two ranks the same card would report
    # a "scaling" figure from time-sliced devices)
    if dist is not None and world > 1 and not a.rehearse_on_one_gpu:
        props = torch.cuda.get_device_properties(local)
        ident = f"{socket.gethostname()}:{getattr(props, 'uuid', None) or getattr(props, 'pci_bus_id', local)}:{local}"
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if len(set(idents)) != world:
            raise SystemExit(f"bench.py --gpus {world}: ranks share a device ({idents})")

    class _NoWatch:          # ranks other than 0 do not sample the power: one sampler per node is enough
        def __enter__(self): return self
        def __exit__(self, *exc): return False
        def summary(self): return None
        t0 = t1 = None

    with (PowerWatch(local) if rank == 0 else _NoWatch()) as pw:
        dt, (tower_ms_sum, launches, tower_ms) = window(sp, a.steps, a.warmup, watch=pw)   # tower_ms: merged (overlap counted once)

    c = dict(sp.counters)
    mine = torch.tensor([dt, c["sims"], c["nn_rows"], c["nn_rows_nodedup"], c["moves"], c["games"], c["depth_sum"]],
                        dtype=torch.float64, device="cpu" if a.rehearse_on_one_gpu else dev)
    rank_rates = [c["sims"] / dt]
    train_leg = None
    if headline and not a.no_side_legs and not a.no_train_step:
        train_leg = train_step_leg(net, dev, dist, world, a.rehearse_on_one_gpu)     # collective: every rank takes part
    if dist is not None:
        tmax = mine[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = mine.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rates = [torch.zeros(1, dtype=torch.float64, device=mine.device) for _ in range(world)]
        dist.all_gather(rates, torch.tensor([c["sims"] / dt], dtype=torch.float64, device=mine.device))
        rank_rates = [float(r[0]) for r in rates]
        dt_max, tot = float(tmax[0]), tot.cpu().tolist()
        # the collectives end here, for every rank: what follows (side legs, CPU baseline) is rank 0's alone, and a
        # barrier inside it would have no partner
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    else:
        dt_max, tot = dt, mine.cpu().tolist()
    if rank != 0:
        return

    sims, rows, rows_raw, moves, games, depth_sum = tot[1:]
    fl_pos = tower_flops_per_position(wl["board"], 21, wl["ch"], wl["blocks"])
    fused = wl["board"] <= 9 and wl["ch"] <= 192
    per_launch_ms = tower_ms / max(1, launches)
    tower_rows = c["nn_rows"]      # this rank's rows: every network row of the window went through a timed tower launch
    achieved = (tower_rows * fl_pos) / (tower_ms * 1e-3) / 1e12 if tower_ms > 0 else 0.0
    traffic, traffic_src = tower_traffic_per_launch(tower_rows / max(1, launches))
    out = {
        "metric": "MCTS sims/sec (node) + self-play games/hour, 9x9 192ch×12blk @256 sims, 1/2/4/8 GPU",
        "value": round(sims / dt_max, 1), "unit": "sims/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt_max / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, games from the empty board)",
        "config": {"workload": wl["text"].format(g=a.games),
                   "games_per_gpu": a.games, "lanes": a.lanes, "inference_batch": a.batch, "dedup_repeated_leaves": not a.no_dedup, "root_evaluation_reuse": True,
                   "parallelism": f"games sharded over {world} GPU(s), one process per GPU, no data-path collective"},
        "sims_per_s": round(sims / dt_max, 1),
        "nn_rows_per_s": round(rows / dt_max, 1),
        "nn_rows_per_s_if_every_leaf_evaluated": round(rows_raw / dt_max, 1),
        "moves_per_s": round(moves / dt_max, 1),
        "games_finished_in_window": int(games),   # slots refill as games end; meaningful over windows of a game length or more (--steps 300)
        "mean_leaf_depth": round(depth_sum / max(1.0, sims), 3),
        "roofline": {"bound": "mfma", "kernel": ("k_tower<192,1,4,96,2,176> (two 256-thread workgroups per CU)" if headline else f"k_tower<{wl['ch']},...>") if fused else
                     "layer-by-layer network (k_conv3_slab + gate / residual / tail kernels), timed as one sequence", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                     "traffic": traffic if headline else None,
                     "traffic_note": (f"bytes per launch on the L2's memory side (Infinity Cache hits included), PMC passes in profiles/{traffic_src}; "
                                      "mostly the bf16 residual stream parked per workgroup between the two convolutions of a block") if headline else None,
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
        "per_rank_sims_per_s": {"min": round(min(rank_rates), 1), "max": round(max(rank_rates), 1),
                                "note": "each rank's own simulations / its own wall time over the timed steps; value uses the slowest rank's time"},
    }
    if train_leg is not None:
        out["train_step"] = train_leg
    if a.steps >= 300:      # a window of several game lengths: finished games counted, not derived
        out["games_per_hour_counted"] = round(games / dt_max * 3600, 1)
        out["games_per_hour_counted_note"] = f"{int(games)} games finished in the {dt_max:.1f} s window (slots refill as games end; the window starts from running games)"
    if pw.summary():
        out["power"] = pw.summary()
    if not headline and world == 1 and not a.no_side_legs:
        # small networks: the two HIP events around every tower launch cost a visible share of a move; the same moves of the
        # same games again without them, on a fresh driver (no roofline figures from this one)
        del sp
        sp = BatchedSelfPlay(net, cfg, games=a.games, seed0=1 + rank * a.games, dedup_leaves=not a.no_dedup, lanes=a.lanes,
                             tower_workgroups=twg)
        if a.sync_search:
            sp.set_async(False)
        sp.begin(num_games=10 ** 9)
        for _ in range(2 + a.warmup):          # the steps the first window ran before its clock started
            sp.step()
        for k in sp.counters:
            sp.counters[k] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            sp.step()
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t0
        out["without_launch_timing"] = {"ms_per_step": round(dt3 / a.steps * 1e3, 3), "sims_per_s": round(sp.counters["sims"] / dt3, 1),
                                        "graph_replays": int(sp.counters.get("graph_replays", 0))}
    del sp
    if not a.no_side_legs and headline:
        # games/hour: moves/s of the timed window over the plies a finished game takes, measured on whole games of the same
        # network and search settings in this run (a 4096-slot window of a few steps finishes no game of its own)
        plies, n_games = mean_plies_per_game(net, cfg)
        out["mean_plies_per_game"] = round(plies, 2)
        out["games_per_hour"] = round(moves / dt_max * 3600 / plies, 1)
        out["games_per_hour_note"] = (f"moves/s of the timed window x 3600 / {plies:.1f} plies per game, the mean over {n_games} whole games played "
                                      "in this run with the same network and search settings (256 lockstep slots, untimed)")
        if not a.no_dedup:   # the like-for-like figure for upstream's every-leaf evaluation: a short window of its own
            sp2 = BatchedSelfPlay(net, cfg, games=a.games, seed0=1 + rank * a.games, dedup_leaves=False, lanes=a.lanes, tower_workgroups=twg)
            dt2, _ = window(sp2, 5, 1, setup=1)
            out["value_no_dedup"] = round(sp2.counters["sims"] / dt2, 1)
            out["value_no_dedup_note"] = (f"sims/s of this rank over 5 steps with every leaf evaluated ({sp2.counters['nn_rows']} network rows for "
                                          f"{sp2.counters['sims']} simulations), as upstream does")
            del sp2
        out["roofline_other"], out["roofline_other_setup"] = kernel_census(net, a.batch, a.games)
    if world == 1 and not a.no_cpu_baseline and headline:
        try:
            out["cpu_baseline"] = cpu_baseline(a.batch, a.cpu_seconds)
        except Exception as e:  # the GPU number stands even if the CPU leg cannot run
            out["cpu_baseline"] = {"value": None, "unit": "sims/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
