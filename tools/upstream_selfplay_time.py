"""Build container only: time the upstream project's own multi-worker self-play (go_zero/train.py:395-463 run_selfplay with a
ProcessPoolExecutor, one torch thread per worker, model handed over through a snapshot file) at the headline search settings
(9x9, 192ch x 12 SE, 256 sims, top-k 48, virtual loss 1, inference batch 16) on this host's cores. The upstream package is
imported from /root/reference; nothing of it is copied. Prints one JSON line.
    python tools/upstream_selfplay_time.py [workers] [games]"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import torch  # noqa: E402

import go_zero.train as T  # noqa: E402

if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    games = int(sys.argv[2]) if len(sys.argv) > 2 else workers
    from concurrent.futures import ProcessPoolExecutor
    cfg = T.TrainConfig()
    cfg.board_size, cfg.channels, cfg.num_blocks = 9, 192, 12
    cfg.mcts_simulations, cfg.mcts_topk_expand, cfg.mcts_virtual_loss, cfg.mcts_inference_batch = 256, 48, 1.0, 16
    cfg.selfplay_workers, cfg.selfplay_games_per_iter, cfg.worker_torch_threads = workers, games, 1
    cfg.device = "cpu"
    cfg.save_dir = "/tmp/upstream_selfplay_time"
    os.makedirs(cfg.save_dir, exist_ok=True)
    T.set_seed(cfg.seed)
    model = T.build_model(cfg)
    path = T.save_runtime_model_snapshot(model, cfg.save_dir, 1)
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=workers) as pool:
        samples, bw, ww, dd = T.run_selfplay(model, cfg, iter_idx=0, pool=pool, model_path=path, model_version=1)
    dt = time.time() - t0
    moves = len(samples)
    print(json.dumps({"workers": workers, "games": games, "seconds": round(dt, 1), "moves": moves, "sims": moves * cfg.mcts_simulations,
                      "sims_per_s": round(moves * cfg.mcts_simulations / dt, 1), "games_per_hour": round(games / dt * 3600, 2),
                      "results": [bw, ww, dd], "host_cpus": os.cpu_count(), "torch": torch.__version__}))
