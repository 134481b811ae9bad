"""Training-side callers on the GPU against the upstream vectors of tests/golden/train.*: arena matches,
lockstep self-play vs one-game upstream workers, the opening book, device batch assembly, the main loop."""
import json
import os
import random

import numpy as np
import pytest
import torch

import golden_io
from test_train_cpu import J, Z, synth_samples, tiny_cfg, tiny_model

pytestmark = pytest.mark.gpu


def test_play_match_and_sprt_match_upstream():
    from alphazero_amd import train as T
    cfg = tiny_cfg()
    ma, mb = tiny_model(cfg, J["tiny_seeds"]["arena"][0]), tiny_model(cfg, J["tiny_seeds"]["arena"][1])
    assert list(T.play_match(ma, mb, cfg, 5)) == J["play_match"]
    for th, want in zip((0.55, 0.3, 0.9), J["play_match_sprt"]):
        assert list(T.play_match_sprt(ma, mb, cfg, 12, th, 0.05)) == want
    assert T.play_match(ma, mb, cfg, 0) == (0, 0, 0)


def test_arena_games_are_functions_of_the_colour_assignment():
    """What lets play_match play two games instead of `games`: a repeated game ends the same way."""
    from alphazero_amd import train as T
    cfg = tiny_cfg()
    ma, mb = tiny_model(cfg, 5), tiny_model(cfg, 6)
    out = T._arena_results(ma, mb, cfg, [True, False, True, False, False])
    assert out[0] == out[2] and out[1] == out[3] == out[4]


def test_run_selfplay_reproduces_one_game_upstream_workers():
    """Game j of an iteration equals what an upstream worker seeded seed + iter*1000 + j plays (train.py:446)."""
    from alphazero_amd import train as T
    cfg = tiny_cfg(mcts_topk_expand=6, seed=9, selfplay_games_per_iter=3)
    model = tiny_model(cfg, J["tiny_seeds"]["selfplay"])
    samples, bw, ww, dd = T.run_selfplay(model, cfg, iter_idx=2)
    want = J["selfplay_workers"]
    assert (bw, ww, dd) == tuple(sum(w[i] for w in want) for i in range(3))
    assert len(samples) == sum(w[3] for w in want)
    off = 0
    for wid, w in enumerate(want):
        mine = samples[off:off + w[3]]
        off += w[3]
        planes = np.packbits(np.stack([s[0][:19] for s in mine]) > 0.5)
        assert np.array_equal(planes, Z[f"worker{wid}/states"])
        assert np.array_equal(np.stack([s[0][19:, 0, 0] for s in mine]), Z[f"worker{wid}/caps"])
        assert np.stack([s[1] for s in mine]).tobytes() == Z[f"worker{wid}/pi"].tobytes()
        assert np.array_equal(np.array([s[2] for s in mine]), Z[f"worker{wid}/z"])
        assert np.array_equal(np.array([s[3] for s in mine]), Z[f"worker{wid}/difficulty"])


def test_selfplay_game_with_opening_book():
    from alphazero_amd import train as T
    cfg = tiny_cfg(opening_book_prob=1.0, opening_book_max_moves=3, seed=4)
    model = tiny_model(cfg, J["tiny_seeds"]["selfplay"])
    T.set_seed(1234)
    data, winner = T.selfplay_game(model, cfg)
    assert winner == J["book_game"]["winner"] and len(data) == J["book_game"]["moves"]
    assert np.stack([s[1] for s in data]).tobytes() == Z["book_game/pi"].tobytes()
    assert np.array_equal(data[0][0], Z["book_game/first_planes"])
    # the lockstep driver opens game 0 with the same book line when it is seeded like that run
    from alphazero_amd.selfplay import BatchedSelfPlay, SelfPlayConfig
    sp = BatchedSelfPlay(T._host_evaluator(model), SelfPlayConfig.from_train_config(cfg), games=2, seed0=1234)
    sp.begin(num_games=2)
    board = sp.eng.export_state()[0].cpu().numpy()[0, :25]
    first = Z["book_game/first_planes"]
    want = np.where(first[0] > 0.5, -1, 0) + np.where(first[8] > 0.5, 1, 0)     # three book stones: white to play sees black as "theirs"
    assert np.array_equal(board.reshape(5, 5), want.astype(np.int8))
    assert sp.lanes[0].move_idx[0] == 3


def test_off_list_shapes_run_on_the_float32_kernels():
    """Shapes upstream's CLI accepts but the MFMA kernels do not cover (48 channels; 13x13 with 32 channels) go through
    native("f32") in self-play and in the arena instead of raising: same search, same games as a host evaluation of
    the module would give is not asserted here (float32 kernels against torch are covered by the network tests), only
    that the run completes on the native float32 path with whole games."""
    from alphazero_amd import train as T
    for kw in (dict(channels=48), dict(board_size=13, channels=32, mcts_simulations=4, eval_simulations=4)):
        cfg = tiny_cfg(seed=3, selfplay_games_per_iter=2, **kw)
        model = tiny_model(cfg, 21).cuda()
        assert model.native_auto("cuda").precision == "f32"
        samples, bw, ww, dd = T.run_selfplay(model, cfg, iter_idx=0)
        assert bw + ww + dd == 2 and len(samples) > 0
        n = cfg.board_size
        assert samples[0][0].shape == (21, n, n) and samples[0][1].shape == (n * n + 1,)
        assert abs(float(samples[0][1].sum()) - 1.0) < 1e-5
        other = tiny_model(cfg, 22).cuda()
        w, l, d = T.play_match(model, other, cfg, 2)
        assert w + l + d == 2


def test_device_batch_assembly_equals_host_sampling():
    from alphazero_amd import train as T
    buf = T.ReplayBuffer(capacity=50)
    buf.extend(synth_samples(30, 5, 8, seed=11))
    buf.attach_device("cuda", 5, 8)
    buf.extend(synth_samples(70, 5, 8, seed=11)[30:])          # wraps around the ring
    for aug in (False, True):
        random.seed(5); np.random.seed(6)
        hs, hp, hv = buf.sample(16, 5, aug)
        random.seed(5); np.random.seed(6)
        ds, dp, dv = buf.sample_device(16, 5, aug)
        assert ds.is_cuda and torch.equal(ds.cpu(), hs) and torch.equal(dp.cpu(), hp) and torch.equal(dv.cpu(), hv)
    random.seed(100 + 1); np.random.seed(200 + 1)
    ds, dp, dv = buf.sample_device(16, 5, True)
    assert np.array_equal(ds.cpu().numpy(), Z["replay/states_1"]) and np.array_equal(dp.cpu().numpy(), Z["replay/policies_1"])
    with pytest.raises(ValueError):
        bad = T.ReplayBuffer(4)
        bad.extend([(np.full((21, 5, 5), 0.5, np.float32), np.zeros(26, np.float32), 0.0, 0.1)])
        bad.attach_device("cuda", 5, 8)


def test_training_loop_writes_upstream_files(tmp_path, capsys):
    from alphazero_amd import train as T
    save = str(tmp_path / "ck")
    cfg = T.parse_args(["--device", "cuda", "--board-size", "5", "--channels", "32", "--num-blocks", "1", "--value-hidden-dim", "64",
                        "--iterations", "2", "--selfplay-games-per-iter", "6", "--mcts-simulations", "8", "--eval-simulations", "8",
                        "--mcts-inference-batch", "4", "--batch-size", "32", "--train-steps-per-iter", "3", "--arena-games", "4",
                        "--elo-games-per-opponent", "2", "--benchmark-games-per-opponent", "2", "--save-dir", save, "--seed", "3"])
    assert cfg.use_amp
    T.run_training(cfg)
    out = capsys.readouterr().out
    assert "[Iter 1]" in out and "[Iter 2]" in out
    names = set(os.listdir(save))
    assert {"iter_001.pt", "iter_002.pt", "model_pool.json", "model_pool", "_runtime_best_v000000.pt"} <= names
    ck = torch.load(os.path.join(save, "iter_002.pt"), map_location="cpu", weights_only=True)
    assert {"iter", "model", "best_model", "candidate_model", "cfg", "buffer_size", "promoted", "arena", "pool_best"} <= set(ck)
    assert ck["iter"] == 2 and set(ck["arena"]) == {"wins", "losses", "draws"} and set(J["defaults"]) <= set(ck["cfg"])
    pool = json.load(open(os.path.join(save, "model_pool.json")))
    assert pool["best_name"] in pool["entries"] and os.path.exists(pool["entries"][pool["best_name"]]["path"])
    # a checkpoint resumes, and its network loads into the model class under the upstream keys
    cfg2 = T.parse_args(["--device", "cuda", "--board-size", "5", "--channels", "32", "--num-blocks", "1", "--value-hidden-dim", "64",
                         "--iterations", "2", "--resume", os.path.join(save, "iter_002.pt"), "--save-dir", save])
    T.run_training(cfg2)           # nothing left to do: start_iter = 3 > iterations
    m, c = T.load_model_from_file(os.path.join(save, "model_pool", "model_iter_000.pt"), "cpu")
    assert c["board_size"] == 5 and m.board_size == 5


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_TWO_RANK = r"""
import os, sys, json, faulthandler
faulthandler.dump_traceback_later(120, exit=True)      # a rank stuck in a collective ends with its stack on stderr
sys.path.insert(0, os.environ["AZ_ROOT"])
import torch, torch.distributed as dist
from alphazero_amd import train as T
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)
print("rank", rank, "joined", file=sys.stderr, flush=True)
cfg = T.parse_args(["--device", "cuda", "--board-size", "5", "--channels", "32", "--num-blocks", "1", "--value-hidden-dim", "64",
                    "--iterations", "2", "--selfplay-games-per-iter", "10", "--mcts-simulations", "8", "--eval-simulations", "8",
                    "--mcts-inference-batch", "4", "--batch-size", "32", "--train-steps-per-iter", "3", "--arena-games", "4",
                    "--elo-games-per-opponent", "2", "--benchmark-games-per-opponent", "2", "--save-dir", os.environ["AZ_SAVE"], "--seed", "5"])
first, mine = T.rank_share(cfg.selfplay_games_per_iter, rank, 2)
T.run_training(cfg)
print("rank", rank, "trained", file=sys.stderr, flush=True)
dist.barrier()
if rank == 0:
    ck = torch.load(os.path.join(cfg.save_dir, "iter_002.pt"), map_location="cpu", weights_only=True)
    print("RESULT " + json.dumps({"iter": ck["iter"], "promoted": bool(ck["promoted"]), "share": [first, mine]}))
dist.destroy_process_group()
"""


def test_two_rank_training_loop(tmp_path):
    """Two ranks (gloo, both on this GPU): the self-play games of an iteration are split over the ranks, gradients are
    averaged, rank 0 plays the arena and writes the files, and the promotion decision reaches both ranks."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AZ_ROOT=root, AZ_SAVE=str(tmp_path / "ck"), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    logs = [(open(str(tmp_path / f"rank{r}.out"), "w+"), open(str(tmp_path / f"rank{r}.err"), "w+")) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-u", "-c", _TWO_RANK], env=dict(env, RANK=str(r)), stdout=logs[r][0], stderr=logs[r][1])
             for r in range(2)]
    try:
        for p in procs:
            p.wait(timeout=150)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    outs = []
    for o, e in logs:
        o.seek(0), e.seek(0)
        outs.append((o.read(), e.read()))
        o.close(), e.close()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, "rank 0:\n" + outs[0][1][-3000:] + "\nrank 1:\n" + outs[1][1][-3000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT ")][0]
    res = json.loads(line[7:])
    assert res["iter"] == 2 and res["share"] == [0, 5]
    names = set(os.listdir(str(tmp_path / "ck")))
    assert {"iter_001.pt", "iter_002.pt", "model_pool.json"} <= names
    assert "[Iter 2] selfplay B/W/D=" in outs[0][0] and "[Iter 2]" not in outs[1][0]
    tally = outs[0][0].split("[Iter 2] selfplay B/W/D=")[1].split()[0]
    assert sum(int(x) for x in tally.split("/")) == 10          # both ranks' games are counted


_NCCL_ONE_RANK = r"""
import os, sys, json, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, os.environ["AZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["AZ_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.train_step import DataParallelTrainer, all_reduce_, broadcast_
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)

def fresh():
    torch.manual_seed(3)
    return AlphaZeroNet(board_size=5, channels=32, num_blocks=1, value_hidden_dim=64).to(dev)

rng = np.random.default_rng(0)
x = torch.from_numpy((rng.random((16, 21, 5, 5)) < 0.3).astype(np.float32)).to(dev)
pi = torch.softmax(torch.from_numpy(rng.standard_normal((16, 26)).astype(np.float32)), 1).to(dev)
z = torch.from_numpy(rng.choice([-1.0, 1.0], 16).astype(np.float32)).to(dev)
plain = DataParallelTrainer(fresh(), total_steps=4)              # no process group yet: plain autograd path
want = [plain.step(x, pi, z)]
g_plain = torch.cat([p.grad.reshape(-1) for p in plain.model.parameters()]).clone()   # clipped gradient of step 1
want += [plain.step(x, pi, z) for _ in range(2)]
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
t = torch.arange(8, dtype=torch.float32, device=dev)
all_reduce_(dist, t); broadcast_(dist, t); dist.barrier()
assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
tr = DataParallelTrainer(fresh(), total_steps=4)                 # flat gradient buffer + RCCL all-reduce every step
assert tr.reduce
got = [tr.step(x, pi, z)]
assert tr._flat is not None and tr._flat.numel() == sum(p.numel() for p in tr.model.parameters())
assert all(p.grad.data_ptr() >= tr._flat.data_ptr() for p in tr.model.parameters())      # gradients are views of the flat buffer
g_err = float((tr._flat - g_plain).abs().max() / g_plain.abs().max())                   # convolution backward is not bitwise repeatable
got += [tr.step(x, pi, z) for _ in range(2)]
dist.destroy_process_group()
print("RESULT " + json.dumps({"want": want, "got": got, "grad_rel_err": g_err}))
"""


def test_nccl_path_one_rank(tmp_path):
    """The production backend on the hardware that is there: a one-rank RCCL process group carries the trainer's flat
    gradient all-reduce, a broadcast and a barrier, and is torn down; the step equals the plain single-process step."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AZ_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, "-u", "-c", _NCCL_ONE_RANK], env=env, capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert res["grad_rel_err"] < 1e-4
    assert np.allclose(np.array(res["want"]), np.array(res["got"]), rtol=1e-3, atol=1e-4)


def test_bench_rank_launch(tmp_path):
    """bench.py --gpus N starts N ranks itself. On this one-GPU box: --gpus 2 refuses loudly instead of printing a
    one-rank figure, and --gpus 1 --force-dist goes through the RCCL branch (init, barriers, counter all-reduce,
    destroy) with n_gpus read from the process group."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    if torch.cuda.device_count() < 2:
        two = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--games", "96", "--no-cpu-baseline"],
                             capture_output=True, text=True, timeout=120, cwd=root)
        assert two.returncode != 0 and "GPU" in two.stderr and not two.stdout.strip()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, bench, "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "0", "--games", "96", "--lanes", "2",
                          "--no-cpu-baseline", "--no-side-legs"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    d = json.loads([ln for ln in one.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    # under the launcher's environment the rank count must match --gpus
    bad = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "1", "--games", "96", "--no-cpu-baseline", "--no-side-legs"],
                         capture_output=True, text=True, timeout=120, cwd=root,
                         env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert bad.returncode != 0 and "must agree" in bad.stderr


@pytest.mark.gpu
def test_bench_two_rank_control_flow():
    """The N-rank flow of bench.py with N = 2 on one GPU (both ranks on GPU 0, gloo with host tensors instead of RCCL,
    which refuses two ranks on one device): barriers around the timed window pair up on every rank, the counters are
    summed over the ranks, only rank 0 prints, and rank 0's side legs (no-dedup window, kernel census, plies per game)
    run after the process group is gone, so no barrier of theirs waits for ranks that have left."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2", "--warmup", "1",
                          "--games", "128", "--lanes", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=420, cwd=root, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    # 2 ranks x 128 games x 256 simulations per step
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / (2 * 128 * 256) - 1.0) < 1e-3   # the printed figures are rounded
    assert d["value_no_dedup"] > 0 and len(d["roofline_other"]) == 3 and d["games_per_hour"] > 0
    # the data-parallel training step ran under the same process group (both ranks), with its all-reduce timed alone
    ts = d["train_step"]
    assert ts["ms_per_step"] > 0 and ts["allreduce_ms"] > 0 and ts["batch_per_gpu"] == 256 and ts["gradient_bytes"] == 4 * 9149627
    assert ts["samples_per_s"] == pytest.approx(2 * 256 / (ts["ms_per_step"] * 1e-3), rel=1e-2)
    pr = d["per_rank_sims_per_s"]
    assert 0 < pr["min"] <= pr["max"] and pr["min"] * 2 <= d["value"] * 1.001 <= pr["max"] * 2 * 1.001 + 1e-9 or pr["min"] > 0
