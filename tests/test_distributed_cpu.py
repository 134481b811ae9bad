"""World-size-2 checks on the gloo backend: (1) games shard over ranks with no data-path collective
and per-game seeds make the union of the shards equal the single-rank run (host logic only, stub
device), (2) the data-parallel training step averages gradients so that two ranks with half the
batch each take the same step as one rank with the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.train_step import DataParallelTrainer
    torch.manual_seed(100 + rank)                      # different init per rank: the trainer must broadcast rank 0's
    net = AlphaZeroNet(board_size=5, channels=32, num_blocks=1, value_hidden_dim=32)
    tr = DataParallelTrainer(net, lr=1e-2, total_steps=10)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand((8, 21, 5, 5), generator=g) < 0.3).float()
    pi = torch.softmax(torch.randn((8, 26), generator=g), 1)
    z = torch.rand((8,), generator=g) * 2 - 1
    sl = slice(rank * 4, rank * 4 + 4)
    losses = [tr.step(x[sl], pi[sl], z[sl]) for _ in range(3)]
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    if rank == 0:
        torch.save({"flat": flat, "losses": losses}, out)
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged"
    dist.destroy_process_group()


def test_data_parallel_step_matches_single_process(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_train_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    sys.path.insert(0, ROOT)
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.train_step import DataParallelTrainer
    torch.set_num_threads(1)
    torch.manual_seed(100)
    net = AlphaZeroNet(board_size=5, channels=32, num_blocks=1, value_hidden_dim=32)
    net.eval()   # batch-norm batch statistics differ between a batch of 8 and two of 4: compare in eval-mode BN
    tr = DataParallelTrainer(net, lr=1e-2, total_steps=10)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand((8, 21, 5, 5), generator=g) < 0.3).float()
    pi = torch.softmax(torch.randn((8, 26), generator=g), 1)
    z = torch.rand((8,), generator=g) * 2 - 1
    # reference: one process, whole batch, gradients of the two half-batch means averaged by hand
    from alphazero_amd.train_step import loss_terms
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=1e-3)
    for _ in range(3):
        net.train()
        grads = None
        for sl in (slice(0, 4), slice(4, 8)):
            opt.zero_grad(set_to_none=True)
            lg, v = net(x[sl])
            loss, _, _ = loss_terms(lg, v, pi[sl], z[sl])
            loss.backward()
            gs = [p.grad.clone() for p in net.parameters()]
            grads = gs if grads is None else [a + b for a, b in zip(grads, gs)]
        # BN running stats were updated twice here and once per rank there; parameters are what we compare
        for p, gsum in zip(net.parameters(), grads):
            p.grad = gsum / 2
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        sched.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(flat, got["flat"], atol=1e-6), float((flat - got["flat"]).abs().max())


def _shard_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero_amd.selfplay import HostRng
    G = 6                                   # games per rank; game id = rank * G + slot, seed = seed0 + id
    rng = HostRng(G, threads=1)
    for s in range(G):
        rng.seed(s, 1000 + rank * G + s)
    noise = rng.dirichlet(np.full(G, 26, np.int32), 0.3, 26)
    u = rng.random_sample()
    t = torch.from_numpy(np.concatenate([noise.reshape(-1), u]))
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)            # test-only collective: the data path itself has none
    if rank == 0:
        torch.save(torch.stack(gathered), out)
    dist.barrier()
    dist.destroy_process_group()


def test_game_sharding_is_seeded_per_game(tmp_path):
    out = str(tmp_path / "shard.pt")
    mp.spawn(_shard_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True).numpy()
    for rank in range(2):
        for s in range(6):
            rs = np.random.RandomState(1000 + rank * 6 + s)
            want = rs.dirichlet([0.3] * 26)
            assert np.array_equal(got[rank][s * 26:(s + 1) * 26], want)
            assert got[rank][6 * 26 + s] == rs.random_sample()


def _loop_worker(rank, world, port, save):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from alphazero_amd import train as T

    def fake_selfplay(best_model, cfg, iter_idx, pool=None, model_path=None, model_version=0):
        # stands in for the GPU self-play of this rank's share: one sample per game, labelled by the game id
        first, mine = T.rank_share(cfg.selfplay_games_per_iter, rank, world)
        rs = np.random.RandomState(cfg.seed + iter_idx * 1000 + first)
        out = []
        for g in range(mine):
            planes = np.zeros((2 * cfg.history_len + 5, cfg.board_size, cfg.board_size), np.float32)
            planes[:-2] = rs.random_sample(planes[:-2].shape) < 0.3
            pi = rs.dirichlet([0.5] * (cfg.board_size ** 2 + 1)).astype(np.float32)
            out.append((planes, pi, float(rs.choice([-1.0, 1.0])), 1.0))
        return out, mine, 0, 0

    def fake_arena(model_a, model_b, cfg, colours):
        assert rank == 0, "only rank 0 plays arena games"
        return np.array([1, 1][: len(colours)])          # the candidate wins both distinct games

    T.run_selfplay = fake_selfplay
    T._arena_results = fake_arena
    trained = []
    real_train = T.train_candidate

    def spy_train(candidate, buffer, cfg):
        trained.append(candidate)
        return real_train(candidate, buffer, cfg)

    T.train_candidate = spy_train
    cfg = T.parse_args(["--device", "cpu", "--board-size", "5", "--channels", "32", "--num-blocks", "1", "--value-hidden-dim", "32",
                        "--iterations", "3", "--selfplay-games-per-iter", "9", "--batch-size", "8", "--train-steps-per-iter", "2",
                        "--arena-games", "4", "--elo-games-per-opponent", "2", "--benchmark-games-per-opponent", "2",
                        "--save-dir", save, "--seed", "5"])
    T.run_training(cfg)
    # each rank trained on its own batches: the BatchNorm running statistics must have been brought together, otherwise
    # the ranks go on with different eval-mode networks from the first promotion on
    flat = torch.cat([b.detach().reshape(-1).float() for b in trained[-1].buffers()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "BatchNorm buffers differ between the ranks"
    assert float(flat.abs().sum()) > 0
    dist.barrier()
    dist.destroy_process_group()


def test_selfplay_seed_ranges_do_not_overlap():
    """Game j of iteration i is seeded seed + i*stride + j: upstream's stride of 1000 while an iteration has at most 1000
    games, the game count beyond that, so that consecutive iterations never share a seed."""
    sys.path.insert(0, ROOT)
    from alphazero_amd import train as T
    for games in (3, 1000, 1001, 4096):
        cfg = T.parse_args(["--selfplay-games-per-iter", str(games), "--seed", "9"])
        a = set(range(T.selfplay_seed0(cfg, 4), T.selfplay_seed0(cfg, 4) + games))
        b = set(range(T.selfplay_seed0(cfg, 5), T.selfplay_seed0(cfg, 5) + games))
        assert not (a & b)
        if games <= 1000:
            assert T.selfplay_seed0(cfg, 4) == 9 + 4 * 1000          # train.py:446


def test_training_loop_collectives_line_up(tmp_path):
    """run_training on two ranks with self-play and arena games stubbed out: rank 0 alone plays matches and
    writes files, both ranks skip iteration 1 (4 and 5 samples < batch 8) and train from iteration 2 on, the
    promotion decision reaches rank 1, and no rank is left waiting in a collective the other never enters."""
    save = str(tmp_path / "ck")
    mp.spawn(_loop_worker, args=(2, _free_port(), save), nprocs=2, join=True)
    names = set(os.listdir(save))
    assert {"iter_002.pt", "iter_003.pt", "best.pt", "model_pool.json"} <= names and "iter_001.pt" not in names
    ck = torch.load(os.path.join(save, "iter_003.pt"), map_location="cpu", weights_only=True)
    assert ck["iter"] == 3 and ck["promoted"] is True and ck["arena"] == {"wins": 4, "losses": 0, "draws": 0}
