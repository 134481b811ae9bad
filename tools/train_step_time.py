"""Time the training step (batch 256, 9x9, 192ch x 12) on one GPU, batches assembled by az_replay_gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphazero_amd import train as T

cfg = T.TrainConfig(device="cuda", batch_size=256, train_steps_per_iter=30)
cfg.use_amp = True
rng = np.random.RandomState(0)
samples = []
for _ in range(4096):
    s = (rng.rand(21, 9, 9) < 0.3).astype(np.float32); s[-2:] = 0.0
    p = rng.dirichlet([0.5] * 82).astype(np.float32)
    samples.append((s, p, float(rng.choice([-1.0, 1.0])), 0.5))
buf = T.ReplayBuffer(100000); buf.extend(samples)
model = T.build_model(cfg).cuda()
T.set_seed(0)
T.train_candidate(model, buf, cfg)       # warm-up (MIOpen find, allocator)
torch.cuda.synchronize(); t0 = time.time()
loss = T.train_candidate(model, buf, cfg)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"train step: {dt / cfg.train_steps_per_iter * 1e3:.1f} ms (batch 256, amp={cfg.use_amp}), {256 * cfg.train_steps_per_iter / dt:.0f} samples/s, loss {loss}")
