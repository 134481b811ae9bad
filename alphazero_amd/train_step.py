"""The training step of the upstream train_candidate (go_zero/train.py:504-554) as a reusable object,
with data-parallel gradient averaging over torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" on CPU for tests). One process per GPU; the all-reduce is bucketed into one flat
buffer (9.15 M parameters = 36.6 MB float32 for the 192x12 net: a single collective per step).

Loss = soft-target cross entropy + Huber(value, z) - entropy_coeff * entropy, AdamW, cosine LR to
0.1*lr, gradient clipping at max-norm 1.0 -- same formulas and order as upstream.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def loss_terms(logits, values, target_policies, target_values, huber_delta=1.0, entropy_coeff=1e-3):
    """train.py:537-541 -> (loss, policy_loss, value_loss)."""
    logp = F.log_softmax(logits, dim=1)
    policy_loss = -(target_policies * logp).sum(dim=1).mean()
    value_loss = F.huber_loss(values, target_values, delta=huber_delta)
    entropy = -(F.softmax(logits, dim=1) * logp).sum(dim=1).mean()
    return policy_loss + value_loss - entropy_coeff * entropy, policy_loss, value_loss


def _staged(dist, t, group):
    """gloo moves device tensors through its own pinned buffers and streams, which is seen to stall on ROCm;
    with that backend a device tensor goes through a host copy made here. RCCL ("nccl") takes it as it is."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def broadcast_(dist, t, src=0, group=None):
    """dist.broadcast(t) in place, for any backend / device combination this package runs with."""
    if _staged(dist, t, group):
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def all_reduce_(dist, t, op=None, group=None):
    """dist.all_reduce(t) in place (default: sum), host-staged under gloo like broadcast_."""
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(dist, t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


class DataParallelTrainer:
    def __init__(self, model, lr=1e-3, weight_decay=1e-4, total_steps=800, grad_clip_norm=1.0, huber_delta=1.0,
                 entropy_coeff=1e-3, target_smoothing=0.0, use_amp=False, process_group=None):
        self.model = model
        self.opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)            # train.py:505
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=max(1, total_steps), eta_min=lr * 0.1)
        self.clip, self.delta, self.ent = grad_clip_norm, huber_delta, entropy_coeff
        self.smooth = float(np.clip(target_smoothing, 0.0, 0.25))
        dev = next(model.parameters()).device
        self.amp = bool(use_amp and dev.type == "cuda")
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.amp)                                  # train.py:512-513
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = process_group
        self.world = self.dist.get_world_size(process_group) if self.dist else 1
        # under an initialised process group the gradients always go through the flat buffer and the collective, also with
        # one rank (an identity there): a one-GPU run then executes the code path the N-GPU run takes
        self.reduce = self.dist is not None
        self._flat = None
        if self.dist:  # identical starting weights on every rank
            for p in model.state_dict().values():
                broadcast_(self.dist, p.data if p.is_floating_point() else p, src=0, group=process_group)

    def _bind_flat_grads(self):
        """All gradients live in one flat buffer (each p.grad is a view of its slice): the all-reduce takes the buffer
        as it is, with no gather / scatter copies around it."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        self._flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            n = p.numel()
            p.grad = self._flat[off:off + n].view_as(p)
            off += n

    def _allreduce_grads(self):
        all_reduce_(self.dist, self._flat, group=self.group)
        self._flat.div_(self.world)

    def step(self, states, target_policies, target_values):
        """One optimisation step on this rank's shard of the global batch (train.py:526-548)."""
        self.model.train()
        if self.smooth > 0.0:
            target_policies = target_policies * (1.0 - self.smooth) + (self.smooth / target_policies.shape[1])
        if self.reduce:
            if self._flat is None:
                self._bind_flat_grads()
            else:
                self._flat.zero_()
        else:
            self.opt.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", enabled=self.amp):
            logits, values = self.model(states)
            loss, pl, vl = loss_terms(logits, values, target_policies, target_values, self.delta, self.ent)
        self.scaler.scale(loss).backward()
        if self.reduce:   # before unscale_: every rank then sees the same overflow verdict and the same scale
            self._allreduce_grads()
        self.scaler.unscale_(self.opt)
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.scaler.step(self.opt)
        self.scaler.update()
        self.sched.step()
        return float(loss.item()), float(pl.item()), float(vl.item())
