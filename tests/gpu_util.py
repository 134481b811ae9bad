"""Helpers shared by the GPU parity tests."""
from __future__ import annotations

import numpy as np
import torch

import stubnets


def stub_evaluator(kind: str, counter=None):
    """Device-tensor wrapper around a stub net: features (f32 NCHW, device) -> logits/values on device."""

    def ev(feats):
        x = feats.detach().cpu().numpy()
        if counter is not None:
            counter[0] += x.shape[0]
        lg, v = stubnets.stub_eval(kind, x)
        return torch.from_numpy(lg).to(feats.device), torch.from_numpy(v).to(feats.device)

    return ev


def device_snapshot(eng):
    """Per-game (board, meta[8]) + legal mask + features + score from the device, as numpy."""
    score, info = eng.status()
    board, hist, sc, seen = eng.export_state()
    return dict(board=board.cpu().numpy(), info=info.cpu().numpy(), score=score.cpu().numpy(),
                legal=eng.legal_mask().cpu().numpy(), feats=eng.features().cpu().numpy(),
                sc=sc.cpu().numpy(), hist=hist.cpu().numpy(), seen=seen.cpu().numpy())
