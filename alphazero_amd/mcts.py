"""MCTS with the upstream constructor and run() signature (go_zero/mcts.py:30-57, :227-284), backed
by the device engine: the tree, the rules and the masked softmax all run in HIP kernels; this class
only sequences the calls, feeds the evaluator and applies the temperature exactly as upstream does.
"""
from __future__ import annotations

import threading

import numpy as np
import torch

from . import _native as N
from .engine import Engine
from .go import GoState
from .model import AlphaZeroNet


def visits_to_pi(visits: np.ndarray, temperature: float) -> np.ndarray:
    """mcts.py:270-279 on a float64 visit vector."""
    if temperature <= 1e-2:
        pi = np.zeros_like(visits)
        pi[np.argmax(visits)] = 1.0
        return pi
    v = np.power(visits, 1.0 / temperature)
    s = v.sum()
    return np.ones_like(v) / len(v) if s <= 0 else v / s


class MCTS:
    def __init__(self, model, board_size: int = 9, num_simulations: int = 128, c_base: float = 19652.0,
                 c_init: float = 1.25, dirichlet_alpha: float = 0.3, dirichlet_eps: float = 0.25,
                 inference_batch_size: int = 1, topk_expand: int = 0, virtual_loss: float = 1.0,
                 root_parallelism: int = 1, device: str = "cpu"):
        self.model = model
        self.board_size = board_size
        self.num_simulations = num_simulations
        self.c_base, self.c_init = c_base, c_init
        self.dirichlet_alpha, self.dirichlet_eps = dirichlet_alpha, dirichlet_eps
        self.inference_batch_size = max(1, inference_batch_size)     # mcts.py:53-56 clamps
        self.topk_expand = max(0, topk_expand)
        self.virtual_loss = float(max(0.0, virtual_loss))
        self.root_parallelism = max(1, int(root_parallelism))
        self.device = device          # where `model` is evaluated; the search itself always runs on the GPU
        self._engines: dict = {}
        self._lock = threading.RLock()   # one search at a time per object: its engines hold the tree being searched

    # ------------------------------------------------------------------ plumbing
    def _engine(self, state: GoState, sims: int) -> Engine:
        key = (state.history_len, float(state.komi), sims)
        eng = self._engines.get(key)
        if eng is None:
            eng = self._engines[key] = Engine(
                board_size=self.board_size, history_len=state.history_len, games=1, komi=state.komi,
                num_simulations=sims, c_base=self.c_base, c_init=self.c_init, dirichlet_alpha=self.dirichlet_alpha,
                dirichlet_eps=self.dirichlet_eps, inference_batch_size=self.inference_batch_size,
                topk_expand=self.topk_expand, virtual_loss=self.virtual_loss)
        return eng

    def _evaluator(self):
        """(layout, fn, net): fn maps a device feature tensor to (logits, values) float32 device tensors; net is the
        native network handle behind fn, if any (its fused kind lets the engine enqueue a search without round trips)."""
        model, dev = self.model, str(self.device)
        if isinstance(model, AlphaZeroNet) and not model.training and next(model.parameters()).is_cuda:
            try:
                net = model.native("bf16")
                return N.AZ_FEAT_BF16_NHWC, net.forward, net
            except N.NativeError:
                net = model.native("f32")
                return N.AZ_FEAT_F32_NCHW, net.forward, None

        def generic(feats):   # any callable x[b,C,n,n] -> (logits, value), evaluated where upstream would (mcts.py:59-66)
            with torch.inference_mode():
                x = feats.to(dev)
                with torch.autocast(device_type="cuda", enabled=dev.startswith("cuda")):
                    logits, value = model(x)
            return (torch.as_tensor(logits).detach().to(feats.device, torch.float32),
                    torch.as_tensor(value).detach().to(feats.device, torch.float32))

        return N.AZ_FEAT_F32_NCHW, generic, None

    def _run_single_root_search(self, root_state: GoState, add_exploration_noise: bool, simulations: int):
        """mcts.py:165-225 -> (visits float64 [A], root value)."""
        with self._lock:
            eng = self._engine(root_state, simulations)
            eng.import_state(*root_state._packed(eng.seen_cap))
            layout, fn, net = self._evaluator()
            visits, rv = eng.run_search(fn, add_noise=add_exploration_noise, rngs=np.random, layout=layout, native_net=net)
            return visits.cpu().numpy()[0].astype(np.float64), float(rv.cpu()[0])

    # ------------------------------------------------------------------ upstream entry point
    def run(self, root_state: GoState, temperature: float = 1.0, add_exploration_noise: bool = True,
            return_root_value: bool = False):
        a = self.board_size * self.board_size + 1
        if root_state.is_terminal():                                   # mcts.py:234-239
            pi = np.zeros(a, dtype=np.float32)
            pi[-1] = 1.0
            if return_root_value:
                return pi, int(pi.argmax()), float(root_state.outcome_for_current_player())
            return pi, int(pi.argmax())
        roots = min(self.root_parallelism, self.num_simulations)
        if roots <= 1:
            visits, root_value = self._run_single_root_search(root_state, add_exploration_noise, self.num_simulations)
        else:                                                          # mcts.py:247-268, run one root after another
            parts = [self.num_simulations // roots] * roots
            for i in range(self.num_simulations % roots):
                parts[i] += 1
            visits, acc = np.zeros(a, np.float64), 0.0
            for sims_i in parts:
                v_i, r_i = self._run_single_root_search(root_state, add_exploration_noise, sims_i)
                visits += v_i
                acc += r_i * sims_i
            root_value = acc / float(max(1, self.num_simulations))
        pi = visits_to_pi(visits, temperature)
        action = int(np.random.choice(len(pi), p=pi))                  # mcts.py:281
        if return_root_value:
            return pi.astype(np.float32), action, root_value
        return pi.astype(np.float32), action
