"""Python handle on the device engine: G lockstep games + their search trees in HBM.

torch is used for device buffers and streams only; every computation is a HIP kernel in
libazamd.so reached through the C ABI of include/azamd.h.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

import os

from . import _native as N

# Replay a captured HIP graph for the batch rounds of a search where the network allows it (launch timing off, scratch
# reserved) and the stream can be captured (not the legacy default stream), instead of enqueueing the rounds' kernels one by
# one. Same results. With a move's host half on the device the rounds of a small network are bound by their launches as much
# as by their execution: 512 games of 64ch x 4 in two lanes 1.96 -> 1.85 ms per move; nothing either way on the 192ch x 12
# network. AZ_GRAPH_ROUNDS=0 switches it off.
GRAPH_ROUNDS = os.environ.get("AZ_GRAPH_ROUNDS", "1") != "0"


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """Device-resident Go games and PUCT trees (see include/azamd.h for the call protocol)."""

    def __init__(self, board_size=9, history_len=8, games=1, komi=5.5, num_simulations=128,
                 c_base=19652.0, c_init=1.25, dirichlet_alpha=0.3, dirichlet_eps=0.25,
                 inference_batch_size=1, topk_expand=0, virtual_loss=1.0, dedup_leaves=True, device=None):
        if not torch.cuda.is_available():
            raise N.NativeError("alphazero_amd.Engine needs a GPU (no CPU fallback)")
        self.L = N.lib()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.n, self.H, self.G = int(board_size), int(history_len), int(games)
        self.A, self.Cp = self.n * self.n + 1, 2 * self.H + 5
        self.cpad = 32 if self.Cp <= 32 else 64
        self.sims = int(num_simulations)
        self.B = max(1, int(inference_batch_size))
        self.alpha, self.eps = float(dirichlet_alpha), float(dirichlet_eps)
        cfg = N.AzConfig(self.n, self.H, self.G, self.sims, self.B, max(0, int(topk_expand)),
                         1 if dedup_leaves else 0, dev.index, float(komi), float(c_base), float(c_init),
                         float(dirichlet_eps), float(max(0.0, virtual_loss)))
        h = C.c_void_p()
        with torch.cuda.device(dev):
            N.check(self.L.az_engine_create(C.byref(cfg), C.byref(h)), "az_engine_create")
        self._h = h
        self.seen_cap = self.L.az_engine_seen_capacity(h)
        self.max_rows = self.L.az_engine_max_rows(h)

    def close(self):
        if getattr(self, "_h", None):
            self.L.az_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ rules
    def _t(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def reset(self, mask=None):
        m = None if mask is None else torch.as_tensor(mask, dtype=torch.uint8, device=self.device).contiguous()
        N.check(self.L.az_rules_reset(self._h, _ptr(m), _stream()), "az_rules_reset")

    def play(self, actions, raise_on_illegal=True):
        a = torch.as_tensor(actions, dtype=torch.int32, device=self.device).contiguous()
        st = self._t((self.G,), torch.int32)
        N.check(self.L.az_rules_play(self._h, _ptr(a), _ptr(st), _stream()), "az_rules_play")
        if raise_on_illegal:
            bad = torch.nonzero(st != 0)
            if bad.numel():
                g = int(bad[0, 0])
                act = int(a[g])
                raise ValueError(f"Illegal move: {divmod(act, self.n)}")
        return st

    def legal_mask(self):
        m = self._t((self.G, self.A), torch.uint8)
        N.check(self.L.az_rules_legal_mask(self._h, _ptr(m), _stream()), "az_rules_legal_mask")
        return m

    def features(self, layout=N.AZ_FEAT_F32_NCHW):
        if layout == N.AZ_FEAT_F32_NCHW:
            out = self._t((self.G, self.Cp, self.n, self.n), torch.float32)
        else:
            out = self._t((self.G, self.n * self.n, self.cpad), torch.bfloat16)
        N.check(self.L.az_rules_features(self._h, _ptr(out), layout, _stream()), "az_rules_features")
        return out

    def status(self):
        """(score f64 [G,2], info i32 [G,8]): see az_rules_status."""
        sc = self._t((self.G, 2), torch.float64)
        info = self._t((self.G, 8), torch.int32)
        N.check(self.L.az_rules_status(self._h, _ptr(sc), _ptr(info), _stream()), "az_rules_status")
        return sc, info

    def export_state(self):
        board = self._t((self.G, self.n * self.n), torch.int8)
        hist = self._t((self.G, self.H, self.n * self.n), torch.int8)
        sc = self._t((self.G, 8), torch.int32)
        seen = self._t((self.G, self.seen_cap), torch.int64)
        N.check(self.L.az_rules_export(self._h, _ptr(board), _ptr(hist), _ptr(sc), _ptr(seen), _stream()), "az_rules_export")
        return board, hist, sc, seen

    def import_state(self, board, hist, sc, seen):
        dev = self.device
        board = torch.as_tensor(board, dtype=torch.int8, device=dev).contiguous()
        hist = torch.as_tensor(hist, dtype=torch.int8, device=dev).contiguous()
        sc = torch.as_tensor(sc, dtype=torch.int32, device=dev).contiguous()
        seen = torch.as_tensor(seen, dtype=torch.int64, device=dev).contiguous()
        assert board.shape == (self.G, self.n * self.n) and hist.shape == (self.G, self.H, self.n * self.n)
        assert sc.shape == (self.G, 8) and seen.shape == (self.G, self.seen_cap)
        N.check(self.L.az_rules_import(self._h, _ptr(board), _ptr(hist), _ptr(sc), _ptr(seen), _stream()), "az_rules_import")

    # ------------------------------------------------------------------ search protocol
    def search_begin(self, enable=None):
        m = None if enable is None else torch.as_tensor(enable, dtype=torch.uint8, device=self.device).contiguous()
        N.check(self.L.az_mcts_begin(self._h, _ptr(m), _stream()), "az_mcts_begin")

    def search_select(self):
        N.check(self.L.az_mcts_select(self._h, _stream()), "az_mcts_select")

    def pending(self) -> int:
        r = C.c_int32()
        N.check(self.L.az_mcts_pending(self._h, C.byref(r), _stream()), "az_mcts_pending")
        return r.value

    def set_root_reuse(self, enable: bool):
        """Carry the logits of the played child over to the next search's root (see az_mcts_set_root_reuse)."""
        with torch.cuda.device(self.device):
            N.check(self.L.az_mcts_set_root_reuse(self._h, 1 if enable else 0), "az_mcts_set_root_reuse")

    def gather_rows(self) -> int:
        r = C.c_int32()
        N.check(self.L.az_mcts_gather_rows(self._h, C.byref(r), _stream()), "az_mcts_gather_rows")
        return r.value

    def set_async(self, enable: bool):
        """Asynchronous mode: gather/expand never wait for the device (see az_mcts_set_async)."""
        N.check(self.L.az_mcts_set_async(self._h, 1 if enable else 0), "az_mcts_set_async")

    def rows_device(self) -> int:
        """Device address of the pending-row count (int32)."""
        p = C.c_void_p()
        N.check(self.L.az_mcts_rows_device(self._h, C.byref(p)), "az_mcts_rows_device")
        return p.value

    def gather(self, rows: int, layout=N.AZ_FEAT_F32_NCHW, out=None):
        if out is None:
            if layout == N.AZ_FEAT_F32_NCHW:
                out = self._t((rows, self.Cp, self.n, self.n), torch.float32)
            else:
                out = self._t((rows, self.n * self.n, self.cpad), torch.bfloat16)
        N.check(self.L.az_mcts_gather(self._h, _ptr(out), layout, _stream()), "az_mcts_gather")
        return out

    def root_legal_counts(self):
        c = self._t((self.G,), torch.int32)
        N.check(self.L.az_mcts_root_legal_counts(self._h, _ptr(c), _stream()), "az_mcts_root_legal_counts")
        return c

    def expand(self, logits, values, noise=None) -> bool:
        """Returns True when the search has completed all simulations."""
        lg = None if logits is None else logits.to(device=self.device, dtype=torch.float32).contiguous()
        vl = None if values is None else values.to(device=self.device, dtype=torch.float32).contiguous()
        nz = None if noise is None else torch.as_tensor(noise, dtype=torch.float64, device=self.device).contiguous()
        self._keep = (lg, vl, nz)
        return N.check(self.L.az_mcts_expand(self._h, _ptr(lg), _ptr(vl), _ptr(nz), _stream()), "az_mcts_expand") == 1

    def result(self):
        v = self._t((self.G, self.A), torch.int32)
        rv = self._t((self.G,), torch.float64)
        N.check(self.L.az_mcts_result(self._h, _ptr(v), _ptr(rv), _stream()), "az_mcts_result")
        return v, rv

    def last_priors(self, rows: int):
        p = self._t((rows, self.A), torch.float64)
        N.check(self.L.az_mcts_last_priors(self._h, _ptr(p), _stream()), "az_mcts_last_priors")
        return p

    def stats(self):
        out = (C.c_int64 * 8)()
        N.check(self.L.az_mcts_stats(self._h, out, _stream()), "az_mcts_stats")
        keys = ("sims", "rows", "rows_raw", "depth_sum", "max_depth", "max_nodes", "terminal", "batches")
        return dict(zip(keys, [int(x) for x in out]))

    # ------------------------------------------------------------------ whole search with a host evaluator
    def _async_buffers(self):
        """Buffers of the sync-free batch loop (allocated once): features for max_rows rows, logits, values."""
        if getattr(self, "_abuf", None) is None:
            self._abuf = (self._t((self.max_rows, self.n * self.n, self.cpad), torch.bfloat16),
                          self._t((self.max_rows, self.A), torch.float32), self._t((self.max_rows,), torch.float32),
                          self.rows_device())
        return self._abuf

    def run_rounds(self, native_net, feat_buf, layout, logits_buf, values_buf, graph: bool = True):
        """All remaining batch rounds of the search in one native call (asynchronous mode; the root must be expanded).
        graph: replay a captured HIP graph of the rounds' launches when the network allows it."""
        return N.check(self.L.az_mcts_run_rounds(self._h, native_net._h, _ptr(feat_buf), int(layout), _ptr(logits_buf), _ptr(values_buf),
                                                 1 if graph else 0, _stream()), "az_mcts_run_rounds") == 1   # True: replayed from the graph

    def run_search(self, evaluator, add_noise=False, rngs=None, enable=None, layout=N.AZ_FEAT_F32_NCHW,
                   capture_priors=None, native_net=None):
        """mcts.py:165-225 for all games. `evaluator(features) -> (logits, values)` gets a device tensor in
        `layout` and returns tensors [rows, A] / [rows]. `rngs`: one numpy RandomState-like object per game
        (or a single one, used game after game) that supplies the root Dirichlet sample exactly where the
        upstream code draws it. Returns (visits int32 [G,A], root_value f64 [G]) on the device."""
        self.search_begin(enable)
        rows = self.pending()
        noise = None
        if rows > 0:
            feats = self.gather(rows, layout)
            logits, values = evaluator(feats)
            if add_noise:
                counts = self.root_legal_counts().cpu().numpy()
                noise_np = np.zeros((self.G, self.A), np.float64)
                for g in range(self.G):
                    k = int(counts[g])
                    if k > 0:
                        rng = rngs[g] if isinstance(rngs, (list, tuple)) else (rngs if rngs is not None else np.random)
                        noise_np[g, :k] = rng.dirichlet([self.alpha] * k)
                noise = torch.from_numpy(noise_np).to(self.device)
            done = self.expand(logits, values, noise)
            if capture_priors is not None:
                capture_priors.append(self.last_priors(rows).cpu().numpy())
        else:
            done = True
        if (not done and native_net is not None and capture_priors is None and layout == N.AZ_FEAT_BF16_NHWC
                and getattr(native_net, "indirect", False)):
            # the bf16 network reads the row count on the device: all batches are enqueued without a round trip
            fb, lb, vb, rows_dev = self._async_buffers()
            self.set_async(True)
            try:
                native_net.reserve(self.max_rows)
                self.run_rounds(native_net, fb, layout, lb, vb, graph=GRAPH_ROUNDS)
                out = self.result()
                self.stats()          # reports a failed select / exhausted node pool of the enqueued batches
            finally:
                self.set_async(False)
            return out
        while not done:
            self.search_select()
            rows = self.pending()
            if rows > 0:
                feats = self.gather(rows, layout)
                logits, values = evaluator(feats)
                done = self.expand(logits, values)
                if capture_priors is not None:
                    capture_priors.append(self.last_priors(rows).cpu().numpy())
            else:
                done = self.expand(None, None)
        return self.result()
