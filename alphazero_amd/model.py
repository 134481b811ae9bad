"""Policy/value network: the torch module keeps the upstream constructor, attributes and state_dict
keys (go_zero/model.py:43-97) so checkpoints interchange; inference on the GPU goes through the HIP
kernels of libazamd (NativeNet), training goes through torch autograd."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
from torch import nn

from . import _native as N


def feature_planes(history_len: int) -> int:
    """go.py:32-34: own history + opponent history + to-play, legal, last move, own/opp captures."""
    return 2 * history_len + 5


def _conv_bn_relu(cin: int, cout: int, k: int) -> list[nn.Module]:
    return [nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


class SqueezeExcite(nn.Module):
    """x * sigmoid(fc2(relu(fc1(mean_hw(x)))))  -- model.py:9-22; hidden = max(8, C // 8)."""

    def __init__(self, channels: int, reduction: int = 8):
        super().__init__()
        hidden = max(8, channels // reduction)
        # indices 1 and 3 carry the weights (state_dict keys se.net.1.*, se.net.3.*)
        self.net = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(channels, hidden, 1), nn.ReLU(inplace=True),
                                 nn.Conv2d(hidden, channels, 1), nn.Sigmoid())

    def forward(self, x):
        return x * self.net(x)


class ResidualBlock(nn.Module):
    """relu(se(bn2(conv2(relu(bn1(conv1(x)))))) + x)  -- model.py:25-40."""

    def __init__(self, channels: int, use_se: bool = True):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)
        self.se = SqueezeExcite(channels) if use_se else nn.Identity()
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.se(self.bn2(self.conv2(y)))
        return self.relu(y + x)


class AlphaZeroNet(nn.Module):
    def __init__(self, board_size: int = 9, history_len: int = 8, input_channels: int | None = None,
                 channels: int = 192, num_blocks: int = 12, use_se: bool = True, value_hidden_dim: int = 256):
        super().__init__()
        self.board_size = board_size
        self.action_dim = board_size * board_size + 1
        value_hidden_dim = max(1, int(value_hidden_dim))
        cin = feature_planes(history_len) if input_channels is None else input_channels
        pts, quarter = board_size * board_size, channels // 4
        self.stem = nn.Sequential(*_conv_bn_relu(cin, channels, 3))
        self.tower = nn.Sequential(*(ResidualBlock(channels, use_se) for _ in range(num_blocks)))
        self.policy_head = nn.Sequential(*_conv_bn_relu(channels, quarter, 1), *_conv_bn_relu(quarter, 2, 1))
        self.policy_fc = nn.Linear(2 * pts, self.action_dim)
        self.value_head = nn.Sequential(*_conv_bn_relu(channels, quarter, 1), nn.Flatten(),
                                        nn.Linear(quarter * pts, value_hidden_dim), nn.ReLU(inplace=True),
                                        nn.Linear(value_hidden_dim, 1), nn.Tanh())
        self._hyper = dict(board_size=board_size, in_planes=cin, channels=channels, blocks=num_blocks,
                           use_se=bool(use_se), se_hidden=max(8, channels // 8), value_hidden=value_hidden_dim)
        self._native: dict = {}

    def forward(self, x):
        """(policy_logits [b, A], value [b])  -- model.py:89-97."""
        t = self.tower(self.stem(x))
        return self.policy_fc(self.policy_head(t).flatten(1)), self.value_head(t).squeeze(1)

    # ------------------------------------------------------------------ native inference
    def flat_parameters(self) -> np.ndarray:
        """All tensors of the state_dict except num_batches_tracked, registration order, float32."""
        parts = [v.detach().to("cpu", torch.float32).reshape(-1).numpy()
                 for k, v in self.state_dict().items() if not k.endswith("num_batches_tracked")]
        return np.ascontiguousarray(np.concatenate(parts))

    def _fingerprint(self):
        return tuple(int(v._version) for v in self.state_dict(keep_vars=True).values())

    def native(self, precision: str = "bf16", device=None, shared: bool = True) -> "NativeNet":
        """HIP inference engine holding a (BN-folded) copy of the current weights; rebuilt when a
        parameter tensor has been modified in place or replaced since the last call. shared=False
        returns a private handle (own activation scratch) for a caller that runs it on its own stream."""
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != "cuda":
            raise N.NativeError("native inference needs the model on a GPU (no CPU fallback)")
        key = (precision, device.index if device.index is not None else torch.cuda.current_device())
        if not shared:
            return NativeNet(self._hyper, self.flat_parameters(), precision, key[1])
        fp = self._fingerprint()
        ent = self._native.get(key)
        if ent is None or ent[0] != fp:
            if ent is not None:
                ent[1].close()
            ent = (fp, NativeNet(self._hyper, self.flat_parameters(), precision, key[1]))
            self._native[key] = ent
        return ent[1]


def _native_auto(self, device=None, shared: bool = True) -> "NativeNet":
    """native("bf16") where the MFMA kernels cover the shape (channels 32/64/128/192 on boards up to 9x9; 64/128/192/256
    with at most 32 input planes on boards up to 19x19), else native("f32"): every shape upstream's CLI accepts runs."""
    try:
        return self.native("bf16", device, shared)
    except N.NativeError as e:
        if "unsupported" not in str(e) and "LDS budget" not in str(e) and "too large" not in str(e):
            raise
        return self.native("f32", device, shared)


AlphaZeroNet.native_auto = _native_auto

PolicyValueNet = AlphaZeroNet  # name used in some descriptions of the upstream project


class NativeNet:
    """az_net_* handle: eval-mode forward of AlphaZeroNet on the GPU (bf16 MFMA or float32 kernels)."""

    def __init__(self, hyper: dict, params: np.ndarray, precision: str = "bf16", device_index: int | None = None):
        if not torch.cuda.is_available():
            raise N.NativeError("alphazero_amd.NativeNet needs a GPU (no CPU fallback)")
        self.L = N.lib()
        self.precision = precision
        self.dev_index = torch.cuda.current_device() if device_index is None else int(device_index)
        self.device = torch.device("cuda", self.dev_index)
        self.n = hyper["board_size"]
        self.A = self.n * self.n + 1
        self.in_planes = hyper["in_planes"]
        self.layout = N.AZ_FEAT_BF16_NHWC if precision == "bf16" else N.AZ_FEAT_F32_NCHW
        cfg = N.AzNetConfig(hyper["board_size"], hyper["in_planes"], hyper["channels"], hyper["blocks"],
                            1 if hyper["use_se"] else 0, hyper["se_hidden"], hyper["value_hidden"],
                            N.AZ_NET_BF16 if precision == "bf16" else N.AZ_NET_F32, self.dev_index)
        h = C.c_void_p()
        N.check(self.L.az_net_create(C.byref(cfg), C.byref(h)), "az_net_create")
        self._h = h
        p = np.ascontiguousarray(params, np.float32)
        want = self.L.az_net_param_count(h)
        if p.size != want:
            self.close()
            raise ValueError(f"parameter blob has {p.size} values, network needs {want}")
        with torch.cuda.device(self.device):
            N.check(self.L.az_net_load(h, p.ctypes.data_as(C.c_void_p), p.size), "az_net_load")
        f = C.c_int32()
        N.check(self.L.az_net_is_fused(h, C.byref(f)), "az_net_is_fused")
        self.fused = bool(f.value)     # single fused tower kernel vs layer by layer (larger boards, 256 channels) / float32
        self.indirect = precision == "bf16"   # forward_indirect: the row count is read on the device

    def close(self):
        if getattr(self, "_h", None):
            self.L.az_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, rows: int):
        """Allocate the activation scratch for forwards of up to `rows` positions now: later forwards of that size
        allocate nothing and never wait for the device."""
        with torch.cuda.device(self.device):
            N.check(self.L.az_net_reserve(self._h, int(rows)), "az_net_reserve")

    def indirect_passes(self, max_rows: int) -> int:
        """Launch sequences forward_indirect makes for an upper bound of max_rows (1: no empty chunk passes)."""
        return int(self.L.az_net_indirect_passes(self._h, int(max_rows)))

    def static_max_rows(self) -> int:
        return int(self.L.az_net_static_max_rows(self._h))

    def forward_static(self, feats: torch.Tensor, logits: torch.Tensor, values: torch.Tensor, rows: int | None = None):
        """forward() into caller-provided float32 buffers with the library's own static device scratch: the call
        allocates nothing (az_net_forward_static). Calls that overlap on one GPU share that scratch and race."""
        rows = feats.shape[0] if rows is None else rows
        assert feats.dtype == torch.bfloat16 and feats.is_contiguous() and feats.shape[0] >= rows
        assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.shape[0] >= rows
        assert values.dtype == torch.float32 and values.is_contiguous() and values.shape[0] >= rows
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(self.L.az_net_forward_static(self._h, C.c_void_p(feats.data_ptr()), rows, C.c_void_p(logits.data_ptr()),
                                             C.c_void_p(values.data_ptr()), st), "az_net_forward_static")

    def forward(self, feats: torch.Tensor, rows: int | None = None):
        """feats: device tensor in this precision's layout (bf16 [rows, n*n, cpad] or f32 [rows, C, n, n])."""
        rows = feats.shape[0] if rows is None else rows
        if self.precision == "bf16":
            assert feats.dtype == torch.bfloat16 and feats.is_contiguous()
        else:
            assert feats.dtype == torch.float32 and feats.is_contiguous()
        logits = torch.empty((rows, self.A), dtype=torch.float32, device=feats.device)
        values = torch.empty((rows,), dtype=torch.float32, device=feats.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(self.L.az_net_forward(self._h, C.c_void_p(feats.data_ptr()), rows, C.c_void_p(logits.data_ptr()),
                                      C.c_void_p(values.data_ptr()), st), "az_net_forward")
        return logits, values

    __call__ = forward

    def forward_indirect(self, feats: torch.Tensor, rows_dev: int, max_rows: int, logits: torch.Tensor, values: torch.Tensor):
        """bf16 forward of the first *rows_dev rows (an int32 in device memory, <= max_rows) into the given
        float32 buffers [max_rows, A] / [max_rows]; nothing waits for the device."""
        assert feats.dtype == torch.bfloat16 and feats.is_contiguous() and feats.shape[0] >= max_rows
        assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.shape[0] >= max_rows
        assert values.dtype == torch.float32 and values.is_contiguous() and values.shape[0] >= max_rows
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(self.L.az_net_forward_indirect(self._h, C.c_void_p(feats.data_ptr()), C.c_void_p(rows_dev), max_rows,
                                               C.c_void_p(logits.data_ptr()), C.c_void_p(values.data_ptr()), st),
                "az_net_forward_indirect")

    def set_device_choice(self, on: bool):
        """Whether forward_indirect may launch the low-latency tower variant beside the main kernel and let the row count
        on the device pick one (default on; drivers with three or more concurrent streams switch it off)."""
        N.check(self.L.az_net_set_device_choice(self._h, 1 if on else 0), "az_net_set_device_choice")

    def set_tower_workgroups(self, workgroups: int):
        N.check(self.L.az_net_set_tower_workgroups(self._h, int(workgroups)), "az_net_set_tower_workgroups")

    def profile_spans(self) -> np.ndarray:
        """[launches, 2] start/end ms of the timed tower launches (device-wide time line); call before
        profile(..., read=True), which clears them."""
        n = C.c_int64()
        N.check(self.L.az_net_profile_spans(self._h, None, 0, C.byref(n)), "az_net_profile_spans")
        out = np.zeros((max(1, n.value), 2), np.float64)
        N.check(self.L.az_net_profile_spans(self._h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)), "az_net_profile_spans")
        return out[: n.value]

    def profile(self, enable: bool, read: bool = False):
        """Switch HIP-event timing of the tower kernel on/off; with read=True first return
        (total_ms, launches, rows) accumulated since it was enabled."""
        if read:
            ms, ln, rw = C.c_double(), C.c_int64(), C.c_int64()
            N.check(self.L.az_net_profile(self._h, 1 if enable else 0, C.byref(ms), C.byref(ln), C.byref(rw)), "az_net_profile")
            return ms.value, ln.value, rw.value
        N.check(self.L.az_net_profile(self._h, 1 if enable else 0, None, None, None), "az_net_profile")
        return None
