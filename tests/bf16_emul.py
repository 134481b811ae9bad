"""PyTorch (CPU) emulation of the bf16 MFMA inference kernels that rounds where the kernels round.

Test infrastructure: it separates bf16's own error from kernel error. The HIP tower must agree with
`emulate(net, x, ...)` to within a few bf16 units of the activations (float32 accumulation order is the
only difference), whatever the weights; the distance of the emulation from the float32 module is the
error bf16 inference itself carries on those weights.

Rounding points of the kernels (alphazero_amd/csrc/az_net_mfma.hip, az_net_layers.hip):
  * every convolution's weights: BN scale folded in float64, rounded once to bf16; biases stay float32;
  * input planes are bf16;
  * float32 accumulation; relu(acc + bias) rounded to bf16 after the stem, after conv1 of a block and
    after both 1x1 head convolutions;
  * conv2 of a block: y = acc + bias stays float32; the squeeze-excite pooling reads a bf16 image of y
    (pooled in float32), fc1/fc2 weights are bf16, their biases float32; gate in float32;
  * x_next = relu(y * gate + x): rounded to bf16 for the next convolution; the copy of x that is added at the
    end of the next block is that bf16 image (`residual="bf16"`, what the kernels do) or, for the error budget only,
    a float32 copy (`residual="f32"`);
  * policy tail: float32 weights on the bf16 head activations; value fc1 weights bf16, the rest float32.
Switches allow the error budget to be taken apart: weights only, + activations, + residual stream.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _r(t: torch.Tensor, on: bool = True) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32) if on else t


def _fold(conv_w, bn, eps=1e-5):
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + eps)
    w = (conv_w.double() * s.view(-1, 1, 1, 1)).float()
    b = (bn.bias.double() - bn.running_mean.double() * s).float()
    return w, b


@torch.no_grad()
def emulate(net, x: torch.Tensor, weights: bool = True, activations: bool = True, residual: str = "bf16",
            se_image: bool | None = None, path: str = "fused", accumulate: str = "f32"):
    """(logits, value) of AlphaZeroNet `net` (eval mode) on float32 NCHW planes `x`.

    weights: round folded conv / SE / value-fc1 weights to bf16. activations: round the activations where the
    kernels do (and the input planes). residual: "f32" keeps the skip connection's copy in float32, "bf16" uses the
    rounded image (what the next convolution reads). se_image: pool over the bf16 image of y (default: = activations).
    path: "fused" = the single-kernel tower (boards up to 9x9, up to 192 channels); "layers" = the layer-by-layer
    kernels (larger boards, 256 channels), which differ in three places: y itself is stored as bf16 before the gate is
    applied, the squeeze-excite weights stay float32, and the skip connection is always the bf16 image.
    accumulate: "f32", or "f64" = every convolution accumulates in float64 and is rounded to float32 once: the distance
    between the two is how much these weights amplify a change of summation order alone (flipped bf16 roundings that
    propagate), i.e. the scale below which two correct implementations cannot be told apart on them.
    """
    def conv(a, w, b=None, **k):
        if accumulate == "f64":
            return F.conv2d(a.double(), w.double(), None if b is None else b.double(), **k).float()
        return F.conv2d(a, w, b, **k)

    if se_image is None:
        se_image = activations
    layers = path == "layers"
    if layers:
        residual = "bf16"
    x = _r(x.float(), activations)
    w, b = _fold(net.stem[0].weight, net.stem[1])
    t = F.relu(conv(x, _r(w, weights), b, padding=1))
    skip = t if residual == "f32" else _r(t, activations)
    t = _r(t, activations)
    for blk in net.tower:
        w1, b1 = _fold(blk.conv1.weight, blk.bn1)
        w2, b2 = _fold(blk.conv2.weight, blk.bn2)
        h = _r(F.relu(conv(t, _r(w1, weights), b1, padding=1)), activations)
        y = conv(h, _r(w2, weights), b2, padding=1)
        if layers:
            y = _r(y, activations)
        if not isinstance(blk.se, torch.nn.Identity):
            fc1, fc2 = blk.se.net[1], blk.se.net[3]
            pooled = _r(y, se_image).mean(dim=(2, 3), keepdim=True)
            hid = F.relu(conv(pooled, _r(fc1.weight, weights and not layers), fc1.bias))
            y = y * torch.sigmoid(conv(hid, _r(fc2.weight, weights and not layers), fc2.bias))
        nxt = F.relu(y + skip)
        skip = nxt if residual == "f32" else _r(nxt, activations)
        t = _r(nxt, activations)
    wp, bp = _fold(net.policy_head[0].weight, net.policy_head[1])
    wv, bv = _fold(net.value_head[0].weight, net.value_head[1])
    p = _r(F.relu(conv(t, _r(wp, weights), bp)), activations)
    v = _r(F.relu(conv(t, _r(wv, weights), bv)), activations)
    w3, b3 = _fold(net.policy_head[3].weight, net.policy_head[4])
    p = F.relu(conv(p, w3, b3)).flatten(1)
    logits = F.linear(p, net.policy_fc.weight, net.policy_fc.bias)
    fc4, fc6 = net.value_head[4], net.value_head[6]
    hv = F.relu(F.linear(v.flatten(1), _r(fc4.weight, weights), fc4.bias))
    value = torch.tanh(F.linear(hv, fc6.weight, fc6.bias)).squeeze(1)
    return logits, value


def errors(lg, v, lg_ref, v_ref):
    """(max |prob|, max |value|, max |logit|) differences."""
    return (float((lg.softmax(1) - lg_ref.softmax(1)).abs().max()), float((v - v_ref).abs().max()),
            float((lg - lg_ref).abs().max()))
