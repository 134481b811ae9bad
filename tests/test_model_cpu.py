"""Host-side torch module: state_dict contract and outputs against golden vectors from the upstream
AlphaZeroNet (model.py) with seed-generated weights."""
import json

import numpy as np
import pytest
import torch

import golden_io
import stubnets
from alphazero_amd.model import AlphaZeroNet, PolicyValueNet
from tests_common import build_golden_net, build_realistic_net, GOLDEN_NETS, REALISTIC_NETS


def test_alias():
    assert PolicyValueNet is AlphaZeroNet


@pytest.mark.parametrize("name", GOLDEN_NETS)
def test_state_dict_contract_and_outputs(name):
    z = golden_io.load_network(name)
    net, x, kw = build_golden_net(name)
    shapes = json.loads(bytes(z[f"{name}/shapes_json"]).decode())
    sd = net.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == shapes        # same keys, order and shapes
    assert sum(p.numel() for p in net.parameters()) == int(z[f"{name}/nparams"][0])
    assert net.board_size == kw["board_size"] and net.action_dim == kw["board_size"] ** 2 + 1
    with torch.inference_mode():
        lg, v = net(torch.from_numpy(x))
    scale = max(1.0, float(np.abs(z[f"{name}/logits"]).max()))   # float32 summation order varies with the thread count
    assert np.abs(lg.numpy() - z[f"{name}/logits"]).max() < 1e-5 * scale
    assert np.abs(v.numpy() - z[f"{name}/value"]).max() < 2e-5


@pytest.mark.parametrize("name", [n for n in REALISTIC_NETS if n != "real_n19_c256_b20"])   # the 20-block tower: GPU tests only (CPU time)
def test_realistic_weights_outputs(name):
    """The host module with the realistic-scale generated weights reproduces the upstream module's stored outputs."""
    net, x, kw, lg_ref, v_ref = build_realistic_net(name)
    with torch.inference_mode():
        lg, v = net(torch.from_numpy(x))
    assert np.abs(lg.numpy() - lg_ref).max() < 1e-5 * max(1.0, float(np.abs(lg_ref).max()))
    assert np.abs(v.numpy() - v_ref).max() < 2e-5


def test_flat_parameter_blob_length():
    net, _, _ = build_golden_net("n5_c32_b1")
    blob = net.flat_parameters()
    assert blob.dtype == np.float32
    n_bn = sum(1 for k in net.state_dict() if k.endswith("num_batches_tracked"))
    assert blob.size == sum(v.numel() for v in net.state_dict().values()) - n_bn
