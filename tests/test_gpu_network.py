"""Network inference kernels (through the C ABI) against the PyTorch float32 module and the golden
outputs of the upstream AlphaZeroNet. Tolerances: float32 path 1e-5 (scaled by the logit magnitude
when that exceeds 1), bf16 path 1e-2 on policy probabilities and value."""
import numpy as np
import pytest
import torch

import golden_io
from tests_common import GOLDEN_NETS, build_golden_net

pytestmark = pytest.mark.gpu


def _to_nhwc_bf16(x, cpad=32):
    b, c, n, _ = x.shape
    out = torch.zeros((b, n * n, cpad), dtype=torch.bfloat16)
    out[:, :, :c] = torch.from_numpy(x).reshape(b, c, n * n).permute(0, 2, 1).to(torch.bfloat16)
    return out.contiguous()


@pytest.mark.parametrize("name", GOLDEN_NETS)
def test_f32_path_vs_golden_and_torch(name):
    z = golden_io.load("network")
    net, x, kw = build_golden_net(name)
    with torch.inference_mode():
        lg_t, v_t = net(torch.from_numpy(x))
    dev = torch.device("cuda:0")
    nn = net.to(dev).native("f32")
    lg, v = nn(torch.from_numpy(x).to(dev))
    lg, v = lg.cpu().numpy(), v.cpu().numpy()
    scale = max(1.0, float(np.abs(z[f"{name}/logits"]).max()))
    e_gold = np.abs(lg - z[f"{name}/logits"]).max()
    e_torch = np.abs(lg - lg_t.numpy()).max()
    e_val = np.abs(v - z[f"{name}/value"]).max()
    print(f"{name}: f32 logits err vs golden {e_gold:.3g} (scale {scale:.3g}), vs torch {e_torch:.3g}, value err {e_val:.3g}")
    assert e_gold <= 1e-5 * scale and e_val <= 1e-5


def _default_init(kw, seed=0):
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(seed)
    net = AlphaZeroNet(**kw)
    # non-trivial BN statistics, otherwise eval-mode BN is the identity
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.6, 1.4)
                m.weight.uniform_(0.8, 1.2)
                m.bias.uniform_(-0.1, 0.1)
    return net.eval()


@pytest.mark.parametrize("name", [n for n in GOLDEN_NETS if "n19" not in n])
@pytest.mark.parametrize("weights", ["default_init", "golden"])
def test_bf16_path(name, weights):
    net, x, kw = build_golden_net(name)
    if weights == "default_init":
        net = _default_init(kw)
    rng = np.random.default_rng(1)
    import stubnets
    c = 2 * kw.get("history_len", 8) + 5
    x = stubnets.synthetic_planes(37, c, kw["board_size"], seed=5)          # 37 rows: ragged last workgroup
    with torch.inference_mode():
        lg_t, v_t = net(torch.from_numpy(x))
    dev = torch.device("cuda:0")
    nn = net.to(dev).native("bf16")
    lg, v = nn(_to_nhwc_bf16(x).to(dev))
    lg, v = lg.cpu(), v.cpu()
    e_prob = float((lg.softmax(1) - lg_t.softmax(1)).abs().max())
    e_val = float((v - v_t).abs().max())
    e_lg = float((lg - lg_t).abs().max())
    scale = float(lg_t.abs().max())
    print(f"{name}/{weights}: bf16 prob err {e_prob:.3g}, value err {e_val:.3g}, logit err {e_lg:.3g} (logit scale {scale:.3g})")
    # bf16 carries 8 significant bits: the absolute bound of 1e-2 holds while |logit| <= 1 and scales
    # with the logit magnitude beyond that (the seed-generated "golden" weights reach |logit| ~ 37)
    if weights == "default_init":
        assert e_val <= 1e-2 and e_lg <= 1e-2 and e_prob <= 1e-2
    else:
        assert e_lg <= 2e-2 * max(1.0, scale) and e_val <= 3e-2 and e_prob <= 1e-2 * max(1.0, scale / 4)


def test_bf16_rows_are_batch_invariant():
    """A position's output must not depend on which other positions share the launch (dedup relies on it)."""
    net, x, kw = build_golden_net("n9_c64_b4")
    import stubnets
    x = stubnets.synthetic_planes(50, 21, 9, seed=9)
    dev = torch.device("cuda:0")
    nn = net.to(dev).native("bf16")
    f = _to_nhwc_bf16(x).to(dev)
    lg_all, v_all = nn(f)
    perm = torch.randperm(50)
    lg_p, v_p = nn(f[perm].contiguous())
    assert torch.equal(lg_all[perm], lg_p) and torch.equal(v_all[perm], v_p)
    lg_1, v_1 = nn(f[7:8].contiguous())
    assert torch.equal(lg_all[7:8], lg_1) and torch.equal(v_all[7:8], v_1)


def test_unsupported_bf16_shape_is_refused():
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd import _native as N
    net = AlphaZeroNet(board_size=19, channels=32, num_blocks=1).to("cuda:0")
    with pytest.raises(N.NativeError):
        net.native("bf16")


LAYERED = [dict(board_size=13, channels=64, num_blocks=2), dict(board_size=19, channels=128, num_blocks=2),
           dict(board_size=9, channels=256, num_blocks=2, value_hidden_dim=64),
           dict(board_size=19, channels=256, num_blocks=1, use_se=False), dict(board_size=19, channels=192, num_blocks=1, history_len=4)]


@pytest.mark.parametrize("kw", LAYERED, ids=lambda k: f"n{k['board_size']}_c{k['channels']}")
def test_bf16_layer_path(kw):
    """Boards above 9x9 and 256-channel towers run layer by layer (activations in HBM): same 1e-2 bound against the
    float32 torch module, same independence of a position's output from its neighbours in the batch."""
    from alphazero_amd.model import AlphaZeroNet
    import stubnets
    torch.manual_seed(11)
    net = AlphaZeroNet(**kw).eval()
    n, c = kw["board_size"], 2 * kw.get("history_len", 8) + 5
    x = stubnets.synthetic_planes(9, c, n, seed=5)
    with torch.inference_mode():
        lg_t, v_t = net(torch.from_numpy(x))
    dev = torch.device("cuda:0")
    nn = net.to(dev).native("bf16")
    assert not nn.fused
    f = _to_nhwc_bf16(x).to(dev)
    lg, v = nn(f)
    e_lg, e_val = float((lg.cpu() - lg_t).abs().max()), float((v.cpu() - v_t).abs().max())
    e_prob = float((lg.cpu().softmax(1) - lg_t.softmax(1)).abs().max())
    print(f"layered {kw}: logit err {e_lg:.3g}, value err {e_val:.3g}, prob err {e_prob:.3g}")
    assert e_lg <= 1e-2 and e_val <= 1e-2 and e_prob <= 1e-2
    perm = torch.randperm(9)
    lg_p, v_p = nn(f[perm].contiguous())
    assert torch.equal(lg[perm], lg_p) and torch.equal(v[perm], v_p)
    lg_1, v_1 = nn(f[4:5].contiguous())
    assert torch.equal(lg[4:5], lg_1) and torch.equal(v[4:5], v_1)


FUSED = [dict(board_size=9, channels=128, num_blocks=2), dict(board_size=7, channels=128, num_blocks=1, history_len=4, use_se=False),
         dict(board_size=5, channels=192, num_blocks=1, value_hidden_dim=64), dict(board_size=9, channels=192, num_blocks=2, use_se=False),
         dict(board_size=9, channels=32, num_blocks=2), dict(board_size=6, channels=64, num_blocks=2, history_len=2)]


@pytest.mark.parametrize("split", ["default", "0", "1"])
@pytest.mark.parametrize("kw", FUSED, ids=lambda k: f"n{k['board_size']}_c{k['channels']}")
def test_bf16_fused_tower_variants(kw, split, monkeypatch):
    """Every instantiation of the fused tower: weights staged through LDS (AZ_TOWER_SPLIT=0), weight fragments read from
    global memory (=1), and the default choice per channel count, on board sizes that put 3 to 10 boards in a workgroup."""
    from alphazero_amd.model import AlphaZeroNet
    import stubnets
    if split == "default":
        monkeypatch.delenv("AZ_TOWER_SPLIT", raising=False)
    else:
        monkeypatch.setenv("AZ_TOWER_SPLIT", split)
    net = _default_init(kw, seed=3)
    n, c = kw["board_size"], 2 * kw.get("history_len", 8) + 5
    x = stubnets.synthetic_planes(41, c, n, seed=6)
    with torch.inference_mode():
        lg_t, v_t = net(torch.from_numpy(x))
    dev = torch.device("cuda:0")
    nn = net.to(dev).native("bf16", shared=False)      # a handle of its own: the variant is chosen when the handle is made
    assert nn.fused
    f = _to_nhwc_bf16(x).to(dev)
    lg, v = nn(f)
    e_lg, e_val = float((lg.cpu() - lg_t).abs().max()), float((v.cpu() - v_t).abs().max())
    e_prob = float((lg.cpu().softmax(1) - lg_t.softmax(1)).abs().max())
    print(f"fused {kw} split={split}: logit err {e_lg:.3g}, value err {e_val:.3g}, prob err {e_prob:.3g}")
    assert e_lg <= 1e-2 and e_val <= 1e-2 and e_prob <= 1e-2
    perm = torch.randperm(41)
    lg_p, v_p = nn(f[perm].contiguous())
    assert torch.equal(lg[perm], lg_p) and torch.equal(v[perm], v_p)
