"""az_net_forward_static: the bf16 forward whose activation scratch is a device array inside the library, so the call
allocates nothing. Through the C ABI, against az_net_forward on the same inputs: bit-equal outputs on two calls in a
row, the second with fewer rows (the shared scratch keeps the first call's contents), the device's free memory unchanged
across the call, and a clear refusal past the scratch's capacity."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _feats(rows, n, seed):
    import stubnets
    x = stubnets.synthetic_planes(rows, 21, n, seed=seed)
    out = torch.zeros((rows, n * n, 32), dtype=torch.bfloat16)
    out[:, :, :21] = torch.from_numpy(x).reshape(rows, 21, n * n).permute(0, 2, 1).to(torch.bfloat16)
    return out.contiguous()


@pytest.mark.parametrize("shape", [dict(board_size=9, channels=192, num_blocks=3), dict(board_size=5, channels=64, num_blocks=2, use_se=False)],
                         ids=["n9_c192", "n5_c64"])
def test_forward_static_equals_forward(shape):
    from alphazero_amd import _native as N
    from alphazero_amd.model import AlphaZeroNet
    L = N.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    net = AlphaZeroNet(**shape).eval().to(dev)
    nn = net.native("bf16", shared=False)
    n, A = shape["board_size"], shape["board_size"] ** 2 + 1
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rows, seed in ((700, 1), (41, 2)):            # the second call is smaller and finds the first call's leftovers
        f = _feats(rows, n, seed).to(dev)
        want_lg, want_v = nn(f)                        # az_net_forward
        lg = torch.full((rows, A), 3.0, dtype=torch.float32, device=dev)
        v = torch.full((rows,), 3.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(dev)[0]
        rc = L.az_net_forward_static(nn._h, C.c_void_p(f.data_ptr()), rows, C.c_void_p(lg.data_ptr()), C.c_void_p(v.data_ptr()), st)
        assert rc == 0, L.az_last_error()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free_before          # the call took no device memory from the driver
        assert torch.equal(lg, want_lg) and torch.equal(v, want_v)
    cap = int(L.az_net_static_max_rows(nn._h))
    assert cap >= 2048
    f = torch.zeros((1, n * n, 32), dtype=torch.bfloat16, device=dev)
    rc = L.az_net_forward_static(nn._h, C.c_void_p(f.data_ptr()), cap + 1, C.c_void_p(lg.data_ptr()), C.c_void_p(v.data_ptr()), st)
    assert rc == N.AZ_ERR_CAPACITY and b"static scratch" in L.az_last_error()


def test_forward_static_refuses_layer_path():
    from alphazero_amd import _native as N
    from alphazero_amd.model import AlphaZeroNet
    L = N.lib()
    dev = torch.device("cuda:0")
    net = AlphaZeroNet(board_size=13, channels=64, num_blocks=1).eval().to(dev)
    nn = net.native("bf16", shared=False)
    f = torch.zeros((2, 169, 32), dtype=torch.bfloat16, device=dev)
    lg = torch.zeros((2, 170), dtype=torch.float32, device=dev)
    v = torch.zeros((2,), dtype=torch.float32, device=dev)
    rc = L.az_net_forward_static(nn._h, C.c_void_p(f.data_ptr()), 2, C.c_void_p(lg.data_ptr()), C.c_void_p(v.data_ptr()),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == N.AZ_ERR_ARG
    assert int(L.az_net_static_max_rows(nn._h)) == 0
