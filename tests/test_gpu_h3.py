"""-m gpu: the r6 operand split of the matrix-pipe convolutions (csrc/conv_x6.h "h3": two fp16 pieces per operand scaled by the
operand tensor's max |x|, three products) — the amax pass and its slot protocol, the route against fp64 and against the six-product
route on well and badly scaled operands, the register-operand kernel against the LDS-staged one, amax words folded by producers.
Reference call sites: mmdet3d/models/backbones/me_resnet.py:56-62 (MinkowskiEngine computes these convolutions in fp32)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import me_oracle as mo
from oracle import x6_oracle as X

pytestmark = pytest.mark.gpu
SLOT_WORDS = 512           # include/fcaf3d_hip.h FC_AMAX_SLOT_BYTES / 4


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _slot_max(slot):
    """what the kernels read from a slot: the maximum of its 32 sub-words (fc_common.h fc_amax_read), as a float (bit patterns of
    non-negative floats order like integers)"""
    sub = slot.cpu().view(32, 16)[:, 0]
    return float(sub.max().reshape(1).view(torch.float32)[0])


def test_amax_pass_and_slot_protocol():
    """fc_amax: the largest FINITE |x| as a bit pattern in sub-word 0, scratch words back to zero, any element count, a slot can be
    used again; bit-exact against numpy (oracle/x6_oracle.py amax_finite)"""
    from fcaf3d_amd import _lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    slot = torch.zeros(SLOT_WORDS, dtype=torch.int32, device=dev)
    for n in (0, 1, 3, 4, 5, 1023, 4096, 100003, 437248 * 64):
        x = (torch.randn(n, generator=g) * torch.exp(2 * torch.randn(n, generator=g))).to(dev)
        if n > 10:
            x[7] = float('inf'); x[n // 2] = float('nan'); x[n - 1] = -float('inf')
        for rep in range(2):
            L.call('fc_amax', L.ptr(x), n, L.ptr(slot), L.stream())
            torch.cuda.synchronize()
            s = slot.cpu()
            want = X.amax_finite(x.cpu().numpy())
            got = s[:1].view(torch.float32).numpy()[0]
            assert got == want, (n, rep, got, want)
            assert int(s[1]) == 0 and int(s[2]) == 0 and not s[3:].any(), (n, s[:4])
            assert _slot_max(s) == float(want)


def _case(dev, seed, n_points, q, B=2):
    from fcaf3d_amd.sparse import CoordMap
    from tests.test_gpu_ops import _scene_coords
    _, c_ref, _ = _scene_coords(seed, n_points=n_points, B=B)
    if q > 1:
        c_ref = c_ref.copy(); c_ref[:, 1:] = np.floor_divide(c_ref[:, 1:], q) * q
    uc, _, _ = mo.unique_first(c_ref)
    cm, _, _ = CoordMap.from_coords(torch.from_numpy(uc).to(dev), q, B)
    return cm, cm.kernel_map(cm, 3), len(uc)


@pytest.mark.parametrize('xs,gs', [(1.0, 1.0), (3e5, 2e-9), (1e-12, 7e10)])
def test_three_product_route_against_fp64_and_the_six_product_route(xs, gs):
    """forward, backward-data and backward-weights of a 27-offset convolution in mode 2 (three fp16 products) and mode 0 (six bf16
    products) against the same computation in fp64, on heavy-tailed operands of very different magnitudes (activations x xs,
    gradients x gs: the power-of-two scales must absorb them): mode 2 within 1.5x of mode 0 and below 1.5e-7 of the tensor scale."""
    import fcaf3d_amd.functional as Fn
    dev = _dev()
    cm, km, n = _case(dev, 11, 60000, 2)
    nbr = km.nbr.cpu().numpy()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(n, 64, generator=g) * torch.exp(torch.randn(n, 64, generator=g))).clamp(min=0) * xs
    w = torch.randn(27, 64, 128, generator=g) / np.sqrt(27 * 64)
    go = torch.randn(n, 128, generator=g) * torch.exp(1.5 * torch.randn(n, 128, generator=g)) * gs
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    out64 = mo.conv(x64, w64, nbr)
    gx64, gw64 = torch.autograd.grad(out64, [x64, w64], go.double())
    errs = {}
    try:
        for mode in (2, 0):
            Fn.set_split_mode(mode)
            xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
            out = Fn.sparse_conv(xg, wg, km, cm.n)
            gx, gw = torch.autograd.grad(out, [xg, wg], go.to(dev))
            assert torch.isfinite(out).all() and torch.isfinite(gx).all() and torch.isfinite(gw).all()
            errs[mode] = [float((a.double().cpu() - b).pow(2).mean().sqrt() / b.abs().max())
                          for a, b in ((out, out64.detach()), (gx, gx64), (gw, gw64))]
    finally:
        Fn.set_split_mode(2)
    print(f'x scale {xs:g}, gradient scale {gs:g}: rms error / tensor scale vs fp64 (fwd, dgrad, wgrad): three fp16 products {errs[2]}, '
          f'six bf16 products {errs[0]}')
    for e3, e6 in zip(errs[2], errs[0]):
        assert e3 <= 1.5 * e6 + 1e-9 and e3 < 1.5e-7, (errs[2], errs[0])


@pytest.mark.parametrize('n_points,Cin,Cout,q', [(100000, 64, 64, 4), (100000, 64, 128, 4), (100000, 128, 128, 8), (100000, 256, 256, 32),
                                                  (3000, 128, 128, 8), (777, 128, 64, 2)])
def test_register_operand_kernel_is_bit_identical(n_points, Cin, Cout, q):
    """csrc/conv_h3r.h: the gathered operand split in registers (one row per lane) instead of staged through LDS — the same pieces
    and products in the same order per accumulator: forward and backward-data bit for bit (dense tables, mask-sorted tables, pair
    lists, ragged tiles; buffer and flat addressing); the statistics epilogue agrees to rounding (other summation order).  The weight
    gradient rides along: k_wgrad_x6t reads its rows through buffer descriptors or flat addresses (flags bit27) — the same bits."""
    from fcaf3d_amd import _lib as L
    import fcaf3d_amd.functional as Fn
    from tests.test_gpu_ops import _stats_case
    dev = _dev()
    x, w, km, cm = _stats_case(dev, n_points, Cin, Cout, q, seed=41)
    g = torch.randn(cm.n, Cout, generator=torch.Generator().manual_seed(6)).to(dev)
    res = {}
    f0 = Fn.FLAGS
    try:
        for mode in (0, 2):
            for flat in (False, True):
                L.call('fc_debug_set_h3r', mode)
                Fn.FLAGS = f0 | ((1 << 27) if flat else 0)
                xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
                y, tab = Fn.sparse_conv(xx, ww, km, cm.n, True, want_stats=True)
                y.backward(g)
                res[(mode, flat)] = (y.detach().clone(), xx.grad.clone(), None if tab is None else tab.clone(), ww.grad.clone())
    finally:
        Fn.FLAGS = f0
        L.call('fc_debug_set_h3r', 1)
    ref = res[(0, False)]
    assert torch.isfinite(ref[0]).all() and float(ref[0].abs().max()) > 0
    for key, (y, gx, tab, gw) in res.items():
        assert torch.equal(y, ref[0]), ('forward', key)
        assert torch.equal(gx, ref[1]), ('backward data', key)
        assert torch.equal(gw, ref[3]), ('backward weights (buffer / flat addressing of k_wgrad_x6t)', key)
        if tab is not None:
            a, b = tab.double().sum(0), ref[2].double().sum(0)
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), ('statistics', key)


def test_producers_fold_the_amax_of_what_they_write():
    """fc_amax_out_hint: the normalisation and max-pooling apply kernels leave max |output| in the caller's (zeroed) slot — the same
    number fc_amax finds — so the executor's convolutions need no pass of their own (the executor-vs-module-path tests hold the
    whole program to the per-operator path bit for bit; here the two entry points with short signatures)."""
    from fcaf3d_amd import _lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    for n, C in ((100003, 64), (437, 128), (5, 64)):
        x = (torch.randn(n, C, generator=g) * 3 + 1).to(dev)
        mean, var = x.mean(0).contiguous(), x.var(0, unbiased=False).contiguous()
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
        y = torch.empty_like(x)
        slot = torch.zeros(SLOT_WORDS, dtype=torch.int32, device=dev)
        L.call('fc_amax_out_hint', L.ptr(slot))
        L.call('fc_norm_act_fwd', L.ptr(x), None, 0, n, C, L.ptr(mean), L.ptr(var), 1e-5, L.ptr(gamma), L.ptr(beta), None, 2, L.ptr(y),
               L.stream())
        torch.cuda.synchronize()
        assert _slot_max(slot.cpu()) == float(y.abs().max()), (n, C)
        # the hint is consumed: a second call leaves a fresh slot alone
        slot2 = torch.zeros(SLOT_WORDS, dtype=torch.int32, device=dev)
        L.call('fc_norm_act_fwd', L.ptr(x), None, 0, n, C, L.ptr(mean), L.ptr(var), 1e-5, L.ptr(gamma), L.ptr(beta), None, 2, L.ptr(y),
               L.stream())
        torch.cuda.synchronize()
        assert not slot2.cpu().any()
