"""Pins oracle/me_oracle.py against an INDEPENDENT dense-grid oracle
(torch.nn.functional conv3d / conv_transpose3d / max_pool3d + autograd) using the
index algebra of SURVEY.md Appendix A.9.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import me_oracle as mo

G = 12


def _random_sparse(seed, C, occupancy=0.15, B=2, T=1):
    rng = np.random.default_rng(seed)
    occ = rng.random((B, G, G, G)) < occupancy       # [b, z, y, x]
    b, z, y, x = np.nonzero(occ)
    perm = rng.permutation(len(b))
    coords = np.stack([b, x * T, y * T, z * T], 1).astype(np.int32)[perm]
    feats = torch.from_numpy(rng.standard_normal((len(coords), C)).astype(np.float64))
    return coords, feats


def _to_dense(coords, feats, T, fill=0.0, B=2, n=G):
    d = torch.full((B, feats.shape[1], n, n, n), fill, dtype=feats.dtype)
    c = torch.from_numpy(coords.astype(np.int64))
    d[c[:, 0], :, c[:, 3] // T, c[:, 2] // T, c[:, 1] // T] = feats
    return d


def _read(dense, coords, T):
    c = torch.from_numpy(coords.astype(np.int64))
    return dense[c[:, 0], :, c[:, 3] // T, c[:, 2] // T, c[:, 1] // T]


def _dense_w3(W):   # (27,Ci,Co) -> (Co,Ci,3,3,3) with [dz,dy,dx]
    K, Ci, Co = W.shape
    return W.reshape(3, 3, 3, Ci, Co).permute(4, 3, 0, 1, 2).contiguous()


@pytest.mark.parametrize('seed', [0, 1])
def test_conv_k3s1_fwd_bwd(seed):
    coords, feats = _random_sparse(seed, 5)
    W = torch.randn(27, 5, 7, dtype=torch.float64)
    feats.requires_grad_(True); W.requires_grad_(True)
    nbr = mo.kernel_map(coords, coords, mo.kernel_offsets(3, 1))
    out = mo.conv(feats, W, nbr)
    g = torch.randn_like(out)
    gf, gw = torch.autograd.grad(out, [feats, W], g)
    f2 = feats.detach().clone().requires_grad_(True); W2 = W.detach().clone().requires_grad_(True)
    dout = _read(F.conv3d(_to_dense(coords, f2, 1), _dense_w3(W2), padding=1), coords, 1)
    gf2, gw2 = torch.autograd.grad(dout, [f2, W2], g)
    assert torch.allclose(out, dout, atol=1e-12)
    assert torch.allclose(gf, gf2, atol=1e-12)
    assert torch.allclose(gw, gw2, atol=1e-12)


def test_conv_k3s2_and_k1s2():
    coords, feats = _random_sparse(3, 4)
    out_c = mo.stride_coords(coords, 1, 2)
    W = torch.randn(27, 4, 6, dtype=torch.float64)
    nbr = mo.kernel_map(coords, out_c, mo.kernel_offsets(3, 1))
    out = mo.conv(feats, W, nbr)
    dense = F.conv3d(_to_dense(coords, feats, 1), _dense_w3(W), stride=2, padding=1)
    assert torch.allclose(out, _read(dense, out_c, 2), atol=1e-12)
    # k1 s2: single offset 0 -> only voxels on the coarse lattice contribute
    W1 = torch.randn(1, 4, 6, dtype=torch.float64)
    nbr1 = mo.kernel_map(coords, out_c, mo.kernel_offsets(1, 1))
    out1 = mo.conv(feats, W1, nbr1)
    dense1 = F.conv3d(_to_dense(coords, feats, 1), W1[0].t().reshape(6, 4, 1, 1, 1), stride=2)
    assert torch.allclose(out1, _read(dense1, out_c, 2), atol=1e-12)
    assert 0 < (nbr1 >= 0).sum() < nbr1.shape[1]


def test_maxpool_k2s2():
    coords, feats = _random_sparse(4, 3)
    out_c = mo.stride_coords(coords, 1, 2)
    nbr = mo.kernel_map(coords, out_c, mo.kernel_offsets(2, 1))
    out = mo.max_pool(feats, nbr)
    dense = F.max_pool3d(_to_dense(coords, feats, 1, fill=-float('inf')), 2, 2)
    assert torch.equal(out, _read(dense, out_c, 2))


def test_gen_conv_transpose():
    coords, feats = _random_sparse(5, 4, T=2)
    W = torch.randn(8, 4, 3, dtype=torch.float64)
    out_c = mo.gen_conv_transpose_coords(coords, 2)
    assert len(np.unique(mo.pack_keys(out_c))) == len(out_c) == 8 * len(coords)
    out = mo.gen_conv_transpose(feats, W)
    Wd = W.reshape(2, 2, 2, 4, 3).permute(3, 4, 0, 1, 2).contiguous()   # (Ci,Co,dz,dy,dx)
    dense = F.conv_transpose3d(_to_dense(coords, feats, 2), Wd, stride=2)
    c = torch.from_numpy(out_c.astype(np.int64))
    assert torch.allclose(out, dense[c[:, 0], :, c[:, 3], c[:, 2], c[:, 1]], atol=1e-12)


def test_first_occurrence_and_negative_floor():
    pts = [np.array([[-0.5, 0.2, 0.1], [0.4, 0.2, 0.1], [-0.1, 0.9, 0.3], [0.6, 0.7, 0.2]], np.float32)]
    f = [np.arange(4, dtype=np.float32)[:, None]]
    c, ff = mo.batch_sparse_collate(pts, f)
    uc, uf = mo.sparse_tensor(c, ff)
    assert uc.tolist() == [[0, -1, 0, 0], [0, 0, 0, 0]]
    assert uf[:, 0].tolist() == [0.0, 1.0]
    sc = mo.stride_coords(np.array([[0, -1, 3, 5], [0, -2, 2, 4], [0, 1, 1, 1]], np.int32), 1, 2)
    assert sc.tolist() == [[0, -2, 2, 4], [0, 0, 0, 0]]


def test_union_interp_prune():
    a_c = np.array([[0, 0, 0, 0], [0, 2, 0, 0]], np.int32)
    b_c = np.array([[0, 2, 0, 0], [0, 4, 0, 0], [0, 0, 0, 0]], np.int32)
    a_f = torch.tensor([[1.0], [2.0]]); b_f = torch.tensor([[10.0], [20.0], [30.0]])
    uc, uf = mo.union_add(a_c, a_f, b_c, b_f)              # a inside b: b's set in b's order
    assert uc.tolist() == [[0, 2, 0, 0], [0, 4, 0, 0], [0, 0, 0, 0]]
    assert uf[:, 0].tolist() == [12.0, 20.0, 31.0]
    c_c = np.array([[0, 2, 0, 0], [0, 4, 0, 0]], np.int32)  # general case: a's rows first, then b's new voxels
    gc, gf = mo.union_add(a_c, a_f, c_c, torch.tensor([[10.0], [20.0]]))
    assert gc.tolist() == [[0, 0, 0, 0], [0, 2, 0, 0], [0, 4, 0, 0]] and gf[:, 0].tolist() == [1.0, 12.0, 20.0]
    uc, uf = gc, torch.tensor([[31.0], [12.0], [20.0]])
    # interpolation: parents at stride 2, query halfway along x between two parents
    q = np.array([[0, 1, 0, 0], [0, 0, 0, 0], [0, 4, 1, 0]], np.float32)
    v = mo.features_at_coordinates(uc, uf, 2, q)
    assert np.allclose(v[:, 0].numpy(), [0.5 * 31 + 0.5 * 12, 31.0, 0.5 * 20.0])
    pc, pf = mo.prune(uc, uf, np.array([True, False, True]))
    assert pc.tolist() == [[0, 0, 0, 0], [0, 4, 0, 0]] and pf[:, 0].tolist() == [31.0, 20.0]


def test_instance_norm_matches_torch():
    x = torch.randn(50, 6, dtype=torch.float64)
    b = np.array([0] * 20 + [1] * 30)
    w = torch.randn(1, 6, dtype=torch.float64); bi = torch.randn(1, 6, dtype=torch.float64)
    y = mo.instance_norm(x, b, w, bi)
    for s, sl in ((0, slice(0, 20)), (1, slice(20, 50))):
        ref = F.instance_norm(x[sl].t()[None], eps=1e-8)[0].t() * w + bi
        assert torch.allclose(y[sl], ref, atol=1e-10)
