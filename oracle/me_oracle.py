"""ORACLE — test infrastructure only.  Never imported by the product path.

CPU restatement (numpy for the integer/coordinate work, torch-CPU fp32 for the
feature arithmetic so that autograd supplies the backward) of the
MinkowskiEngine v0.5.4 operators that FCAF3D executes.  MinkowskiEngine is a
third-party dependency pinned by the reference at docker/Dockerfile:27-32 and is
NOT vendored under /root/reference, so every function here follows the
*published* operator semantics and cites the reference
call site it stands in for.

Parity status: **parity unpinned** with respect to MinkowskiEngine itself (the
reference holds no test / golden vector at this boundary).
The restatement is pinned instead against an independent dense-grid oracle
(torch.nn.functional.conv3d / conv_transpose3d / max_pool3d + autograd) in
tests/test_oracle_dense.py.

Row-order rule (shared with the HIP path): a coordinate set keeps
its rows in order of FIRST OCCURRENCE in the producing sequence (ME CPU rule,
Appendix A.2).
"""
import numpy as np
import torch

_B = 1 << 15  # bias making each spatial axis non-negative in 16 bits


def pack_keys(coords):
    """(N,4) int [b,x,y,z] -> (N,) int64, lexicographic in (b,x,y,z)."""
    c = np.asarray(coords).astype(np.int64)
    return (c[:, 0] << 48) | ((c[:, 1] + _B) << 32) | ((c[:, 2] + _B) << 16) | (c[:, 3] + _B)


def batch_sparse_collate(points_xyz, feats):
    """ME.utils.batch_sparse_collate (called at single_stage_sparse.py:34-36).

    points_xyz: list of (n,3) float arrays ALREADY divided by voxel_size.
    Returns coords (ΣN,4) int32 [b, floor(x), floor(y), floor(z)] and feats (ΣN,C).
    """
    cs, fs = [], []
    for b, (p, f) in enumerate(zip(points_xyz, feats)):
        q = np.floor(np.asarray(p, np.float32)).astype(np.int32)
        cs.append(np.concatenate([np.full((len(q), 1), b, np.int32), q], 1))
        fs.append(np.asarray(f, np.float32))
    return np.concatenate(cs), np.concatenate(fs)


def unique_first(coords):
    """Rows of first occurrence, in order of first occurrence (Appendix A.2).

    Returns (unique_coords, first_row_index (ascending), inverse (row -> unique row)).
    """
    keys = pack_keys(coords)
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')          # sorted-unique slot -> rank by first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    first_sorted = first[order]
    return coords[first_sorted], first_sorted, rank[inv]


def sparse_tensor(coords, feats):
    """ME.SparseTensor(coordinates, features) — single_stage_sparse.py:37.
    First occurrence wins (RANDOM_SUBSAMPLE on CPU == first inserted)."""
    uc, first, _ = unique_first(coords)
    return uc, feats[first]


def stride_coords(coords, tensor_stride, s):
    """Output coordinate set of a stride-s op on a stride-T tensor (Appendix A.3)."""
    ts = tensor_stride * s
    c = coords.copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], ts) * ts
    uc, _, _ = unique_first(c)
    return uc


def kernel_offsets(kernel_size, tensor_stride):
    """Offsets (K,3) in x-fastest order; centred for odd k, {0,T} for even k (A.3)."""
    if kernel_size % 2 == 1:
        r = np.arange(kernel_size) - kernel_size // 2
    else:
        r = np.arange(kernel_size)
    r = r * tensor_stride
    dz, dy, dx = np.meshgrid(r, r, r, indexing='ij')
    return np.stack([dx.ravel(), dy.ravel(), dz.ravel()], 1).astype(np.int32)


def kernel_map(in_coords, out_coords, offsets):
    """nbr (K, N_out) int32: row of the input voxel at out_coord + offset_k, or -1."""
    ikeys = pack_keys(in_coords)
    order = np.argsort(ikeys, kind='stable')
    skeys = ikeys[order]
    K = len(offsets)
    nbr = np.full((K, len(out_coords)), -1, np.int32)
    if len(in_coords) == 0 or len(out_coords) == 0:
        return nbr
    for k in range(K):
        q = out_coords.copy()
        q[:, 1:] = q[:, 1:] + offsets[k][None, :]
        qk = pack_keys(q)
        pos = np.searchsorted(skeys, qk)
        pos_c = np.minimum(pos, len(skeys) - 1)
        hit = skeys[pos_c] == qk
        nbr[k, hit] = order[pos_c[hit]]
    return nbr


def conv(feats, weight, nbr):
    """MinkowskiConvolution forward, Appendix A.3:  out[o] = Σ_k in[nbr[k,o]] @ W[k].
    feats (N_in,Cin) torch, weight (K,Cin,Cout) torch, nbr (K,N_out) numpy."""
    K, n_out = nbr.shape
    out = feats.new_zeros((n_out, weight.shape[2]))
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        if len(o) == 0:
            continue
        i = torch.from_numpy(nbr[k, o].astype(np.int64))
        out = out.index_add(0, torch.from_numpy(o), feats.index_select(0, i) @ weight[k])
    return out


def gen_conv_transpose_coords(coords, tensor_stride):
    """MinkowskiGenerativeConvolutionTranspose(k2,s2) output coords (A.4):
    child row 8*i + k at c_i + offset_k, offsets in units of T/2."""
    half = tensor_stride // 2
    offs = kernel_offsets(2, half)
    c = np.repeat(coords, 8, axis=0)
    c[:, 1:] += np.tile(offs, (len(coords), 1))
    return c


def gen_conv_transpose(feats, weight):
    """out[8*i + k] = in[i] @ W[k]; weight (8,Cin,Cout)."""
    out = torch.einsum('nc,kcd->nkd', feats, weight)
    return out.reshape(-1, weight.shape[2])


def max_pool(feats, nbr):
    """MinkowskiMaxPooling(k2,s2) (A.5): channel-wise max over present children."""
    K, n_out = nbr.shape
    neg = torch.full((1, feats.shape[1]), -float('inf'), dtype=feats.dtype)
    padded = torch.cat([feats, neg])
    idx = torch.from_numpy(np.where(nbr >= 0, nbr, len(feats)).astype(np.int64))
    g = padded[idx.reshape(-1)].reshape(K, n_out, -1)
    return g.max(0).values


def instance_norm(feats, batch_idx, weight, bias, eps=1e-8):
    """MinkowskiInstanceNorm (A.6): per scene, per channel, biased variance."""
    out = torch.empty_like(feats)
    b = torch.from_numpy(np.asarray(batch_idx).astype(np.int64))
    for s in range(int(b.max()) + 1 if len(b) else 0):
        m = b == s
        x = feats[m]
        mu = x.mean(0, keepdim=True)
        var = ((x - mu) ** 2).mean(0, keepdim=True)
        out[m] = (x - mu) / torch.sqrt(var + eps)
    return out * weight + bias


def batch_norm(feats, weight, bias, eps=1e-5):
    """MinkowskiBatchNorm == BatchNorm1d over all rows, training mode (A.7)."""
    return torch.nn.functional.batch_norm(feats, None, None, weight, bias, True, 0.1, eps)


def union_add(coords_a, feats_a, coords_b, feats_b):
    """SparseTensor a + b with different maps (A.8): rows of a first, then rows
    of b absent from a; features zero-filled then added.
    Row-order rule shared with the HIP path: when every voxel of a already lies in b (the backbone level inside the
    generated children set) the union is b's coordinate set IN b's ORDER (the map, and its kernel maps, are reused)."""
    ka, kb = pack_keys(coords_a), pack_keys(coords_b)
    if len(ka) and len(kb) and np.isin(ka, kb).all() and len(kb) > len(ka):
        order_b = np.argsort(kb, kind='stable')
        pos = np.searchsorted(kb[order_b], ka)
        row_a = order_b[pos]
        out = feats_b.new_zeros((len(kb), feats_b.shape[1]))
        out = out.index_add(0, torch.arange(len(kb)), feats_b)
        out = out.index_add(0, torch.from_numpy(row_a), feats_a)
        return coords_b, out
    order = np.argsort(ka, kind='stable')
    ska = ka[order]
    pos = np.minimum(np.searchsorted(ska, kb), max(len(ska) - 1, 0))
    hit = (ska[pos] == kb) if len(ska) else np.zeros(len(kb), bool)
    row_b = np.empty(len(kb), np.int64)
    row_b[hit] = order[pos[hit]]
    n_new = int((~hit).sum())
    row_b[~hit] = len(ka) + np.arange(n_new)
    coords = np.concatenate([coords_a, coords_b[~hit]])
    out = feats_a.new_zeros((len(coords), feats_a.shape[1]))
    out = out.index_add(0, torch.arange(len(ka)), feats_a)
    out = out.index_add(0, torch.from_numpy(row_b), feats_b)
    return coords, out


def features_at_coordinates(coords, feats, tensor_stride, query):
    """SparseTensor.features_at_coordinates (A.8): trilinear weights over the 2^3
    lattice corners (step = tensor_stride) of the cell containing each query."""
    S = tensor_stride
    q = np.asarray(query, np.float32)
    base = np.floor(q[:, 1:] / S).astype(np.int64) * S
    ikeys = pack_keys(coords)
    order = np.argsort(ikeys, kind='stable')
    skeys = ikeys[order]
    out = torch.zeros((len(q), feats.shape[1]), dtype=feats.dtype)
    for k in range(8):
        d = np.array([(k >> 0) & 1, (k >> 1) & 1, (k >> 2) & 1]) * S
        corner = base + d
        w = np.prod(1.0 - np.abs(q[:, 1:] - corner) / S, axis=1).astype(np.float32)
        cc = np.concatenate([q[:, :1].astype(np.int64), corner], 1)
        ck = pack_keys(cc)
        pos = np.minimum(np.searchsorted(skeys, ck), len(skeys) - 1)
        hit = skeys[pos] == ck
        rows = order[pos[hit]]
        contrib = feats[torch.from_numpy(rows)] * torch.from_numpy(w[hit])[:, None]
        out = out.index_add(0, torch.from_numpy(np.nonzero(hit)[0]), contrib)
    return out


def prune(coords, feats, mask):
    """MinkowskiPruning (A.8): keep rows where mask, order preserved."""
    m = np.asarray(mask, bool)
    return coords[m], feats[torch.from_numpy(np.nonzero(m)[0])]


def elu(x):
    return torch.nn.functional.elu(x)


def relu(x):
    return torch.relu(x)
