"""ORACLE — test infrastructure only.  The arithmetic of the split-bf16 convolution kernels (fcaf3d_amd/csrc/conv_x6.h,
wgrad_x6.h) restated in numpy: an fp32 value as three bf16 pieces, an fp32 product as the six bf16 x bf16 products the
kernels keep, accumulated in fp32 in blocks of 16 products (one v_mfma_f32_32x32x16_bf16 per piece pair).
Pins the properties the kernels rely on (tests/test_oracle_golden.py):
  * x == x1 + x2 + x3 exactly for every finite fp32 (24 = 8 + 8 + 8 significand bits, bf16 has the fp32 exponent range);
  * every piece product is exact in fp32 (8 x 8 = 16 bits);
  * r4, the kernels' split: pieces by ROUND-TO-NEAREST-EVEN (v_cvt_pk_bf16_f32): |x2| <= 2^-8 |x|, |x3| <= 2^-17 |x|, residuals
    of either sign, so the three dropped products (x2 y3, x3 y2, x3 y3) sum to at most 2^-24 |x y| and are unbiased;
  * r3 (`split3_trunc`, the kernels built with -DFC_X6_TRUNC): pieces by truncation, |x2| < 2^-7 |x|, |x3| < 2^-15 |x|, dropped
    products <= 2^-21 |x y|, all with the sign of x y.
There is no reference counterpart: the reference computes the same convolutions in fp32 on ME's kernels (me_resnet.py:56-62)."""
import numpy as np


def _rn_bf16(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 (finite inputs)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    """fp32 array -> (x1, x2, x3) fp32 arrays, each representable in bf16 (low 16 bits zero), x1 + x2 + x3 == x; pieces
    rounded to nearest even (the kernels' split since r4)"""
    x = np.asarray(x, dtype=np.float32)
    x1 = _rn_bf16(x)
    r = x - x1
    x2 = _rn_bf16(r)
    x3 = _rn_bf16(r - x2)
    return x1, x2, x3


def split3_trunc(x):
    """the truncating split of r3 (-DFC_X6_TRUNC)"""
    x = np.asarray(x, dtype=np.float32)
    mask = np.uint32(0xffff0000)
    x1 = (x.view(np.uint32) & mask).view(np.float32)
    r = x - x1
    x2 = (r.view(np.uint32) & mask).view(np.float32)
    r2 = r - x2
    x3 = (r2.view(np.uint32) & mask).view(np.float32)
    return x1, x2, x3


# (piece of a, piece of b) in the order the kernels accumulate them inside a 16-channel block: smallest products first
TERMS = ((0, 2), (1, 1), (0, 1), (0, 0), (1, 0), (2, 0))


def matmul_x6(a, b, block=16):
    """(M,K) @ (K,N) the way the kernels do it: per block of `block` reduction indices and per kept piece pair ONE fp32-rounded
    update of the accumulator with the exactly summed products (the MFMA's internal sum is modelled as exact)."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    pa, pb = split3(a), split3(b)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k0 in range(0, a.shape[1], block):
        for i, j in TERMS:
            upd = pa[i][:, k0:k0 + block].astype(np.float64) @ pb[j][k0:k0 + block].astype(np.float64)
            acc = (acc.astype(np.float64) + upd).astype(np.float32)
    return acc


# ---- r6: the two-piece fp16 split ("h3", csrc/conv_x6.h MODE 2) --------------------------------------------------------------------
# Pins: the scale rule (integer arithmetic on the exponent field of max |x|), (h, l) holds s x to 2^-23 |s x| (one fp32 ulp; exactly
# for three values in four) whenever both pieces are normal fp16 numbers and to 2^-25 absolute below that, no piece overflows fp16,
# the three kept products are exact in fp32, and what is dropped is at most 2^-21 of a product (2^-25 on average, unbiased).
TERMS_H3 = ((0, 1), (1, 0), (0, 0))          # (piece of a, piece of b), the kernels' order inside a 16-channel block


def h3_scale(amax):
    """the kernels' h3_scale: 2^(14 - floor(log2 amax)) from the exponent FIELD of the fp32 amax (clamped to the normal range), so
    the largest element lands in [2^14, 2^15)"""
    e = (np.float32(amax).view(np.uint32) >> np.uint32(23)) & np.uint32(0xff)
    se = int(np.clip(268 - int(e), 1, 254))
    return np.uint32(se << 23).view(np.float32)


def h3_unscale(sa, sb):
    e = 381 - (int(np.float32(sa).view(np.uint32)) >> 23) - (int(np.float32(sb).view(np.uint32)) >> 23)
    return np.uint32(int(np.clip(e, 1, 254)) << 23).view(np.float32)


def amax_finite(x):
    """fc_amax: the largest FINITE |x| (0 for an empty / all-non-finite tensor)"""
    a = np.abs(np.asarray(x, np.float32))
    a = a[np.isfinite(a)]
    return np.float32(a.max()) if a.size else np.float32(0)


def split2_h(x, s):
    """fp32 array, power-of-two scale -> (h, l) float16 arrays: h = RN11(s x), l = RN11(s x - h)  (numpy rounds to nearest even, as
    v_cvt_pk_f16_f32 does; the residual s x - h is exact in fp32)"""
    xs = np.asarray(x, np.float32) * np.float32(s)
    with np.errstate(over='ignore'):
        h = xs.astype(np.float16)
        r = xs - h.astype(np.float32)
        lo = r.astype(np.float16)
    return h, lo


def matmul_h3(a, b, block=16):
    """(M,K) @ (K,N) the way the h3 kernels do it (matmul_x6 with two fp16 pieces per operand, tensor-wide scales, three products)"""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    sa, sb = h3_scale(amax_finite(a)), h3_scale(amax_finite(b))
    pa, pb = split2_h(a, sa), split2_h(b, sb)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k0 in range(0, a.shape[1], block):
        for i, j in TERMS_H3:
            upd = pa[i][:, k0:k0 + block].astype(np.float64) @ pb[j][k0:k0 + block].astype(np.float64)
            acc = (acc.astype(np.float64) + upd).astype(np.float32)
    return acc * h3_unscale(sa, sb)
