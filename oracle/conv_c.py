"""ORACLE — test infrastructure only: ctypes front-end of the C / OpenMP convolution restatement (oracle/conv_oracle.c),
MinkowskiEngine's CPU algorithm (per offset: gather -> GEMM -> scatter-add).  Used by tests/ (checked against
oracle/me_oracle.py) and by bench.py's cpu_baseline leg; never imported by the product."""
import ctypes

import numpy as np

from . import bev

_sigs = False


def _lib():
    global _sigs
    l = bev.lib()
    if not _sigs:
        l.oc_num_threads.restype = ctypes.c_int
        for f in (l.oc_conv_fwd, l.oc_conv_dgrad, l.oc_conv_wgrad):
            f.restype = None
        _sigs = True
    return l


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(_lib().oc_num_threads())


def conv_fwd(feats, weight, nbr):
    feats, weight, nbr = (np.ascontiguousarray(feats, np.float32), np.ascontiguousarray(weight, np.float32),
                          np.ascontiguousarray(nbr, np.int32))
    K, n_out = nbr.shape
    Cin, Cout = weight.shape[1:]
    out = np.empty((n_out, Cout), np.float32)
    _lib().oc_conv_fwd(_p(feats), _p(weight), _p(nbr), ctypes.c_int64(n_out), K, Cin, Cout, _p(out))
    return out


def conv_dgrad(gout, weight, nbr, n_in):
    gout, weight, nbr = (np.ascontiguousarray(gout, np.float32), np.ascontiguousarray(weight, np.float32),
                         np.ascontiguousarray(nbr, np.int32))
    K, n_out = nbr.shape
    Cin, Cout = weight.shape[1:]
    gin = np.empty((n_in, Cin), np.float32)
    _lib().oc_conv_dgrad(_p(gout), _p(weight), _p(nbr), ctypes.c_int64(n_in), ctypes.c_int64(n_out), K, Cin, Cout, _p(gin))
    return gin


def conv_wgrad(feats, gout, nbr, Cin, Cout):
    feats, gout, nbr = (np.ascontiguousarray(feats, np.float32), np.ascontiguousarray(gout, np.float32),
                        np.ascontiguousarray(nbr, np.int32))
    K, n_out = nbr.shape
    gw = np.empty((K, Cin, Cout), np.float32)
    _lib().oc_conv_wgrad(_p(feats), _p(gout), _p(nbr), ctypes.c_int64(n_out), K, Cin, Cout, _p(gw))
    return gw
