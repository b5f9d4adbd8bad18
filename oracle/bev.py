"""ORACLE — test infrastructure only: ctypes front-end of oracle/libfcaf3d_oracle.so."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, 'libfcaf3d_oracle.so')
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(['make', '-C', _HERE, 'c'])
        _lib = ctypes.CDLL(path)
        _lib.oracle_nms.restype = ctypes.c_int
    return _lib


def iou_matrix(a, b, rotated=True):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.zeros((len(a), len(b)), np.float32)
    lib().oracle_iou_matrix(a.ctypes.data_as(ctypes.c_void_p), len(a), b.ctypes.data_as(ctypes.c_void_p), len(b),
                            int(rotated), out.ctypes.data_as(ctypes.c_void_p))
    return out


def nms(boxes, scores, thresh, rotated=True):
    """pcdet nms_gpu / nms_normal_gpu semantics (pcdet_nms_utils.py:86-117): indices into the input."""
    order = np.argsort(-np.asarray(scores), kind='stable')
    sb = np.ascontiguousarray(np.asarray(boxes, np.float32)[order])
    keep = np.zeros(len(sb), np.int64)
    k = lib().oracle_nms(sb.ctypes.data_as(ctypes.c_void_p), len(sb), ctypes.c_float(thresh), int(rotated),
                         keep.ctypes.data_as(ctypes.c_void_p))
    return order[keep[:k]]
