"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/fcaf3d_hip.h declares."""
import ctypes
import os

from fcaf3d_amd import _lib as L


def test_library_exports_every_declared_symbol():
    from fcaf3d_amd.build import build
    path = build(verbose=False)
    assert os.path.exists(path)
    protos = L.parse_header()
    assert len(protos) >= 20
    lib = ctypes.CDLL(path)
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, missing


def test_size_queries_run_without_gpu():
    assert L.query('fc_hash_unique_ws_bytes', 1000) > 0
    assert L.query('fc_conv_wgrad_ws_bytes', 100000, 27, 64, 64, 0) >= 27 * 64 * 64 * 4
    assert L.query('fc_col_stats_ws_bytes', 1000, 64, 1) > 0
