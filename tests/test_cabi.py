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


def test_abi_version_matches_the_header_and_the_host():
    import re
    from fcaf3d_amd._lib import HEADER
    v = int(re.search(r'#define FC_ABI_VERSION (\d+)', open(HEADER).read()).group(1))
    assert L.lib().fc_abi_version() == v == L.ABI_VERSION


def test_size_queries_run_without_gpu():
    assert L.query('fc_hash_unique_ws_bytes', 1000) > 0
    assert L.query('fc_conv_wgrad_ws_bytes', 100000, 27, 64, 64, 0) >= 27 * 64 * 64 * 4
    assert L.query('fc_col_stats_ws_bytes', 1000, 64, 1) > 0


def test_every_prototype_cites_its_reference_interface():
    """include/fcaf3d_hip.h: the comment above each declaration names the reference file:line it replaces (or says that
    there is no reference counterpart)"""
    import re
    from fcaf3d_amd._lib import HEADER
    txt = open(HEADER).read()
    last, missing = '', []
    for block in re.split(r'(/\*.*?\*/)', txt, flags=re.S):
        if block.startswith('/*'):
            last = block
            continue
        for m in re.finditer(r'\b(?:int64_t|int)\s+(fc_\w+)\s*\(', block):
            if not re.search(r'\.(py|cu|cpp|cuh)\s*:\s*\d+|No reference counterpart', last):
                missing.append(m.group(1))
    assert not missing, missing


def test_statistics_table_sizes_fit_the_executors_arena_bound():
    """fc_conv_stats_blocks (pure host function) against the bound fcaf3d_amd/executor.py allocates a statistics table with:
    rows * (C / 8) * 4 + 8 * C + 256 bytes must hold blocks * 2 * C floats for every route and size (r5)."""
    X6 = (1 << 24) | (1 << 26)
    for C in (64, 128, 256, 512):
        for n in (1, 15, 16, 17, 100, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 16384, 50000, 437248):
            for K, pairs in ((27, 0), (27, 1), (1, 0)):
                nb = L.query('fc_conv_stats_blocks', n, K, C, C, X6, pairs)
                assert nb > 0, (n, K, C, pairs)
                assert nb * 2 * C * 4 <= n * (C // 8) * 4 + 8 * C + 256, (n, K, C, pairs, nb)
                if n <= 4096:
                    assert nb <= 64, (n, nb)      # up to 4 096 rows: at most 64 row blocks -> the BatchNorm behind it is ONE launch
    assert L.query('fc_conv_stats_blocks', 1000, 27, 64, 64, 0, 0) == 0                # fp32 route: no statistics epilogue
    assert L.query('fc_conv_stats_blocks', 1000, 27, 3, 64, X6, 0) == 0                # stem shape: not an MFMA launch
