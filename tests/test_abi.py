"""The C ABI: include/codeps_hip.h <-> libcodeps_hip.so exports <-> ctypes table (no GPU needed)."""
import ctypes
import os
import re

from codeps_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(header="codeps_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|long long|const char\*)\s+(codeps_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_header_matches_binding_and_exports():
    funcs = _header_functions()
    assert len(funcs) >= 20
    handle = _lib.lib()
    assert set(funcs) == set(_lib.SIGNATURES), set(funcs) ^ set(_lib.SIGNATURES)
    for name, nargs in funcs.items():
        assert hasattr(handle, name), f"{name} not exported by libcodeps_hip.so"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args"


def test_diag_header_matches_binding_and_exports():
    """include/codeps_diag.h <-> libcodeps_diag.so: the micro-benchmarks and the rejected prototype live OUTSIDE the product ABI."""
    funcs = _header_functions("codeps_diag.h")
    handle = _lib.diag_lib()
    assert set(funcs) == set(_lib.DIAG_SIGNATURES), set(funcs) ^ set(_lib.DIAG_SIGNATURES)
    for name, nargs in funcs.items():
        assert hasattr(handle, name), f"{name} not exported by libcodeps_diag.so"
        assert len(_lib.DIAG_SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args"
    assert handle.codeps_diag_version() == 1
    product = _lib.lib()
    for name in ("codeps_diag_mfma_rate", "codeps_diag_feed_rate", "codeps_split_pack_rows", "codeps_conv1x1_split_fwd"):
        assert not hasattr(product, name), f"{name} must not be exported by the product library"


def test_version_and_error_string():
    handle = _lib.lib()
    assert handle.codeps_version() == _lib.ABI_VERSION
    # argument validation happens on the host, before any device work -> testable without a GPU
    d = _lib.PhotometricDesc(0, 0, 0, 5, 0.85, 0)
    assert handle.codeps_photometric_workspace_bytes(ctypes.byref(d)) == 0
    assert b"bad shape" in handle.codeps_last_error()
    d = _lib.PhotometricDesc(1, 16, 16, 5, 0.85, 0)          # coarsest scale 1x1 < 2x2
    assert handle.codeps_photometric_workspace_bytes(ctypes.byref(d)) == 0
    assert b"smaller than 2x2" in handle.codeps_last_error()
    c = _lib.ConvDesc(1, 8, 8, 3, 0, 0, 16, 8, 8, 3, 3, 1, 1, 0, 0, 1, 1)
    assert handle.codeps_conv2d_wgrad_workspace_bytes(ctypes.byref(c)) == 0
    assert b"multiples of 16" in handle.codeps_last_error()


def test_structs_match_c_layout():
    assert ctypes.sizeof(_lib.PhotometricDesc) == 32 and _lib.PhotometricDesc.noise_seed.offset == 24
    assert ctypes.sizeof(_lib.ConvDesc) == 19 * 4 and _lib.ConvDesc.stride_w.offset == 15 * 4 and _lib.ConvDesc.row_pitch.offset == 18 * 4
    # codeps_conv_epilogue: 2 pointers-with-ints up front, 5 pointers, 2 + 8 ints, 2 + 16 ints
    assert ctypes.sizeof(_lib.ConvEpilogue) == 184 and _lib.ConvEpilogue.stat_part.offset == 16 and _lib.ConvEpilogue.bn_x.offset == 32
    assert _lib.ConvEpilogue.bn_group_b0.offset == 80 and _lib.ConvEpilogue.applied.offset == 112 and _lib.ConvEpilogue.seg_tiles.offset == 168
