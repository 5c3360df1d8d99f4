"""CPU tests of the drop-in boundary: libv2x_amd.so loads and exports every symbol that
include/v2x_amd.h declares, the ctypes mirror matches the header, and argument validation
returns errno-style codes without touching a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "v2x_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|size_t|long long|char\s*\*|const char\s*\*)\s*\**\s*(v2x_[a-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for must in ("v2x_voxelize_bits", "v2x_conv2d", "v2x_warp_fuse", "v2x_attn_handshake",
                 "v2x_seg_argmax_confusion", "v2x_bits_to_indices", "v2x_conv_tile_rows"):
        assert must in names
    assert len(names) >= 12


def test_library_exports_every_declared_symbol():
    from v2x_sim_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    for n in names:
        assert hasattr(lib, n), n
    assert lib.v2x_abi_version() == _lib.ABI_VERSION == 15


def test_conv_desc_mirror_matches_header():
    from v2x_sim_amd._lib import ConvDesc
    src = open(HEADER).read()
    body = re.search(r"typedef struct v2x_conv_desc \{(.*?)\} v2x_conv_desc;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
        for n in names.split(","):
            fields.append(n.replace("*", "").strip())
    assert fields == [f[0] for f in ConvDesc._fields_]
    # 6 pointers + 19 int32 (+ padding) -- layout sanity on LP64
    assert ctypes.sizeof(ConvDesc) % 8 == 0


def test_tile_rows_and_validation_without_gpu():
    from v2x_sim_amd import _lib
    lib = _lib.load()
    assert lib.v2x_conv_tile_rows(32, 0) == 32
    assert lib.v2x_conv_tile_rows(48, 1) == 48
    assert lib.v2x_conv_tile_rows(64, 0) == 64
    assert lib.v2x_conv_tile_rows(256, 0) == 128
    assert lib.v2x_conv_tile_rows(256, 2) == 96
    # argument validation happens before any HIP call
    assert lib.v2x_conv2d(None, None) == -22
    assert b"null descriptor" in lib.v2x_last_error()
    d = _lib.ConvDesc()
    assert lib.v2x_conv2d(ctypes.byref(d), None) == -22
    assert lib.v2x_warp_fuse(None, 5, 1, 32, 32, 256, None, None, 1, None, 0, None, None) == -22
    assert lib.v2x_conv2d_pair(None, None, None) == -22 and b"null descriptor" in lib.v2x_last_error()
    d2 = _lib.ConvDesc()
    assert lib.v2x_conv2d_pair(ctypes.byref(d), ctypes.byref(d2), None) == -22     # not halo-packed 3x3 32 -> 32 layers
    assert b"v2x_conv2d_pair" in lib.v2x_last_error()
    assert lib.v2x_voxelize_bits(None, None, 1, 1, 4, None, None, None, None, None) == -22
    assert lib.v2x_attn_handshake(None, None, None, None, 5, 1, 1024, 32, 0, 0.2, None, None, None) == -22
    with pytest.raises(_lib.V2XLibraryError):
        _lib.check(-22, "probe")
