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
    assert lib.v2x_abi_version() == _lib.ABI_VERSION == 18


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


# ---- every prototype of the header against the ctypes signature table, argument by argument -------------------------
_INT = {"int": 4, "int32_t": 4, "uint32_t": 4, "unsigned": 4, "unsigned int": 4, "int64_t": 8, "long long": 8, "unsigned long long": 8, "size_t": 8,
        "uint64_t": 8}
_STRUCTS = {"v2x_conv_desc": "ConvDesc", "v2x_pack_spec": "PackSpec", "v2x_pack_job": "PackJob", "v2x_adam_tensors": "AdamTensors"}


def _c_kind(ctype_text):
    """C parameter / return type -> (kind, size, pointee): kind in {"int", "float", "ptr", "void"}; pointee = struct name, scalar type or None."""
    t = " ".join(ctype_text.replace("const", " ").split())
    if t.endswith("*") or t == "v2x_stream_t":
        base = t.rstrip("* ").strip()
        return ("ptr", 8, base if t != "v2x_stream_t" else "void")
    if t == "void":
        return ("void", 0, None)
    if t in ("float",):
        return ("float", 4, None)
    if t in ("double",):
        return ("float", 8, None)
    assert t in _INT, "unknown C type %r in the header" % ctype_text
    return ("int", _INT[t], None)


def _ctypes_kind(ct):
    import ctypes as C
    from v2x_sim_amd import _lib
    if ct is None:
        return ("void", 0, None)
    if ct in (C.c_void_p, C.c_char_p):
        return ("ptr", 8, "char" if ct is C.c_char_p else None)
    if isinstance(ct, type) and issubclass(ct, C._Pointer):
        tgt = ct._type_
        for cname, pyname in _STRUCTS.items():
            if tgt is getattr(_lib, pyname):
                return ("ptr", 8, cname)
        return ("ptr", 8, {C.c_double: "double", C.c_int32: "int32_t", C.c_int: "int", C.c_int64: "int64_t", C.c_float: "float"}.get(tgt, "?"))
    if ct in (C.c_float,):
        return ("float", 4, None)
    if ct in (C.c_double,):
        return ("float", 8, None)
    return ("int", C.sizeof(ct), None)


def _prototypes():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    src = re.sub(r"typedef struct.*?\}\s*\w+;", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*((?:const\s+)?(?:unsigned\s+)?[a-z_0-9]+(?:\s+long)?\s*\**)\s*(v2x_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        plist = []
        if params and params != "void":
            for prm in params.split(","):
                prm = prm.strip()
                mm = re.match(r"^(.*?[\s\*])([A-Za-z_][A-Za-z_0-9]*)$", prm)
                assert mm, (name, prm)
                plist.append(mm.group(1).strip())
        out[name] = (ret.strip(), plist)
    return out


def test_every_ctypes_signature_matches_its_header_prototype():
    """The per-function argtypes / restype of v2x_sim_amd/_lib.py::SIGNATURES are hand-written; a slip (an int where the header takes a pointer, a
    missing argument, a float passed as an int) would corrupt a call silently.  Every prototype of include/v2x_amd.h is parsed and compared
    with its table entry: argument count, and per argument the machine class ctypes will marshal (integer of N bytes / float / pointer);
    pointers to the ABI structs must be typed as that struct (or left opaque for DEVICE arrays of them), typed scalar pointers must name the
    header's scalar."""
    from v2x_sim_amd import _lib
    protos = _prototypes()
    assert set(protos) == set(_lib.SIGNATURES), sorted(set(protos) ^ set(_lib.SIGNATURES))
    checked = 0
    for name, (ret, params) in sorted(protos.items()):
        res, args = _lib.SIGNATURES[name]
        assert len(args) == len(params), "%s: header has %d parameters, ctypes table %d" % (name, len(params), len(args))
        rk, ck = _c_kind(ret), _ctypes_kind(res)
        assert rk[:2] == ck[:2], "%s: return type %r vs %r" % (name, ret, res)
        for i, (ptxt, ct) in enumerate(zip(params, args)):
            hk, pk = _c_kind(ptxt), _ctypes_kind(ct)
            assert hk[:2] == pk[:2], "%s: parameter %d is `%s` in the header, %r in the ctypes table" % (name, i, ptxt, ct)
            if hk[0] == "ptr" and hk[2] in _STRUCTS:
                assert pk[2] in (hk[2], None), "%s: parameter %d points to %s, the table types it as %r" % (name, i, hk[2], pk[2])
            elif hk[0] == "ptr" and pk[2] not in (None, "char"):
                assert _c_kind(pk[2])[:2] == _c_kind(hk[2])[:2], "%s: parameter %d is `%s`, typed pointer to %s in the table" % (name, i, ptxt, pk[2])
            checked += 1
    assert checked > 400, checked


def test_pack_structs_mirror_the_header():
    """v2x_pack_spec / v2x_pack_job field order and sizes against their ctypes mirrors."""
    import ctypes as C
    from v2x_sim_amd import _lib
    src = open(HEADER).read()
    for cname, pyname in (("v2x_pack_spec", "PackSpec"), ("v2x_pack_job", "PackJob")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            tokens = decl.replace("const ", "").split(" ", 1)
            base, rest = tokens[0], tokens[1]
            for n in rest.split(","):
                n = n.strip()
                fields.append((n.replace("*", "").strip(), ("ptr", 8) if "*" in n else _c_kind(base)[:2]))
        mirror = getattr(_lib, pyname)._fields_
        assert [f[0] for f in fields] == [f[0] for f in mirror], (cname, fields, mirror)
        for (n, k), (_, ct) in zip(fields, mirror):
            assert k == _ctypes_kind(ct)[:2], (cname, n, k, ct)
        assert C.sizeof(getattr(_lib, pyname)) % 4 == 0


def test_adam_tensor_table_mirrors_the_header():
    """v2x_adam_tensors (six arrays of V2X_ADAM_MAX_TENSORS entries, passed by host pointer): names, order, element sizes and length against _lib.AdamTensors;
    the table must fit the kernel-argument segment (4 KiB) with the launch's scalars."""
    import ctypes as C
    from v2x_sim_amd import _lib
    src = open(HEADER).read()
    n = int(re.search(r"#define V2X_ADAM_MAX_TENSORS (\d+)", src).group(1))
    assert n == _lib.ADAM_MAX_TENSORS
    body = re.search(r"typedef struct v2x_adam_tensors \{(.*?)\} v2x_adam_tensors;", src, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"^(.*?)(\w+)\[V2X_ADAM_MAX_TENSORS\]$", decl)
        assert m, decl
        fields.append((m.group(2), 8))          # pointers and long long: 8 bytes each
    mirror = _lib.AdamTensors._fields_
    assert [f[0] for f in fields] == [f[0] for f in mirror]
    for (_, size), (_, ct) in zip(fields, mirror):
        assert ct._length_ == n and C.sizeof(ct._type_) == size
    assert C.sizeof(_lib.AdamTensors) == 6 * 8 * n
    # device-side table: 5 pointers + int numel + int start (+1) per tensor, + 56 bytes of scalars
    assert n * (5 * 8 + 4) + (n + 1) * 4 + 56 <= 4096
