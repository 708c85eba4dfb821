"""The C-ABI shared library: builds/loads without a GPU and exports every symbol include/dmt_hip.h declares."""
import ctypes as C
import os
import re

from cikm2020_dmt_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "dmt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dmt_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.load()
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/dmt_hip.h is not exported" % n
    assert lib.dmt_build_arch() == b"gfx950"
    assert set(L.EXPORTED_SYMBOLS) == set(names)


def test_ctypes_struct_layouts_match_the_header():
    lib = L.load()
    lib.dmt_struct_size.restype = C.c_int
    lib.dmt_struct_size.argtypes = [C.c_int]
    for i, st in enumerate([L.GatherFeature, L.GatherDesc, L.EmbGradDesc, L.GemmDesc, L.AttnDesc, L.AttnBwdDesc, L.TableMap, L.CastJob,
                            L.ChainDesc, L.WgradDesc, L.MhsaDesc, L.MmoeDesc, L.HeadsDesc, L.Q1memDesc, L.MhsaBwdDesc]):
        assert C.sizeof(st) == lib.dmt_struct_size(i), st.__name__
    assert C.sizeof(L.EmbGradDesc) < 4096 and C.sizeof(L.GatherDesc) < 4096   # passed by value as kernel arguments


def test_argument_validation_needs_no_gpu():
    """Descriptor checks run on the host before any launch: bad arguments return an error code + message."""
    lib = L.load()
    d = L.GemmDesc()
    assert lib.dmt_gemm(C.byref(d), None) == -1
    assert b"dmt_gemm" in lib.dmt_last_error()
    assert lib.dmt_gather_fwd(None, None) == -1
    assert lib.dmt_ln_bwd_partials(10 ** 7) == 1024


def test_integration_md_gemm_binding_matches_the_header():
    """The ctypes stub INTEGRATION.md shows a maintainer is executed as written: its GemmDesc must have the header's fields, in
    order, and sizeof(dmt_gemm_desc) (a stale snippet mis-lays every field behind the missing one)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class GemmDesc\(C\.Structure\):.*?\n(\s+_fields_ = \[.*?\)\])", text, flags=re.S)
    assert m, "GemmDesc snippet not found in INTEGRATION.md"
    ns = {"C": C}
    exec("class GemmDesc(C.Structure):\n" + m.group(1), ns)
    doc = ns["GemmDesc"]
    lib = L.load()
    lib.dmt_struct_size.restype = C.c_int
    lib.dmt_struct_size.argtypes = [C.c_int]
    assert C.sizeof(doc) == lib.dmt_struct_size(3)
    assert [f[0] for f in doc._fields_] == [f[0] for f in L.GemmDesc._fields_]
    # ... and the header's own field list
    hdr = open(os.path.join(ROOT, "include", "dmt_hip.h")).read()
    body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} dmt_gemm_desc;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.sub(r"[\*\s]", " ", part).split()[-1])
    assert names == [f[0] for f in doc._fields_]
