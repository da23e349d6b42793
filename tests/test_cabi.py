"""CPU test: the C-ABI library loads and exports every symbol include/imt_hip.h declares (no compute calls)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "imt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(imt_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from imagetranslate_amd import _lib
    names = _declared()
    assert len(names) >= 20
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), "libimt_hip.so does not export %s" % n
    assert set(names) == set(_lib.SIGNATURES), (set(names) ^ set(_lib.SIGNATURES))
    assert lib.imt_version() >= 100


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (safe without a GPU)."""
    import ctypes
    from imagetranslate_amd import _lib
    lib = _lib.load()
    a = _lib.GemmArgs()
    a.dtype = 7
    assert lib.imt_gemm(ctypes.byref(a), None) == -1
    assert b"dtype" in lib.imt_last_error()
    assert lib.imt_layernorm_fwd(0, None, None, None, None, None, None, 4, 6, 1e-12, 0.0, 0, None) == -1  # d % 4


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from imagetranslate_amd import hip_ops as O
    from imagetranslate_amd._lib import ImtError
    with pytest.raises(ImtError):
        O.gemm(torch.zeros(8, 8), torch.zeros(8, 8), O.IMT_NT)


def test_struct_layouts_of_binding_and_library_agree():
    """imt_abi_sizeof: every argument structure has the same size in the ctypes binding as in the compiled library
    (checked at load time too; here also the refusal of an unknown name)."""
    import ctypes
    from imagetranslate_amd import _lib as L
    lib = L.load()
    for cname, cls in L.ABI_STRUCTS.items():
        assert lib.imt_abi_sizeof(cname.encode()) == ctypes.sizeof(cls), cname
    assert lib.imt_abi_sizeof(b"no_such_struct") == -1
