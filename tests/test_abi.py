"""The C-ABI library loads without a GPU and exports exactly the entry points include/avsd.h declares
(-m "not gpu": no compute call is made)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "avsd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(avsd_[a-z0-9_]+)\s*\(", src)))


def test_header_binding_and_library_agree():
    from asva_amd import _lib

    declared = _declared()
    assert declared, "no declarations parsed"
    assert sorted(_lib.SIGNATURES) == declared                 # the ctypes table mirrors the header
    from asva_amd import precision

    for prec in ("bf16", "fp16"):                              # both builds of the sources export the same ABI
        precision.set_precision(prec)
        try:
            handle = _lib.lib()                                # raises if the .so is missing or a symbol is absent
            for name in declared:
                assert hasattr(handle, name)
            assert handle.avsd_abi_version() == 3
            assert handle.avsd_precision() == prec.encode()
            assert handle.avsd_sizeof_gemm_desc() == ctypes.sizeof(_lib.GemmDesc)
        finally:
            precision.set_precision("bf16")


def test_argument_errors_are_reported_without_a_device():
    from asva_amd import _lib

    h = _lib.lib()
    assert h.avsd_gemm_bf16(None, None) == -1                   # AVSD_EINVAL, no launch attempted
    assert b"null descriptor" in h.avsd_last_error()
    assert h.avsd_groupnorm_nchunks(2, 12288, 320) >= 1
    name = ctypes.create_string_buffer(64)
    n = ctypes.c_int(0)
    rc = h.avsd_device_info(name, 64, ctypes.byref(n))
    assert rc in (0, -3)                                        # AVSD_ENODEV on the GPU-less build container


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from asva_amd import _lib

    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATHS", {"bf16": str(tmp_path / "libavsd_hip.so"), "fp16": str(tmp_path / "libavsd_hip_f16.so")})
    import pytest

    with pytest.raises(_lib.AvsdError, match="no fallback compute path"):
        _lib.lib()
