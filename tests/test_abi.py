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
            assert handle.avsd_abi_version() == 10
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


def test_plan_bundle_file_format_and_errors(tmp_path):
    """The launch-plan loader (include/avsd.h "launch plans") without a device: a hand-written bundle parses, reports its
    buffers and plans, refuses to run with an unbound buffer, and malformed / foreign files are rejected with a message."""
    import struct

    from asva_amd import _lib, plan

    h = _lib.lib()

    def s(b):
        return struct.pack("<I", len(b)) + b

    body = plan.MAGIC + struct.pack("<I", h.avsd_abi_version()) + b"bf16".ljust(8, b"\0")
    body += struct.pack("<I", 2) + struct.pack("<q", 4096) + struct.pack("<q", 1 << 20)                      # two buffers
    body += struct.pack("<I", 1) + s(b"weights") + struct.pack("<iqqI", 0, 256, 1024, plan.CONST)             # one region
    call = s(b"avsd_copy") + struct.pack("<I", 5) + b"p" + struct.pack("<iq", 0, 256) + b"p" + struct.pack("<iq", 1, 512)
    call += b"i" + struct.pack("<q", 1024) + b"i" + struct.pack("<q", 2) + b"S"
    body += struct.pack("<I", 1) + s(b"forward") + struct.pack("<I", 1) + call
    path = tmp_path / "t.plan"
    path.write_bytes(body)
    b = plan.Bundle(str(path))
    assert b.buffer_sizes() == [4096, 1 << 20] and b.regions() == {"weights": (0, 256, 1024, plan.CONST)}
    assert h.avsd_plan_bundle_find_region(b._h, b"weights") == 0 and h.avsd_plan_bundle_find_region(b._h, b"x") == -1
    assert h.avsd_plan_bundle_num_plans(b._h) == 1 and h.avsd_plan_bundle_plan_name(b._h, 0) == b"forward"
    assert h.avsd_plan_num_calls(b._h, 0) == 1 and h.avsd_plan_bundle_find_plan(b._h, b"decode") == -1
    assert h.avsd_plan_run(b._h, 0, None) == -1 and b"buffer 0" in h.avsd_last_error() and b"not bound" in h.avsd_last_error()
    b.close()
    for bad, msg in ((b"garbage" * 8, b"not a plan bundle"), (body[:-3], b"truncated"),
                     (body.replace(b"bf16", b"fp16", 1), b"other storage precision"),
                     (body.replace(b"avsd_copy", b"avsd_nope"), b"does not record"),
                     (body.replace(struct.pack("<iqqI", 0, 256, 1024, plan.CONST), struct.pack("<iqqI", 0, 4000, 1024, plan.CONST)), b"outside its buffer")):
        path.write_bytes(bad)
        out = ctypes.c_void_p()
        assert h.avsd_plan_bundle_load(str(path).encode(), ctypes.byref(out)) == -1 and msg in h.avsd_last_error(), h.avsd_last_error()


def test_every_recordable_entry_point_is_replayable():
    """asva_amd/plan.py records every launching entry point it sees; csrc/plan.hip replays only those in its table — they must agree"""
    import re
    from asva_amd import _lib, plan

    src = open(os.path.join(os.path.dirname(__file__), "..", "asva_amd", "csrc", "plan.hip")).read()
    table = set(re.findall(r"AVSD_PLAN_ENTRY(?:_DESC)?\((avsd_\w+)", src))
    recordable = {n for n in _lib.SIGNATURES
                  if n not in plan._NOT_RECORDED and not n.startswith(("avsd_plan_", "avsd_unet_")) and n != "avsd_vae_decode"}
    # queries that never launch and so never reach a plan
    recordable -= {n for n in recordable if n.startswith("avsd_sizeof_") or n.endswith("_supported")}
    missing = {n for n in recordable if n not in table}
    assert not missing - {"avsd_gemm_f32"}, sorted(missing)


def test_graft_entry_build_runs():
    """the driver's "does it build" check, as the driver calls it (the libraries are built incrementally: seconds when nothing changed)"""
    import importlib
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ge = importlib.import_module("__graft_entry__")
    ge.build()
