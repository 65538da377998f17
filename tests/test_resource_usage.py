"""Build-time guard (-m "not gpu"): register / scratch / occupancy figures of every kernel in the built libraries against the
committed table asva_amd/resource_usage_gfx950.json (tools/resource_usage.py).  Round 4 lost 8-12 % on the VAE decode and cfg 4 to
an allocator change that no same-build A/B could see (profiles/r4_regalloc_ab.txt); round 5 found the most-used convolution tile
keeping its 160 accumulators in scratch memory.  A kernel that starts to spill, grows its scratch segment or loses a wave per SIMD
now fails here, on the CPU box, before any GPU time is spent."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import resource_usage as ru  # noqa: E402


@pytest.fixture(scope="module")
def built():
    if not all(os.path.isfile(p) for p in ru.LIBS.values()):
        from asva_amd.build import build

        build()
    return ru.collect()


def test_no_kernel_regressed_against_the_committed_table(built):
    with open(ru.TABLE) as f:
        table = json.load(f)
    bad, notes = ru.compare(built, table)
    assert not bad, "\n".join(bad)
    # membership: the table describes exactly the kernels that ship (a new instantiation must be looked at once, then committed)
    stale = [n for n in notes if "new kernel" in n or "is gone" in n]
    assert not stale, "\n".join(stale)


def test_hot_path_kernels_keep_their_accumulators_in_registers(built):
    """the kernels a denoising step spends its time in (profiles/r5_kernel_trace_bench.md) use no scratch memory at all, and the
    most-used tiles hold the occupancy their LDS footprint admits"""
    bf = built["bf16"]
    for name, waves in (("conv3r_kernel<256, 160, 4, 1, 3, 4>", 2), ("conv3r_kernel<256, 128, 4, 2, 3, 4>", 2),
                        ("gemm4_kernel<2, 2, 0, false, 2, 0>", 2), ("gemm4_kernel<1, 2, 0, false, 2, 0>", 2), ("gemm4_kernel<1, 2, 1, false, 2, 0>", 2),
                        ("gemm4_kernel<1, 1, 1, false, 2, 0>", 3), ("gemm2_kernel<256, 128, 4, 2, 3, 2, 4, false, 0>", 3),
                        ("attn_kernel<40, 1, false>", 4), ("attn_kernel<80, 1, false>", 3)):
        r = bf[name]
        # (<= 16 B: a couple of address registers parked around the epilogue; the accumulator array itself was 1216 B in round 4)
        assert r["private_segment_fixed_size"] <= 16 and r["vgpr_spill_count"] <= 2, (name, r)
        assert r["waves_per_simd"] >= waves, (name, r)


def test_waves_per_simd_rule():
    # MI355X_MICROARCH.md, register files: 8-register granule, 512 per SIMD lane
    assert [ru.waves_per_simd(v) for v in (57, 64, 65, 96, 128, 129, 168, 169, 256, 257, 512)] == [8, 8, 7, 5, 4, 3, 3, 2, 2, 1, 1]


def test_compare_flags_spills_scratch_and_occupancy():
    ref = {"bf16": {"k": {"vgpr_count": 128, "vgpr_spill_count": 0, "private_segment_fixed_size": 0, "waves_per_simd": 4}}}
    ok = {"bf16": {"k": dict(ref["bf16"]["k"])}}
    assert ru.compare(ok, ref) == ([], [])
    for field, val in (("vgpr_spill_count", 3), ("private_segment_fixed_size", 16), ("waves_per_simd", 3)):
        cur = {"bf16": {"k": dict(ref["bf16"]["k"], **{field: val})}}
        bad, _ = ru.compare(cur, ref)
        assert len(bad) == 1
    bad, notes = ru.compare({"bf16": {"k": dict(ref["bf16"]["k"]), "new": dict(ref["bf16"]["k"])}}, ref)
    assert not bad and any("new kernel" in n for n in notes)
