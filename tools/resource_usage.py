"""Register / LDS / scratch usage of EVERY kernel of the shipped libraries, read from the code objects themselves.

Round 4 lost 8-12 % on the VAE decode and cfg 4 to a register-allocation change that no switch-level A/B could see
(profiles/r4_regalloc_ab.txt): edits that never touch a main loop moved the most-used GEMM tile from 2 waves per SIMD to 1 and
made another spill 96 registers.  This tool makes that visible at build time, with no GPU:

    python tools/resource_usage.py                      # table of both libraries -> stdout
    python tools/resource_usage.py --write              # refresh asva_amd/resource_usage_gfx950.json (the committed table) and
                                                        #   profiles/r6_resource_usage.txt (the readable dump)
    python tools/resource_usage.py --check              # compare the built libraries with the committed table (exit 1 on a regression)

Source of the numbers: the AMDGPU metadata note of each gfx950 code object embedded in libavsd_hip*.so (llvm-objdump --offloading,
llvm-readelf --notes) — .vgpr_count (unified VGPR + AGPR budget), .agpr_count, .sgpr_count, .vgpr_spill_count,
.private_segment_fixed_size (scratch bytes per lane), .group_segment_fixed_size (static LDS), .max_flat_workgroup_size.  The same
figures hipcc prints under -Rpass-analysis=kernel-resource-usage, taken from the binaries that ship instead of from a second compile.
waves/SIMD = min(8, 512 // (ceil(vgpr_count / 8) * 8)) (MI355X_MICROARCH.md, register files); dynamic LDS is a launch argument and is
not part of the table.

tests/test_resource_usage.py runs --check semantics in the CPU suite: a kernel that starts to spill, grows its scratch, or loses a wave
per SIMD against the committed table fails the build check; new / removed kernels ask for `--write`.
"""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIBS = {"bf16": os.path.join(ROOT, "asva_amd", "libavsd_hip.so"), "fp16": os.path.join(ROOT, "asva_amd", "libavsd_hip_f16.so")}
TABLE = os.path.join(ROOT, "asva_amd", "resource_usage_gfx950.json")
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def waves_per_simd(vgpr_count: int) -> int:
    alloc = max(8, -(-vgpr_count // 8) * 8)
    return min(8, 512 // alloc)


def _demangle(names):
    r = subprocess.run([shutil.which("c++filt") or "c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    out = r.stdout.splitlines()
    clean = []
    for n in out:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "", 1) if n.startswith("void ") else n.replace("(anonymous namespace)::", "")
        n = re.sub(r"\((avsd_gemm_desc|.*)\)$", "", n) if n.endswith(")") else n
        clean.append(n)
    return clean


def kernels_of(lib_path: str) -> dict:
    """{kernel name: {field: value, 'waves_per_simd': n}} over every gfx950 code object in the library"""
    tmp = tempfile.mkdtemp(prefix="avsd_ru_")
    try:
        local = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, text=True, check=True, cwd=tmp)
        rows = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            cur = None
            entries = []
            for line in notes.splitlines():
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
                if not m:
                    continue
                key, val = m.groups()
                if key == "agpr_count":          # first key of a kernel entry (keys are sorted)
                    cur = {}
                    entries.append(cur)
                if cur is not None and (key in FIELDS or key == "name"):
                    cur[key] = val
            names = _demangle([e["name"] for e in entries])
            for e, n in zip(entries, names):
                row = {k: int(e.get(k, 0)) for k in FIELDS}
                row["waves_per_simd"] = waves_per_simd(row["vgpr_count"])
                rows[n] = row
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def collect() -> dict:
    return {var: kernels_of(p) for var, p in LIBS.items() if os.path.isfile(p)}


def compare(built: dict, table: dict):
    """-> (regressions, notes): regressions fail the check; notes are improvements / membership changes"""
    bad, notes = [], []
    for var, ref in table.items():
        cur = built.get(var)
        if cur is None:
            bad.append(f"{var}: library not built")
            continue
        for name, r in ref.items():
            c = cur.get(name)
            if c is None:
                notes.append(f"{var}: {name} is gone (refresh the table: python tools/resource_usage.py --write)")
                continue
            if c["vgpr_spill_count"] > r["vgpr_spill_count"]:
                bad.append(f"{var}: {name} spills {c['vgpr_spill_count']} VGPRs (table: {r['vgpr_spill_count']})")
            if c["private_segment_fixed_size"] > r["private_segment_fixed_size"]:
                bad.append(f"{var}: {name} uses {c['private_segment_fixed_size']} B of scratch per lane (table: {r['private_segment_fixed_size']})")
            if c["waves_per_simd"] < r["waves_per_simd"]:
                bad.append(f"{var}: {name} {r['vgpr_count']} -> {c['vgpr_count']} registers: {r['waves_per_simd']} -> {c['waves_per_simd']} waves per SIMD")
            elif c["waves_per_simd"] > r["waves_per_simd"] or c["vgpr_spill_count"] < r["vgpr_spill_count"]:
                notes.append(f"{var}: {name} improved ({r['vgpr_count']} -> {c['vgpr_count']} registers, spills {r['vgpr_spill_count']} -> {c['vgpr_spill_count']})")
        for name in cur:
            if name not in ref:
                notes.append(f"{var}: new kernel {name} (refresh the table: python tools/resource_usage.py --write)")
    return bad, notes


def dump(built: dict) -> str:
    lines = ["kernel | VGPR+AGPR (AGPR) | SGPR | waves/SIMD | spilled VGPR / SGPR | scratch B | static LDS B | max threads"]
    for var, rows in built.items():
        lines.append(f"== {os.path.basename(LIBS[var])}: {len(rows)} kernels, {sum(1 for r in rows.values() if r['vgpr_spill_count'])} spill VGPRs, "
                     f"{sum(1 for r in rows.values() if r['private_segment_fixed_size'])} use scratch")
        for n in sorted(rows):
            r = rows[n]
            lines.append(f"{n} | {r['vgpr_count']} ({r['agpr_count']}) | {r['sgpr_count']} | {r['waves_per_simd']} | {r['vgpr_spill_count']} / {r['sgpr_spill_count']} | "
                         f"{r['private_segment_fixed_size']} | {r['group_segment_fixed_size']} | {r['max_flat_workgroup_size']}")
    return "\n".join(lines) + "\n"


def main():
    built = collect()
    if "--write" in sys.argv:
        with open(TABLE, "w") as f:
            json.dump(built, f, indent=0, sort_keys=True)
        out = os.path.join(ROOT, "profiles", "r6_resource_usage.txt")
        with open(out, "w") as f:
            f.write(dump(built))
        print(f"wrote {TABLE} and {out}: " + ", ".join(f"{v} {len(r)} kernels" for v, r in built.items()))
        return 0
    if "--check" in sys.argv:
        with open(TABLE) as f:
            table = json.load(f)
        bad, notes = compare(built, table)
        for n in notes:
            print("note:", n)
        for b in bad:
            print("REGRESSION:", b)
        return 1 if bad else 0
    sys.stdout.write(dump(built))
    return 0


if __name__ == "__main__":
    sys.exit(main())
