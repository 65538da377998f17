"""Generates asva_amd/csrc/gemm4_loops.inc: the hand-scheduled main loops of csrc/gemm4.hip, one `asm volatile` block per tile
variant with every register named by hand (the schedule is described in gemm4.hip).  hipcc allocates registers of inline-asm
operands well until the register file is full — at 256 accumulator registers it starts copying accumulators between statements,
without the wait states an MFMA result needs — so each loop is ONE statement: the accumulators are its only outputs.

    python tools/gen_gemm4_loops.py > asva_amd/csrc/gemm4_loops.inc

Variants: (FM, FN) = 32-row / 32-column fragments per wave (4 waves as 2 x 2); planes = 1 (16-bit operands) or 2 (split precision,
AVSD_GEMM_X2: every operand a (main, rest) pair, three MFMAs per fragment pair); tmix = the temporal-mix A operand.
Register map (per lane): v0-v31 stay the compiler's; from v32: vector offsets VA[NA] VW[NW], LDS addresses (write / read x A / W x
stage x plane), fragment sets XF[sets][planes][FM] WF[sets][planes][FN] (4 registers each), staging G[NSTG][planes][NA + NW]
(4 registers each), and for tmix the per-vector segment jumps D01[NA] D12[NA].
"""
import os
import sys

ROWB = 144


def gen(FM, FN, NSTG, mfma_op, ablate="", deep=None, tmix=False, planes=1, base=32, rot=True):
    P = planes
    NA, NW = 2 * FM, 2 * FN
    NL = NA + NW
    BM, BN = 64 * FM, 64 * FN
    A_BYTES, W_BYTES = BM * ROWB, BN * ROWB
    PLANE = A_BYTES + W_BYTES
    STAGE = P * PLANE
    NMF, NFR = FM * FN, FM + FN
    NPASS = 3 if P == 2 else 1
    if deep is None:
        deep = NMF * NPASS <= 8  # small wave tiles: a k-step is <= 256 cycles of MFMA — fragments are read TWO k-steps ahead (4 sets)
    NSETS = 4 if deep else 2
    NWK = 2 if deep else 3       # k-steps that carry the write + reload pairs (the barrier follows the last of them)
    WPK = (NL + NWK - 1) // NWK
    r = base                     # first register of the loop (the compiler keeps v0 .. base - 1)

    def take(n):
        nonlocal r
        b = r
        r += n
        return b

    VA = [take(1) for _ in range(NA)]
    VW = [take(1) for _ in range(NW)]
    WRA = [[take(1) for _ in range(P)] for _ in range(2)]
    WRW = [[take(1) for _ in range(P)] for _ in range(2)]
    RDA = [[take(1) for _ in range(P)] for _ in range(2)]
    RDW = [[take(1) for _ in range(P)] for _ in range(2)]
    if tmix:       # temporal-mix A operand: per-vector jumps at the two K-segment boundaries (frame 0 -> previous frame -> current frame)
        D01 = [take(1) for _ in range(NA)]
        D12 = [take(1) for _ in range(NA)]
        DW = [take(1) for _ in range(NA)] if rot else None      # ... and from the last K tile of the slice back to its first (rotated K walk)
        TMP = take(1)
    if r % 2:
        r += 1                                   # 64-bit-aligned tuples from here on
    XF = [[[take(4) for _ in range(FM)] for _ in range(P)] for _ in range(NSETS)]
    WF = [[[take(4) for _ in range(FN)] for _ in range(P)] for _ in range(NSETS)]
    G = [[[take(4) for _ in range(NL)] for _ in range(P)] for _ in range(NSTG)]
    last = r - 1
    assert last <= 255, (FM, FN, P, last)
    assert NSTG * NL * P - P < 64 and (NFR + WPK) * P < 16, "vmcnt is a 6-bit, lgkmcnt a 4-bit counter"

    def v4(b):
        return f"v[{b}:{b + 3}]"

    out = []
    emit = out.append
    RS = {(0, 0): "%[rsA]", (1, 0): "%[rsW]", (0, 1): "%[rsAr]", (1, 1): "%[rsWr]"}

    def load(s, j):
        if "g" in ablate:
            return
        for pl in range(P):
            if j < NA:
                emit(f"buffer_load_dwordx4 {v4(G[s][pl][j])}, v{VA[j]}, {RS[(0, pl)]}, 0 offen")
            else:
                emit(f"buffer_load_dwordx4 {v4(G[s][pl][j])}, v{VW[j - NA]}, {RS[(1, pl)]}, 0 offen")

    def bump(j, inc):
        if "g" in ablate:
            return
        reg = VA[j] if j < NA else VW[j - NA]
        if tmix and j < NA:
            emit(f"v_cndmask_b32_e64 v{TMP}, 0, v{D01[j]}, %[m1]")
            emit(f"v_cndmask_b32_e64 v{TMP}, v{TMP}, v{D12[j]}, %[m2]")
            if rot:
                emit(f"v_cndmask_b32_e64 v{TMP}, v{TMP}, v{DW[j]}, %[m3]")
                emit(f"v_add3_u32 v{reg}, v{reg}, v{TMP}, %[inca]")
            else:
                emit(f"v_add3_u32 v{reg}, v{reg}, v{TMP}, {inc}")
        else:
            emit(f"v_add_u32 v{reg}, {inc}, v{reg}")

    def write(s, st, j):
        if "w" in ablate:
            return
        for pl in range(P):
            if j < NA:
                emit(f"ds_write_b128 v{WRA[st][pl]}, {v4(G[s][pl][j])} offset:{j * 32 * ROWB}")
            else:
                emit(f"ds_write_b128 v{WRW[st][pl]}, {v4(G[s][pl][j])} offset:{(j - NA) * 32 * ROWB}")

    def fread(fs, st, ks, rr):
        # fragment read rr of 0 .. P * NFR - 1: plane-major (all main-plane fragments, then the rest planes)
        if "r" in ablate:
            return
        pl, q = rr // NFR, rr % NFR
        if q < FM:
            emit(f"ds_read_b128 {v4(XF[fs][pl][q])}, v{RDA[st][pl]} offset:{q * 32 * ROWB + ks * 32}")
        else:
            emit(f"ds_read_b128 {v4(WF[fs][pl][q - FM])}, v{RDW[st][pl]} offset:{(q - FM) * 32 * ROWB + ks * 32}")

    def set_inc(ahead):
        # Offsets advance from sequence index t + ahead - 1 to t + ahead.  The K walk starts at tile %[kst] of the slice and wraps:
        # position = (kst + index) mod nk.  %[inc] = 0 past the end (the last tile is re-read, never consumed), 128 - 128 nk at the
        # wrap, else 128.
        emit(f"s_add_u32 %[tmp], %[t], {ahead}")
        if rot:
            emit("s_add_u32 %[tmp2], %[tmp], %[kst]")
            emit("s_cmp_eq_u32 %[tmp2], %[nk]")
            emit("s_cselect_b32 %[inc], %[winc], 128")
            if tmix:
                emit("s_cselect_b64 %[m3], -1, 0")
        emit("s_cmp_lt_u32 %[tmp], %[nk]")
        if rot:
            emit("s_cselect_b32 %[inc], %[inc], 0")
        else:
            emit("s_cselect_b32 %[inc], 128, 0")
        if tmix:
            # m1 / m2 = all ones when the offsets advance INTO segment 1 / 2 (global tile index == tps / 2 tps), m3 at the wrap; only when
            # that tile exists.  Global index of the next tile: kt0 + ((kst + index) mod nk)
            emit("s_cselect_b64 %[mv], -1, 0")
            if rot:
                emit("s_cselect_b32 %[inca], 128, 0")
                emit("s_and_b64 %[m3], %[m3], %[mv]")
                emit("s_cmp_ge_u32 %[tmp2], %[nk]")
                emit("s_cselect_b32 %[tmp], %[nk], 0")
                emit("s_sub_u32 %[tmp], %[tmp2], %[tmp]")
            emit("s_add_u32 %[tmp], %[tmp], %[kt0]")
            emit("s_cmp_eq_u32 %[tmp], %[tps]")
            emit("s_cselect_b64 %[m1], -1, 0")
            emit("s_and_b64 %[m1], %[m1], %[mv]")
            emit("s_cmp_eq_u32 %[tmp], %[tps2]")
            emit("s_cselect_b64 %[m2], -1, 0")
            emit("s_and_b64 %[m2], %[m2], %[mv]")
            if rot:          # a wrap that lands exactly on a segment boundary takes the wrap jump only
                emit("s_andn2_b64 %[m1], %[m1], %[m3]")
                emit("s_andn2_b64 %[m2], %[m2], %[m3]")

    # ---- prologue -----------------------------------------------------------------------------------------------------------------
    emit("s_nop 4")
    if tmix:
        # per-thread table left by the host code in the (still unused) second LDS stage: VA[NA] | D01[NA] | D12[NA]
        for i in range(NA):
            emit(f"ds_read_b32 v{VA[i]}, %[va0] offset:{4 * i}")
            emit(f"ds_read_b32 v{D01[i]}, %[va0] offset:{4 * (NA + i)}")
            emit(f"ds_read_b32 v{D12[i]}, %[va0] offset:{4 * (2 * NA + i)}")
            if rot:
                emit(f"ds_read_b32 v{DW[i]}, %[va0] offset:{4 * (3 * NA + i)}")
        emit("s_waitcnt lgkmcnt(0)")
    else:
        emit(f"v_mov_b32 v{VA[0]}, %[va0]")
        for i in range(1, NA):
            emit(f"v_add_u32 v{VA[i]}, %[sa], v{VA[i - 1]}")
    emit(f"v_mov_b32 v{VW[0]}, %[vw0]")
    for i in range(1, NW):
        emit(f"v_add_u32 v{VW[i]}, %[sw], v{VW[i - 1]}")
    for st in range(2):
        for pl in range(P):
            off = st * STAGE + pl * PLANE
            emit(f"v_add_u32 v{WRA[st][pl]}, {off}, %[wr0]" if off else f"v_mov_b32 v{WRA[st][pl]}, %[wr0]")
            emit(f"v_add_u32 v{WRW[st][pl]}, {off + A_BYTES}, %[wr0]")
            emit(f"v_add_u32 v{RDA[st][pl]}, {off}, %[rda0]" if off else f"v_mov_b32 v{RDA[st][pl]}, %[rda0]")
            emit(f"v_add_u32 v{RDW[st][pl]}, {off}, %[rdw0]" if off else f"v_mov_b32 v{RDW[st][pl]}, %[rdw0]")
    emit("s_mov_b32 %[t], 0")
    # tiles 0 .. NSTG - 1 in flight; offsets advance to tile (index of the load + 1) when that tile exists
    for s in range(NSTG):
        set_inc(s + 1)
        for j in range(NL):
            load(s, j)
            bump(j, "%[inc]")
    emit(f"s_waitcnt vmcnt({(NSTG - 1) * NL * P})")
    for j in range(NL):
        write(0, 0, j)
    set_inc(NSTG + 1)
    for j in range(NL):
        load(0, j)
        bump(j, "%[inc]")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for ks0 in range(2 if deep else 1):
        for rr in range(NFR * P):
            fread(ks0, 0, ks0, rr)
    emit("s_waitcnt lgkmcnt(0)")

    # ---- one K tile -----------------------------------------------------------------------------------------------------------------
    def tile(cs, gs, zero_c):
        # reloads of this iteration fetch tile t + NSTG + 1; afterwards the offsets advance when tile t + NSTG + 2 exists
        set_inc(NSTG + 2)
        for ks in range(4):
            fs = ks if deep else ks & 1
            nwr = max(0, min(WPK, NL - ks * WPK)) if ks < NWK else 0
            fillers = [("r", f) for f in range(NFR * P)] + [("w", ks * WPK + f) for f in range(nwr)]
            nf = len(fillers)
            dist = 2 if deep else 1                     # fragment prefetch distance in k-steps
            # MFMAs of the k-step.  Split precision: three passes in the order of gemm2_kernel<X2> (Wr.A, W.Ar, W.A) — pass-major, so the
            # three MFMAs of one accumulator are NMF instructions apart while every element still sums its products in that order
            mf = []
            for ps in range(NPASS):
                for i in range(NMF):
                    a, b = i // FM, i % FM
                    if P == 1:
                        wsrc, xsrc = WF[fs][0][a], XF[fs][0][b]
                    else:
                        wsrc = WF[fs][1][a] if ps == 0 else WF[fs][0][a]
                        xsrc = XF[fs][1][b] if ps == 1 else XF[fs][0][b]
                    c = "0" if (zero_c and ks == 0 and ps == 0) else f"%[acc{a * FM + b}]"
                    mf.append(f"{mfma_op} %[acc{a * FM + b}], {v4(wsrc)}, {v4(xsrc)}, {c}")
            nslots = len(mf)
            for i in range(nslots):
                emit(mf[i])
                for fi, (kind, x) in enumerate(fillers):
                    slot = fi if nf <= nslots else fi * nslots // nf
                    if slot != i:
                        continue
                    if kind == "r":
                        kt = ks + dist                  # the k-step whose fragments are fetched: of this tile, or of the next one
                        fset = (kt % 4) if deep else (fs ^ 1)
                        if kt < 4:
                            fread(fset, cs, kt, x)
                        else:
                            fread(fset, cs ^ 1, kt - 4, x)
                    else:
                        emit(f"s_waitcnt vmcnt({NSTG * NL * P - P})")
                        write(gs, cs ^ 1, x)
                        load(gs, x)
                        bump(x, "%[inc]")
            if ks == NWK - 1:
                # all writes of tile t+1 issued; the barrier also needs every fragment read of THIS stage that will ever be issued
                # before the next write to it: they all precede this point (the reads after it go to the other stage)
                emit("s_waitcnt lgkmcnt(0)")
                if "b" not in ablate:
                    emit("s_barrier")
            elif deep:
                # the reads issued ONE k-step earlier (older than this k-step's) are in.  lgkmcnt is a 4-bit counter: with this k-step's operations
                # still out and the next k-step's on top, the 128 x 128 tile could stand at 16 if LDS answered nothing for a whole k-step.  Never
                # more than 15 outstanding, whatever LDS does (`worst` below checks the whole stream and inserts waits where a burst exceeds it).
                nwr_next = max(0, min(WPK, NL - (ks + 1) * WPK)) if ks + 1 < NWK else 0
                emit(f"s_waitcnt lgkmcnt({min((NFR + nwr) * P, 15 - (NFR + nwr_next) * P)})")
            elif ks == 3:
                emit("s_waitcnt lgkmcnt(0)")
            else:
                nwr_next = max(0, min(WPK, NL - (ks + 1) * WPK)) if ks + 1 < NWK else 0
                emit(f"s_waitcnt lgkmcnt({min(nwr * P, 15 - (NFR + nwr_next) * P)})")

    # K tile t multiplies LDS stage t & 1 and refills staging set (t + 1) % NSTG: the loop body repeats every L = lcm(2, NSTG) tiles
    import math
    L = 2 * NSTG // math.gcd(2, NSTG)
    tile(0, 1 % NSTG, True)                       # t = 0 (first MFMA of every accumulator takes C = 0)
    emit("s_add_u32 %[t], %[t], 1")
    emit("s_cmp_ge_u32 %[t], %[nk]")
    emit("s_cbranch_scc1 L_end_%=")
    emit("L_loop_%=:")
    for u in range(1, L + 1):                     # t = u (mod L)
        tile(u & 1, (u + 1) % NSTG, False)
        emit("s_add_u32 %[t], %[t], 1")
        if u < L:
            emit("s_cmp_ge_u32 %[t], %[nk]")
            emit("s_cbranch_scc1 L_end_%=")
        else:
            emit("s_cmp_lt_u32 %[t], %[nk]")
            emit("s_cbranch_scc1 L_loop_%=")
    emit("L_end_%=:")
    # NOTHING may be in flight when the block ends: on the exit paths of the deep schedule the fragment prefetch of the (non-existent)
    # next K tile is still out, and a ds_read that lands AFTER the block overwrites whatever the compiler has put into those VGPRs by
    # then (it only knows them as clobbered: the epilogue keeps accumulator values there).  Found as 1-2 % grossly wrong launches of
    # the 128 x 128 / 64 x 128 tiles — always the same accumulator registers — once a second process kept the CUs' LDS busy
    # (tests/test_determinism_gpu.py runs its two children side by side since round 4); never on an exclusive GPU, where the data
    # lands within the two s_nop below.  36 000 contended launches clean with this wait.
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")

    # worst case of the two memory counters over the whole instruction stream (nothing completes until a wait forces it; the loop body
    # walked twice): lgkmcnt holds 0..15, vmcnt 0..63, and both WRAP
    def worst(prefix, wait_re, limit, fix=None):
        import re
        while True:
            lo = next(i for i, ln in enumerate(out) if ln.startswith("L_loop_"))
            hi = next(i for i, ln in enumerate(out) if ln.startswith("L_end_"))
            cnt = peak = 0
            over = None
            for i in list(range(hi)) + list(range(lo, hi)):
                ln = out[i]
                m = re.match(wait_re, ln)
                if m:
                    cnt = min(cnt, int(m.group(1)))
                elif ln.startswith(prefix):
                    cnt += 1
                    peak = max(peak, cnt)
                    if cnt > limit:
                        over = i
                        break
            if over is None:
                return peak
            assert fix is not None, (FM, FN, P, tmix, prefix, peak)
            out.insert(over, fix)               # the oldest operation completes before this one is issued
    worst(("ds_read", "ds_write"), r"s_waitcnt lgkmcnt\((\d+)\)", 15, "s_waitcnt lgkmcnt(14)")
    worst(("buffer_load",), r"s_waitcnt vmcnt\((\d+)\)", 63)

    name = f"g4_loop_{FM}x{FN}_s{NSTG}" + (f"_ab_{ablate}" if ablate else "") + ("_tmix" if tmix else "") + ("_x2" if P == 2 else "")
    nacc = FM * FN
    lines = []
    lines.append(f"// {BM} x {BN} tile, {32 * FM} x {32 * FN} per wave, {NSTG} K tile(s) of global loads in flight, {P} plane(s)"
                 f"{', temporal-mix A' if tmix else ''}{', rotated K walk' if rot else ''}; registers v{base} .. v{last}")
    lines.append(f"__device__ __forceinline__ void {name}(f32x16 (&acc)[{FN}][{FM}], const G4Args& a) {{")
    if tmix:
        lines.append("  unsigned long long m1, m2, mv" + (", m3;" if rot else ";"))
    lines.append("  unsigned t, tmp, inc" + (", tmp2" if rot else "") + (", inca;" if (rot and tmix) else ";"))
    lines.append("  asm volatile(")
    for ln in out:
        lines.append(f'      "{ln}\\n\\t"')
    outs = ", ".join(f'[acc{x * FM + b}] "=&a"(acc[{x}][{b}])' for x in range(FN) for b in range(FM))
    lines.append(f"      : {outs},")
    lines.append('        [t] "=&s"(t), [tmp] "=&s"(tmp), [inc] "=&s"(inc)' + (', [tmp2] "=&s"(tmp2)' if rot else "") + (', [inca] "=&s"(inca), [m3] "=&s"(m3)' if (rot and tmix) else "") +
                 (', [m1] "=&s"(m1), [m2] "=&s"(m2), [mv] "=&s"(mv)' if tmix else ""))
    lines.append('      : [va0] "v"(a.va0), [vw0] "v"(a.vw0), [wr0] "v"(a.wr0), [rda0] "v"(a.rda0), [rdw0] "v"(a.rdw0), [rsA] "s"(a.rsA), [rsW] "s"(a.rsW),')
    if P == 2:
        lines.append('        [rsAr] "s"(a.rsAr), [rsWr] "s"(a.rsWr),')
    if rot:
        lines.append('        [kst] "s"(a.kst), [winc] "s"(a.winc),')
    if tmix:
        lines.append('        [kt0] "s"(a.kt0), [sw] "s"(a.sw), [nk] "s"(a.nk), [tps] "s"(a.tps), [tps2] "s"(a.tps2)')
    else:
        lines.append('        [sa] "s"(a.sa), [sw] "s"(a.sw), [nk] "s"(a.nk)')
    clob = ", ".join(f'"v{i}"' for i in range(base, last + 1))
    lines.append(f'      : "memory", "scc", {clob});')
    lines.append("}")
    if not ablate:
        lines.append(f"template <> struct G4Loop<{FM}, {FN}, {NSTG}, {'true' if tmix else 'false'}, {'true' if P == 2 else 'false'}> {{")
        lines.append(f"  static constexpr bool rot = {'true' if rot else 'false'};      // takes a rotated K walk (G4Args::kst != 0)")
        lines.append(f"  static __device__ __forceinline__ void run(f32x16 (&acc)[{FN}][{FM}], const G4Args& a) {{ {name}(acc, a); }}")
        lines.append("};")
    lines.append("")
    nops = nacc + 3 + 10 + (5 if tmix else 0) + (2 if P == 2 else 0) + ((3 + (2 if tmix else 0)) if rot else 0)
    assert nops <= 30, (FM, FN, P, tmix, nops)
    return "\n".join(lines), name


def main():
    print("// GENERATED by tools/gen_gemm4_loops.py — do not edit; see csrc/gemm4.hip for the schedule.")
    print("// clang-format off")
    print("""// Arguments of a loop (all wave-uniform except the five per-lane addresses): byte offsets of this thread's first A / W vector in the
// first K tile it loads (tmix: LDS address of its per-vector table instead of va0), LDS addresses of its vector slot / fragment rows,
// buffer descriptors (rsAr / rsWr: the rest planes, split precision), kst = first K tile of the walk inside the slice and
// winc = 128 - 128 nk (the wrap), sa / sw = byte distance between this thread's consecutive vectors (32 rows), nk = K tiles of the slice,
// tmix: kt0 = first K tile of the slice, tps / tps2 = K tiles per segment / twice that.
struct G4Args {
  unsigned va0, vw0, wr0, rda0, rdw0;
  u32x4 rsA, rsW, rsAr, rsWr;
  unsigned kst, winc, sa, sw, nk, kt0, tps, tps2;
};
template <int FM, int FN, int NS, bool TMIX, bool X2> struct G4Loop;
""")

    def put(*a, **k):
        txt, _ = gen(*a, "AVSD_MFMA_OP_PLACEHOLDER", **k)
        print(txt.replace('"AVSD_MFMA_OP_PLACEHOLDER ', 'AVSD_MFMA_OP " '))

    for FM, FN in ((4, 4), (4, 2), (2, 4), (2, 2), (2, 1), (1, 2), (1, 1)):
        put(FM, FN, 2, rot=(FM * FN < 16))
    for FM, FN in ((4, 2), (2, 4), (2, 2), (2, 1), (1, 2), (1, 1)):
        put(FM, FN, 2, tmix=True, deep=(FM * FN <= 4), rot=(FM * FN <= 4))
    # 128 x 320 (round 5): the N = 320 layers of the 32 x 32 level in ONE column tile — A is read once per row band, no padded columns
    # (128-column tiles pad 320 to 384), 7 fragment reads per 10 MFMAs.  The temporal-mix form walks K unrotated (an asm statement
    # takes 30 operands; the rotated temporal-mix walk needs 33 with 10 accumulators) — W of these layers is L2-resident anyway.
    put(2, 5, 2, rot=True)
    put(2, 5, 2, tmix=True, deep=False, rot=False)
    # (NSTG = 4, four K tiles of global loads in flight, was built for the three small tiles as ids 67-69: no shape of the
    #  tuned table chose it over NSTG = 2 once the K walk was rotated, so the variants are not emitted; gen() still takes NSTG)
    # split precision (two planes per operand, three MFMA passes)
    for FM, FN in ((2, 2), (2, 1), (1, 2), (1, 1)):
        put(FM, FN, 2, planes=2)
        put(FM, FN, 2, planes=2, tmix=True, base=(28 if (FM, FN) == (2, 2) else 32), rot=((FM, FN) == (1, 1)))
    if "--ablate" in sys.argv:
        print("#define AVSD_G4_ABLATE 1")
        for ab in ("g", "w", "r", "gw", "gwr", "b", "gwrb"):
            put(4, 4, 2, ablate=ab)


if __name__ == "__main__":
    main()
