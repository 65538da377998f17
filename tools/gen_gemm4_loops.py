"""Generates asva_amd/csrc/gemm4_loops.inc: the hand-scheduled main loops of csrc/gemm4.hip, one `asm volatile` block per tile
variant with every register named by hand (the schedule is described in gemm4.hip).  hipcc allocates registers of inline-asm
operands well until the register file is full — at 256 accumulator registers it starts copying accumulators between statements,
without the wait states an MFMA result needs — so each loop is ONE statement: the accumulators are its only outputs.

    python tools/gen_gemm4_loops.py > asva_amd/csrc/gemm4_loops.inc

Register map (per lane): v0-v31 stay the compiler's; from v32: 16-byte-vector offsets VA[NA] VW[NW], LDS addresses (write A/W x 2
stages, read A/W x 2 stages), fragment sets XF[2][FM] WF[2][FN] (4 registers each), staging G[NSTG][NA + NW] (4 registers each).
"""
import sys

ROWB = 144


def gen(FM, FN, NSTG, mfma_op, ablate="", deep=None, tmix=False):
    NA, NW = 2 * FM, 2 * FN
    NL = NA + NW
    BM, BN = 64 * FM, 64 * FN
    A_BYTES, W_BYTES = BM * ROWB, BN * ROWB
    STAGE = A_BYTES + W_BYTES
    NMF, NFR = FM * FN, FM + FN
    if deep is None:
        deep = NMF <= 8          # small wave tiles: a k-step is <= 256 cycles of MFMA — fragments are read TWO k-steps ahead (4 sets)
    NSETS = 4 if deep else 2
    NWK = 2 if deep else 3       # k-steps that carry the write + reload pairs (the barrier follows the last of them)
    WPK = (NL + NWK - 1) // NWK
    r = 32

    def take(n):
        nonlocal r
        b = r
        r += n
        return b

    VA = [take(1) for _ in range(NA)]
    VW = [take(1) for _ in range(NW)]
    WRA = [take(1) for _ in range(2)]
    WRW = [take(1) for _ in range(2)]
    RDA = [take(1) for _ in range(2)]
    RDW = [take(1) for _ in range(2)]
    if r % 2:
        r += 1                                   # 64-bit-aligned tuples from here on (ds_read_b128 / MFMA sources want even bases)
    XF = [[take(4) for _ in range(FM)] for _ in range(NSETS)]
    WF = [[take(4) for _ in range(FN)] for _ in range(NSETS)]
    G = [[take(4) for _ in range(NL)] for _ in range(NSTG)]
    if tmix:       # temporal-mix A operand: per-vector jumps at the two K-segment boundaries (frame 0 -> previous frame -> current frame)
        D01 = [take(1) for _ in range(NA)]
        D12 = [take(1) for _ in range(NA)]
        TMP = take(1)
    last = r - 1
    assert last <= 255, last

    def v4(b):
        return f"v[{b}:{b + 3}]"

    out = []
    emit = out.append

    def load(s, j):
        if "g" in ablate:
            return
        if j < NA:
            emit(f"buffer_load_dwordx4 {v4(G[s][j])}, v{VA[j]}, %[rsA], 0 offen")
        else:
            emit(f"buffer_load_dwordx4 {v4(G[s][j])}, v{VW[j - NA]}, %[rsW], 0 offen")

    def bump(j, inc):
        if "g" in ablate:
            return
        reg = VA[j] if j < NA else VW[j - NA]
        if tmix and j < NA:
            emit(f"v_cndmask_b32_e64 v{TMP}, 0, v{D01[j]}, %[m1]")
            emit(f"v_cndmask_b32_e64 v{TMP}, v{TMP}, v{D12[j]}, %[m2]")
            emit(f"v_add3_u32 v{reg}, v{reg}, v{TMP}, {inc}")
        else:
            emit(f"v_add_u32 v{reg}, {inc}, v{reg}")

    def write(s, st, j):
        if "w" in ablate:
            return
        if j < NA:
            emit(f"ds_write_b128 v{WRA[st]}, {v4(G[s][j])} offset:{j * 32 * ROWB}")
        else:
            emit(f"ds_write_b128 v{WRW[st]}, {v4(G[s][j])} offset:{(j - NA) * 32 * ROWB}")

    def fread(fs, st, ks, rr):
        if "r" in ablate:
            return
        if rr < FM:
            emit(f"ds_read_b128 {v4(XF[fs][rr])}, v{RDA[st]} offset:{rr * 32 * ROWB + ks * 32}")
        else:
            emit(f"ds_read_b128 {v4(WF[fs][rr - FM])}, v{RDW[st]} offset:{(rr - FM) * 32 * ROWB + ks * 32}")

    def set_inc(ahead):
        # %[inc] = (t + ahead < nk) ? 128 : 0   — the offsets of a tile past the end of K stay on the last tile
        emit(f"s_add_u32 %[tmp], %[t], {ahead}")
        emit("s_cmp_lt_u32 %[tmp], %[nk]")
        emit("s_cselect_b32 %[inc], 128, 0")
        if tmix:
            # m1 / m2 = all ones when the offsets advance INTO segment 1 / 2 (global tile index kt0 + t + ahead == tps / 2 tps) and that tile exists
            emit("s_cselect_b64 %[mv], -1, 0")
            emit("s_add_u32 %[tmp], %[tmp], %[kt0]")
            emit("s_cmp_eq_u32 %[tmp], %[tps]")
            emit("s_cselect_b64 %[m1], -1, 0")
            emit("s_and_b64 %[m1], %[m1], %[mv]")
            emit("s_cmp_eq_u32 %[tmp], %[tps2]")
            emit("s_cselect_b64 %[m2], -1, 0")
            emit("s_and_b64 %[m2], %[m2], %[mv]")

    # ---- prologue -----------------------------------------------------------------------------------------------------------------
    emit("s_nop 4")
    if tmix:
        # per-thread table left by the host code in the (still unused) second LDS stage: VA[NA] | D01[NA] | D12[NA]
        for i in range(NA):
            emit(f"ds_read_b32 v{VA[i]}, %[va0] offset:{4 * i}")
            emit(f"ds_read_b32 v{D01[i]}, %[va0] offset:{4 * (NA + i)}")
            emit(f"ds_read_b32 v{D12[i]}, %[va0] offset:{4 * (2 * NA + i)}")
        emit("s_waitcnt lgkmcnt(0)")
    else:
        emit(f"v_mov_b32 v{VA[0]}, %[va0]")
        for i in range(1, NA):
            emit(f"v_add_u32 v{VA[i]}, %[sa], v{VA[i - 1]}")
    emit(f"v_mov_b32 v{VW[0]}, %[vw0]")
    for i in range(1, NW):
        emit(f"v_add_u32 v{VW[i]}, %[sw], v{VW[i - 1]}")
    emit(f"v_mov_b32 v{WRA[0]}, %[wr0]")
    emit(f"v_add_u32 v{WRW[0]}, {A_BYTES}, v{WRA[0]}")
    emit(f"v_add_u32 v{WRA[1]}, {STAGE}, v{WRA[0]}")
    emit(f"v_add_u32 v{WRW[1]}, {STAGE + A_BYTES}, v{WRA[0]}")
    emit(f"v_mov_b32 v{RDA[0]}, %[rda0]")
    emit(f"v_add_u32 v{RDA[1]}, {STAGE}, v{RDA[0]}")
    emit(f"v_mov_b32 v{RDW[0]}, %[rdw0]")
    emit(f"v_add_u32 v{RDW[1]}, {STAGE}, v{RDW[0]}")
    emit("s_mov_b32 %[t], 0")
    # tiles 0 .. NSTG - 1 in flight; offsets advance to tile (index of the load + 1) when that tile exists
    for s in range(NSTG):
        set_inc(s + 1)
        for j in range(NL):
            load(s, j)
            bump(j, "%[inc]")
    emit(f"s_waitcnt vmcnt({(NSTG - 1) * NL})")
    for j in range(NL):
        write(0, 0, j)
    set_inc(NSTG + 1)
    for j in range(NL):
        load(0, j)
        bump(j, "%[inc]")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for ks0 in range(2 if deep else 1):
        for rr in range(NFR):
            fread(ks0, 0, ks0, rr)
    emit("s_waitcnt lgkmcnt(0)")

    # ---- one K tile -----------------------------------------------------------------------------------------------------------------
    def tile(cs, gs, zero_c):
        # reloads of this iteration fetch tile t + NSTG + 1; afterwards the offsets advance when tile t + NSTG + 2 exists
        set_inc(NSTG + 2)
        for ks in range(4):
            fs = ks if deep else ks & 1
            nwr = max(0, min(WPK, NL - ks * WPK)) if ks < NWK else 0
            fillers = [("r", f) for f in range(NFR)] + [("w", ks * WPK + f) for f in range(nwr)]
            nf = len(fillers)
            dist = 2 if deep else 1                     # fragment prefetch distance in k-steps
            for i in range(NMF):
                a, b = i // FM, i % FM
                c = "0" if (zero_c and ks == 0) else f"%[acc{a * FM + b}]"
                emit(f"{mfma_op} %[acc{a * FM + b}], {v4(WF[fs][a])}, {v4(XF[fs][b])}, {c}")
                for fi, (kind, x) in enumerate(fillers):
                    slot = fi if nf <= NMF else fi * NMF // nf
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
                        emit(f"s_waitcnt vmcnt({NSTG * NL - 1})")
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
                emit(f"s_waitcnt lgkmcnt({NFR + nwr})")       # the reads issued ONE k-step earlier (older than this k-step's) are in
            elif ks == 3:
                emit("s_waitcnt lgkmcnt(0)")
            else:
                emit(f"s_waitcnt lgkmcnt({nwr})")

    gs_even = NSTG - 1          # staging set that holds tile t + 1 when t is even
    tile(0, gs_even, True)                        # t = 0
    emit("s_add_u32 %[t], %[t], 1")
    emit("s_cmp_ge_u32 %[t], %[nk]")
    emit("s_cbranch_scc1 L_end_%=")
    emit("L_loop_%=:")
    tile(1, 0, False)                             # odd t
    emit("s_add_u32 %[t], %[t], 1")
    emit("s_cmp_ge_u32 %[t], %[nk]")
    emit("s_cbranch_scc1 L_end_%=")
    tile(0, gs_even, False)                       # even t
    emit("s_add_u32 %[t], %[t], 1")
    emit("s_cmp_lt_u32 %[t], %[nk]")
    emit("s_cbranch_scc1 L_loop_%=")
    emit("L_end_%=:")
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")

    name = f"g4_loop_{FM}x{FN}_s{NSTG}" + (f"_ab_{ablate}" if ablate else "") + ("_tmix" if tmix else "")
    nacc = FM * FN
    lines = []
    lines.append(f"// {BM} x {BN} tile, {32 * FM} x {32 * FN} per wave, {NSTG} K tile(s) of global loads in flight; registers v32 .. v{last}")
    lines.append(f"__device__ __forceinline__ void {name}(f32x16 (&acc)[{FN}][{FM}], unsigned va0, unsigned vw0, unsigned wr0, unsigned rda0, unsigned rdw0,")
    if tmix:
        lines.append("                                           u32x4 rsA, u32x4 rsW, unsigned kt0, unsigned sw, unsigned nk, unsigned tps, unsigned tps2) {")
        lines.append("  unsigned long long m1, m2, mv;")
    else:
        lines.append("                                           u32x4 rsA, u32x4 rsW, unsigned sa, unsigned sw, unsigned nk) {")
    lines.append("  unsigned t, tmp, inc;")
    lines.append("  asm volatile(")
    for ln in out:
        lines.append(f'      "{ln}\\n\\t"')
    outs = ", ".join(f'[acc{a * FM + b}] "=&a"(acc[{a}][{b}])' for a in range(FN) for b in range(FM))
    lines.append(f"      : {outs},")
    lines.append('        [t] "=&s"(t), [tmp] "=&s"(tmp), [inc] "=&s"(inc)' + (', [m1] "=&s"(m1), [m2] "=&s"(m2), [mv] "=&s"(mv)' if tmix else ""))
    lines.append('      : [va0] "v"(va0), [vw0] "v"(vw0), [wr0] "v"(wr0), [rda0] "v"(rda0), [rdw0] "v"(rdw0), [rsA] "s"(rsA), [rsW] "s"(rsW),')
    if tmix:
        lines.append('        [kt0] "s"(kt0), [sw] "s"(sw), [nk] "s"(nk), [tps] "s"(tps), [tps2] "s"(tps2)')
    else:
        lines.append('        [sa] "s"(sa), [sw] "s"(sw), [nk] "s"(nk)')
    clob = ", ".join(f'"v{i}"' for i in range(32, last + 1))
    lines.append(f'      : "memory", "scc", {clob});')
    lines.append("}")
    lines.append("")
    assert nacc + 3 + 10 + (5 if tmix else 0) <= 30
    return "\n".join(lines), name


def main():
    print("// GENERATED by tools/gen_gemm4_loops.py — do not edit; see csrc/gemm4.hip for the schedule.")
    print("// clang-format off")
    for FM, FN, NSTG in ((4, 4, 2), (4, 2, 2), (2, 4, 2), (2, 2, 2), (2, 1, 2), (1, 2, 2), (1, 1, 2)):
        txt, _ = gen(FM, FN, NSTG, "AVSD_MFMA_OP_PLACEHOLDER")
        print(txt.replace('"AVSD_MFMA_OP_PLACEHOLDER ', 'AVSD_MFMA_OP " '))
    for FM, FN in ((4, 2), (2, 4), (2, 2), (2, 1), (1, 2), (1, 1)):
        txt, _ = gen(FM, FN, 2, "AVSD_MFMA_OP_PLACEHOLDER", tmix=True, deep=(FM * FN <= 4))
        print(txt.replace('"AVSD_MFMA_OP_PLACEHOLDER ', 'AVSD_MFMA_OP " '))
    if "--ablate" in sys.argv:
        print("#define AVSD_G4_ABLATE 1")
        for ab in ("g", "w", "r", "gw", "gwr", "b", "gwrb"):
            txt, _ = gen(4, 4, 2, "AVSD_MFMA_OP_PLACEHOLDER", ab)
            print(txt.replace('"AVSD_MFMA_OP_PLACEHOLDER ', 'AVSD_MFMA_OP " '))


if __name__ == "__main__":
    main()
