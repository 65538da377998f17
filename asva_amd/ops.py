"""Torch-tensor front end of the C ABI (include/avsd.h): one thin function per entry point.

Tensors are device-resident; activations are channels-last bf16 matrices [rows, C].  Every call
launches on torch's current HIP stream, so a `torch.cuda.graph` capture records the whole step.
PyTorch is used for memory and streams only; all arithmetic happens in libavsd_hip.so.
"""
from __future__ import annotations

import os
import ctypes as C
from typing import Optional

import torch

from . import _lib
from . import precision as P
from ._lib import GemmDesc, XAttnDesc, check

PLAIN, TMIX, CONV3 = 0, 1, 2
GEGLU, OUT_F32, GELU, XCD_N, ROWSTATS, LNFUSE, RES1_F32, RES2_F32, X2, KROT, W_FRAG, OUT_REST = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048
_XCD_MODE = "auto"     # auto | m | n  (which operand each XCD's L2 fetches once)
F32 = torch.float32      # (16-bit storage dtype: P.ACT, asva_amd/precision.py)


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg): while
    installed via `set_timer`, every wrapper below brackets its launch with two events and records
    (family, algorithmic flops, algorithmic bytes).  Off (None) in the product path."""

    def __init__(self):
        self.records = []
        self.replays = []      # (family, callable re-issuing the identical launch, tensors kept alive)

    def start(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, ev0, family: str, flops: float, nbytes: float, flops_once: Optional[float] = None, flops_ref: Optional[float] = None):
        """flops: multiply-adds x 2 the launch ISSUES (the three passes of a split product counted three times); flops_once: the same
        product at one pass; flops_ref: the product as the reference states it (one pass; the 3x3 taps on the upsampled image for the
        sub-pixel form of the upsample convolution)"""
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        once = flops if flops_once is None else flops_once
        self.records.append((family, flops, nbytes, ev0, ev1, once, once if flops_ref is None else flops_ref))

    def add_replay(self, family: str, fn, keep):
        self.replays.append((family, fn, keep))

    def replay_ms(self, families, reps: int = 3) -> float:
        """GPU time of all recorded launches of `families`, issued back to back from one captured graph (no per-launch
        event, no host launch floor): what a rocprofv3 kernel trace of the graph-replayed step sees for them."""
        fns = [fn for fam, fn, _ in self.replays if fam in families]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for fn in fns:
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, fl, nb, e0, e1, fo, fr in self.records:
            d = out.setdefault(fam, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "flops_once": 0.0, "flops_ref": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
            d["flops_once"] += fo
            d["flops_ref"] += fr
        return out


_TIMER: Optional[KernelTimer] = None


def set_timer(t: Optional[KernelTimer]) -> None:
    global _TIMER
    _TIMER = t


def _timed(family: str, flops: float, nbytes: float, keep, *calls) -> None:
    """issues `calls` (zero-argument callables, one library launch each, reading the current stream when called) and, under a
    KernelTimer, records their event-pair time and registers them for the graph-replayed per-family timing"""
    ev = _TIMER.start() if _TIMER is not None else None
    for c in calls:
        c()
    if ev is not None:
        _TIMER.add_replay(family, lambda: [c() for c in calls], keep)
        _TIMER.stop(ev, family, flops, nbytes)


def _nbytes(*ts) -> float:
    return float(sum(t.numel() * t.element_size() for t in ts if t is not None))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- per-shape tile selection ---------------------------------------------------------------------------
# The GEMM kernel exists in several tile/wave/stage configurations (gemm.hip: dispatch_tile).  Which one is
# fastest depends on (mode, M, N, K, epilogue).  Selection is DETERMINISTIC by default, so two processes produce
# bit-identical results (the f32 summation order depends on the tile):
#   1. the committed table asva_amd/tiles_gfx950.json (tuned on MI355X for the BASELINE.json shapes by
#      tools/tune_tiles.py; AVSD_TILE_CACHE=<file> replaces it),
#   2. else a static rule of thumb (_heuristic_tile): the largest tile that still yields >= ~1 workgroup per CU, split-K
#      for the low-resolution layers.
# AVSD_AUTOTUNE=1 / set_autotune(True) switches on the measuring tuner for shapes missing from the table: every
# candidate is timed with HIP events on the caller's real buffers and the winner is cached for the life of the
# process (measure, don't guess) — that is how the table is produced; results then depend on timing noise.
TILE_CANDIDATES = ((4, 1), (6, 1), (7, 1), (9, 1), (3, 1), (11, 1), (12, 1), (13, 1), (14, 1), (17, 1), (19, 1),
                   (20, 1), (24, 1), (25, 1), (30, 1), (31, 1), (38, 1))
# extra (tile, split_k) candidates for GEMMs whose output is too small to fill 256 CUs with big tiles
SPLITK_CANDIDATES = ((6, 2), (6, 4), (6, 8), (9, 2), (9, 4), (4, 2), (4, 4), (4, 8), (7, 2), (7, 4),
                     (20, 2), (20, 4), (24, 2), (24, 4), (24, 8), (25, 2), (25, 4),
                     (30, 2), (30, 4), (30, 8), (31, 2), (31, 4), (31, 8))
# split precision (AVSD_GEMM_X2): the tiles whose doubled LDS stage fits (gemm.hip dispatch_tile_x2)
X2_TILE_CANDIDATES = ((7, 1), (11, 1), (13, 1), (24, 1), (25, 1), (34, 1), (35, 1), (36, 1))
X2_SPLITK_CANDIDATES = ((34, 2), (34, 4), (35, 2), (35, 4), (36, 2), (36, 4), (7, 2), (7, 4), (7, 8), (11, 2), (11, 4), (24, 2), (24, 4), (24, 8), (25, 2), (25, 4), (25, 8))
# 3x3 stride-1 convolutions with the input tile resident in LDS (csrc/conv3r.hip): tile ids 40-49, geometry-dependent
# (avsd_gemm_conv3r_supported); split_k cuts the cin / 64 channel chunks
# 4-wave tiles with a hand-scheduled main loop (csrc/gemm4.hip): 60 = 256x256, 61 = 256x128, 62 = 128x256, 63 = 128x128; PLAIN, K % 64 == 0
# 64 = 128x64, 65 = 64x128, 66 = 64x64, 67 = 128x320 (the N = 320 layers in one column tile); TMIX (cseg % 64 == 0): 61..67
ASM_CANDIDATES = ((60, 1), (61, 1), (62, 1), (63, 1), (64, 1), (65, 1), (66, 1), (67, 1))
ASM_SPLITK_CANDIDATES = ((63, 2), (63, 4), (63, 8), (61, 2), (62, 2), (62, 4), (64, 2), (64, 4), (65, 2), (65, 4), (66, 2), (66, 4),
                         (67, 2), (67, 3), (67, 4), (67, 5), (67, 6))
ASM_TILES = tuple(range(60, 68))
# column-tile width of the LDS-direct tile ids (gemm.hip dispatch_tile / dispatch_tile_x2): a sub-pixel upsample convolution (ups = 2) needs
# whole column tiles per output-pixel parity, cout % BN == 0
SUBPIX_TILES = (4, 6, 9, 11, 13, 17, 20, 24, 25, 30, 38)          # gemm.hip avsd_gemm_dispatch_subpix / _x2_subpix
SUBPIX_X2_TILES = (7, 11, 13, 24, 25, 34, 35)
TILE_BN = {4: 64, 6: 128, 7: 64, 9: 128, 11: 128, 12: 64, 13: 64, 14: 128, 17: 320, 19: 320, 20: 128, 24: 64, 25: 64, 30: 128, 31: 256, 34: 160,
           35: 64, 36: 192, 38: 160}
ASM_X2_CANDIDATES = ((63, 1), (64, 1), (65, 1), (66, 1))        # split precision: 128x128 ... 64x64
ASM_X2_SPLITK_CANDIDATES = ((63, 2), (63, 4), (64, 2), (64, 4), (65, 2), (65, 4), (66, 2), (66, 4), (66, 8))
_ASM_TILES = True          # (module attribute: tools set it to False to time the LDS-direct tiles alone)
# Row bands start their K walk at different tiles (AVSD_GEMM_KROT, include/avsd.h; LDS-direct, resident-convolution and asm tiles) — the
# weights of the low-resolution layers come from HBM inside a step, and a lockstep walk is one chain of round trips.
# AVSD_KROT=0 / set_krot(False): the unrotated walk (every tile of a family then sums K in the same order).
_KROT = os.environ.get("AVSD_KROT", "1") != "0"
_KROT_ASM_ONLY = False          # probe (profiles/r4_krot_ab.txt): rotate the asm tiles only
# ... and only where the weights one XCD walks (its share of W under the column banding) stay in its 4-MB L2 for a whole rotation: bands
# that stand at different K positions re-read W a rotation apart, so a larger W (the 3x3 convolutions at 8 x 8: 29 MB) would come from
# the Infinity Cache once per band instead of once (measured: rotating everything 80.9 steps/s, this rule 81.8, nothing 80.0)
_KROT_MAX_W = 16 << 20


def set_krot(on: bool) -> None:
    global _KROT
    _KROT = bool(on)


# A-resident, N-streaming tile (csrc/nstream.hip): the wide short-K projections whose weights the caller also holds in fragment order.
# Chosen by rule, not by the table: it needs the second weight layout, which only the caller can provide (gemm(w_frag=...)).
NSTREAM_TILE = 70
_NSTREAM = os.environ.get("AVSD_NSTREAM", "1") != "0"


def nstream_supported(M: int, N: int, K: int) -> bool:
    """the shapes tile 70 is built for: all of A's K in LDS (K = 320 / 640), at least 8 N fragments per wave set, and N wide enough that
    streaming it pays (N >= 4 K: the GEGLU projections; a square layer is better off on the tiled kernels)"""
    return K in (320, 640) and N % 256 == 0 and N >= 4 * K and M >= 96


_RASTER_G = 0       # probe knob: rows of the tile blocks an XCD walks (0 = the kernel's default)
CONV3R_TILES = (40, 42, 43, 44, 48)
CONV3R2D_TILES = (51, 52, 53, 54)   # rectangular resident tiles (TH rows x 32 pixels) for images wider than 32 pixels: the VAE decoder, cfg 4
_CONV3R2D_BN = {51: 128, 52: 160, 53: 128, 54: 128}
CONV3R_SPLITS = (1, 2, 4, 5, 8, 10)
_CONV3R = True


_CONV3R_BN = (128, 128, 160, 160, 128, 128, 256, 320, 256, 64)


def conv3r_candidates(hs: int, ws: int, cin: int, M: int, N: int, one_d_only: bool = False):
    """(tile, split_k) pairs of the LDS-resident convolution tiles that fit this image geometry and leave >= 2 chunks per slice"""
    out = []
    for t in CONV3R_TILES:
        bm = _lib.lib().avsd_gemm_conv3r_supported(t, hs, ws, cin)
        if bm <= 0:
            continue
        bn = _CONV3R_BN[t - 40]
        wgs = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        for sk in CONV3R_SPLITS:
            if sk > 1 and (wgs >= 256 or (cin // 64) // sk < 2 or (cin // 64) % sk != 0):
                continue
            out.append((t, sk))
    if not one_d_only:
        for t in CONV3R2D_TILES:
            bm = _lib.lib().avsd_gemm_conv3r2d_supported(t, hs, ws, cin)
            if bm <= 0:
                continue
            wgs = (M // bm) * ((N + _CONV3R2D_BN[t] - 1) // _CONV3R2D_BN[t])
            for sk in CONV3R_SPLITS:
                if sk > 1 and (wgs >= 256 or (cin // 64) // sk < 2 or (cin // 64) % sk != 0):
                    continue
                out.append((t, sk))
    return tuple(out)


def _heuristic_conv3r(cands, M: int, N: int):
    """static pick among the admissible resident tiles when the table has no entry: the candidate that wastes the fewest
    padded columns, then the one closest to one workgroup per CU"""
    def cost(c):
        t, sk = c
        bn = _CONV3R_BN[t - 40] if t < 50 else _CONV3R2D_BN[t]
        bm = 256 if t in (40, 41, 42, 43, 48, 51, 52, 53) else 128
        wg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * sk
        pad = ((N + bn - 1) // bn) * bn / N
        return (round(pad, 2), abs(wg - 256) if wg < 256 else (wg - 256) // 4, sk)
    return min(cands, key=cost)


_TILE_CACHE: dict = {}
_AUTOTUNE = os.environ.get("AVSD_AUTOTUNE", "0") == "1"


def set_autotune(on: bool) -> None:
    global _AUTOTUNE
    _AUTOTUNE = bool(on)


def tile_cache() -> dict:
    return _TILE_CACHE


def save_tile_cache(path: str) -> None:
    """Writes the tuned (shape -> tile, split_k) table as JSON; AVSD_TILE_CACHE=<path> loads it at import, so a
    profiling run (rocprofv3 --pmc serialises every dispatch) can skip the tuning launches."""
    import json

    with open(path, "w") as f:
        json.dump([[list(k), list(v)] for k, v in _TILE_CACHE.items()], f)


def load_tile_cache(path: str) -> int:
    import json

    with open(path) as f:
        for k, v in json.load(f):
            _TILE_CACHE[tuple(k)] = tuple(v)
    return len(_TILE_CACHE)


DEFAULT_TILE_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiles_gfx950.json")
if os.environ.get("AVSD_TILE_CACHE"):
    if os.path.isfile(os.environ["AVSD_TILE_CACHE"]):
        load_tile_cache(os.environ["AVSD_TILE_CACHE"])
elif os.path.isfile(DEFAULT_TILE_TABLE):
    load_tile_cache(DEFAULT_TILE_TABLE)


def _heuristic_tile(M: int, N: int, K: int, geglu: bool, splitk_ok: bool):
    """Static (tile, split_k) for shapes the table does not hold.  Per-tile efficiency grows with the tile (the
    global->LDS path delivers (BM + BN) / (BM * BN) bytes per MFMA flop), but a launch wants >= ~1 workgroup per CU."""
    def tiles(bm, bn):
        return ((M + bm - 1) // bm) * ((N + bn - 1) // bn)

    nk = (K + 63) // 64
    # measured (profiles/r2_probes.md, tools/tile_probe.py): 256x128 wins once it fills the chip — with loader waves and a
    # 3-deep ring for long K, the 2-stage form for short K; 256x256 only pays at K >= 4096 and is left to the tuned table;
    # N = 320 layers lose 17 % of a 128-wide tile to padding and take 128x64
    if N >= 512 and tiles(256, 128) >= 224:
        return (20 if nk >= 16 else 14), 1
    if N >= 512 and tiles(128, 128) >= 224:
        return (30 if nk >= 8 else 11), 1
    if tiles(128, 64) >= 224:
        return (24 if nk >= 16 else 12), 1
    t64 = tiles(64, 64)
    if t64 >= 160 or not splitk_ok or geglu or nk < 8:
        return (25 if nk >= 8 else 13), 1
    sk = 1
    while sk < 8 and t64 * sk < 224 and nk // (2 * sk) >= 4:
        sk *= 2
    return 25, sk


def _heuristic_tile_x2(M: int, N: int, K: int, geglu: bool, splitk_ok: bool):
    """the same rule over the split-precision tile set: 128x128 (2-stage), 128x64, 64x64, split-K for the small outputs"""
    def tiles(bm, bn):
        return ((M + bm - 1) // bm) * ((N + bn - 1) // bn)

    nk = (K + 63) // 64
    if N % 160 == 0 and N <= 640 and tiles(128, 160) >= 224:
        return 34, 1
    if N >= 128 and tiles(128, 128) >= 224:
        return 11, 1
    if tiles(128, 64) >= 224:
        return 24, 1
    t64 = tiles(64, 64)
    if t64 >= 160 or not splitk_ok or geglu or nk < 8:
        return (25 if nk >= 8 else 13), 1
    sk = 1
    while sk < 8 and t64 * sk < 224 and nk // (2 * sk) >= 4:
        sk *= 2
    return 25, sk


_TUNE_COLD = True
_FLUSH = None


def _evict_weights(warm):
    """Reproduce the cache state a launch meets inside a denoising step: every weight is streamed from HBM once per
    step (2.3 GB of them pass between two uses), while the activations were written by the previous kernel.  Write a
    buffer larger than L2 + Infinity Cache, then read the activations back in."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(384 << 20, dtype=torch.uint8, device="cuda")
    _FLUSH.zero_()
    for t in warm:
        if t is not None:
            t.view(torch.int16 if t.element_size() == 2 else torch.int32).max()


def _time_hot(launch, cand, reps=8):
    """GPU time per launch of `reps` back-to-back launches replayed from a captured graph: eager launches through
    Python/ctypes cannot be issued faster than one per ~11 us, which would hide every difference between candidates
    for the many GEMMs of this UNet that run 10-20 us."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            launch(*cand)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (2 * reps)


def _time_cold(launch, cand, warm, reps=7):
    ms = []
    for _ in range(reps):
        _evict_weights(warm)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch(*cand)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]


_CHALLENGE = os.environ.get("AVSD_TUNE_CHALLENGE", "0") == "1"     # tools/tune_tiles.py --challenge: re-time table entries against new tiles
_CHALLENGE_TILES = tuple(int(v) for v in os.environ.get("AVSD_TUNE_CHALLENGE_TILES", "").split(",") if v)      # ... against these tile ids only
_CHALLENGED: set = set()
RECORD_KEYS = None
CHALLENGE_LOG: list = []


def _challenge(key, launch, challengers, warm):
    """the table's entry for `key` against `challengers` (new tile ids): replaced when one of them is >= 3 % faster hot (two
    interleaved rounds, best-of) and not slower against cold weights"""
    inc = _TILE_CACHE[key]
    _CHALLENGED.add(key)
    if _CHALLENGE_TILES:
        challengers = tuple(c for c in challengers if c[0] in _CHALLENGE_TILES)
        if not challengers:
            return
    times = {}
    for rnd in range(2):
        for cand in (inc,) + tuple(c for c in challengers if c != inc):
            if rnd == 0:
                launch(*cand)
            times[cand] = min(_time_hot(launch, cand), times.get(cand, float("inf")))
    best = min(times, key=times.get)
    if best != inc and times[best] < 0.97 * times[inc] and (warm is None or _time_cold(launch, best, warm) < _time_cold(launch, inc, warm)):
        CHALLENGE_LOG.append((key, inc, best, times[inc] * 1e3, times[best] * 1e3))
        _TILE_CACHE[key] = best


def _pick_tile(key, launch, candidates=TILE_CANDIDATES, warm=None, challengers=()):
    """-> (tile, split_k).  Two passes: every candidate is timed back to back on a hot L2 (two interleaved rounds,
    best-of: robust to clock ramp / noise); the four fastest are then re-timed launch by launch against cold weights
    and warm activations (`warm`: the activation tensors; median of 7) — the state a launch meets inside a denoising
    step — and the winner of that pass is cached.  AVSD_TUNE_COLD=0 keeps the first pass only."""
    t = _TILE_CACHE.get(key)
    if t is not None:
        if _CHALLENGE and challengers and key not in _CHALLENGED and _AUTOTUNE and _TIMER is None and not torch.cuda.is_current_stream_capturing():
            _challenge(key, launch, challengers, warm)
            return _TILE_CACHE[key]
        return t
    if not _AUTOTUNE or _TIMER is not None or torch.cuda.is_current_stream_capturing():
        return None
    candidates = tuple(candidates) + tuple(challengers)
    times = {}
    for rnd in range(2):
        for cand in candidates:
            if rnd == 0:
                launch(*cand)              # warm (also sets the kernel's LDS attribute)
            times[cand] = min(_time_hot(launch, cand), times.get(cand, float("inf")))
    ranked = sorted(times, key=times.get)
    best = ranked[0]
    if _TUNE_COLD and warm is not None and len(ranked) > 1:
        finalists = ranked[:4]
        cold = {c: _time_cold(launch, c, warm) for c in finalists}
        best = min(cold, key=cold.get)
    _TILE_CACHE[key] = best
    return best


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ---- split-precision storage (asva_amd/precision.py): a 16-bit tensor is a view into the FIRST half of its storage, its rest
# plane lies at the same offset in the second half ------------------------------------------------------------------------
def alloc16(shape, device) -> torch.Tensor:
    """uninitialised 16-bit activation tensor; in split mode the main plane of a twin allocation"""
    if not P.SPLIT:
        return torch.empty(shape, dtype=P.ACT, device=device)
    n = 1
    for d in shape:
        n *= int(d)
    npad = (n + 7) // 8 * 8
    return torch.empty((2, npad), dtype=P.ACT, device=device)[0, :n].view(tuple(shape))


def _lo(t: Optional[torch.Tensor]) -> int:
    """element offset from `t` to its rest plane (0 outside split mode / for None)"""
    if t is None or not P.SPLIT:
        return 0
    if t.dtype != P.ACT:
        raise TypeError(f"split precision: expected a {P.ACT} tensor, got {t.dtype}")
    nb = t.untyped_storage().nbytes()
    last = sum((s - 1) * st for s, st in zip(t.shape, t.stride()))
    if nb % 32 or (t.storage_offset() + last + 1) * 2 > nb // 2:
        raise ValueError("split precision: tensor is not a view into the first half of a twin allocation "
                         "(allocate with ops.alloc16 / ops.to_act / weights.to_act)")
    return nb // 4


def to_act(x: torch.Tensor) -> torch.Tensor:
    """any float device tensor -> the 16-bit storage type (one launch in split mode: main and rest planes)"""
    if not P.SPLIT:
        return x.to(P.ACT).contiguous()
    x = x.to(F32).contiguous()
    out = alloc16(tuple(x.shape), x.device)
    check(_lib.lib().avsd_split_f32(_p(x), _p(out), _lo(out), x.numel(), _stream()), "avsd_split_f32")
    return out


def from_act(t: torch.Tensor) -> torch.Tensor:
    """f32 values of a 16-bit activation (main + rest in split mode); for tests and host-side checks"""
    if not P.SPLIT:
        return t.float()
    half = _lo(t)
    return t.float() + torch.as_strided(t, t.shape, t.stride(), t.storage_offset() + half).float()


def alloc_planes(shape, device):
    """(main, rest): two 16-bit planes of one allocation, whatever the process-wide mode — operands of an explicit three-pass
    product (gemm(..., a_rest=, w_rest=)) under the per-layer precision plan (asva_amd/precision.py)"""
    n = 1
    for d in shape:
        n *= int(d)
    npad = (n + 7) // 8 * 8
    buf = torch.empty((2, npad), dtype=P.ACT, device=device)
    return buf[0, :n].view(tuple(shape)), buf[1, :n].view(tuple(shape))


def split_planes(x: torch.Tensor):
    """f32 device tensor -> (main, rest) 16-bit planes: main = round16(x), rest = round16(x - main) (one launch)"""
    x = x.contiguous()
    _req(x, F32, "x")
    main, rest = alloc_planes(tuple(x.shape), x.device)
    check(_lib.lib().avsd_split_f32(_p(x), _p(main), _rest_off(main, rest), x.numel(), _stream()), "avsd_split_f32")
    return main, rest


def _rest_off(main: torch.Tensor, rest: Optional[torch.Tensor]) -> int:
    """element offset from `main` to its explicit rest plane (same shape and strides)"""
    if rest is None:
        return 0
    if rest.dtype != main.dtype or rest.shape != main.shape or rest.stride() != main.stride():
        raise ValueError("rest plane must match its main plane in dtype, shape and strides")
    off = rest.data_ptr() - main.data_ptr()
    if off == 0 or off % 16:
        raise ValueError("rest plane must be a distinct tensor, a multiple of 8 elements away from its main plane")
    return off // 2


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a device tensor (libavsd_hip.so has no host path)")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.dim() == 2 else t.stride(-2)


def gemm(
    a: torch.Tensor,
    w: torch.Tensor,
    *,
    n: Optional[int] = None,
    k: Optional[int] = None,
    a2: Optional[torch.Tensor] = None,
    bias: Optional[torch.Tensor] = None,
    rowvec: Optional[torch.Tensor] = None,
    rows_per_vec: int = 0,
    res1: Optional[torch.Tensor] = None,
    res2: Optional[torch.Tensor] = None,
    alpha: float = 1.0,
    geglu: bool = False,
    gelu: bool = False,
    rowstats: Optional[torch.Tensor] = None,   # out: f32 [M, N/32, 2] (sum, sumsq) of the rounded outputs per 32 columns
    ln: Optional[tuple] = None,                # (stats [rows, K/32, 2] f32, colsum [N] f32, eps): LayerNorm(A) folded in
    stats_pos: Optional[tuple] = None,         # (pos [frames, N] f32, hw, frames): rowstats of out + pos[frame of the row]
    ln_pos: Optional[tuple] = None,            # (pos . W'^T [frames, N] f32, hw, frames) with ln: LayerNorm(A + pos[frame]) folded in
    out_f32: bool = False,
    out: Optional[torch.Tensor] = None,
    master: Optional[torch.Tensor] = None,     # out: f32 [M, N] un-rounded copy of the result (f32 residual stream)
    mode: int = PLAIN,
    tmix: Optional[tuple] = None,      # (hw, frames)
    conv: Optional[tuple] = None,      # (n_img, hs, ws, stride, ups[, pad])
    m: Optional[int] = None,
    tile: int = 0,
    split_k: int = 1,
    a_rest: Optional[torch.Tensor] = None,     # explicit rest planes of a / a2 / w: THIS product runs as three MFMA passes (AVSD_GEMM_X2)
    a2_rest: Optional[torch.Tensor] = None,    # whatever the process-wide mode (per-layer precision plan); f32 output and f32 residuals
    w_rest: Optional[torch.Tensor] = None,
    out_rest: Optional[torch.Tensor] = None,   # out: the rest plane round16(v - out) of a 16-bit output (AVSD_GEMM_OUT_REST; with w_rest: the second
                                               # output plane of the three-pass product) — what split_planes(master) would make in another launch
    w_frag: Optional[torch.Tensor] = None,     # the same weights in MFMA-fragment order (weights.pack_frag): lets the A-resident N-streaming
                                               # tile (70, csrc/nstream.hip) take the product where it applies (nstream_supported)
) -> torch.Tensor:
    """out = epilogue(alpha * A' . W^T); see avsd_gemm_bf16 in include/avsd.h."""
    _req(a, P.ACT, "a")
    _req(w, P.ACT, "w")
    planes = w_rest is not None
    if planes:
        if P.SPLIT:
            raise ValueError("gemm: explicit rest planes are for the non-split modes (split precision carries them implicitly)")
        if a_rest is None or (a2 is not None) != (a2_rest is not None) or rowstats is not None or ln is not None:
            raise ValueError("gemm: a three-pass product needs a_rest and w_rest (a2_rest with a2), no rowstats / LayerNorm fold")
        if out_f32 == (out_rest is not None) or (out_f32 and master is not None):
            raise ValueError("gemm: a three-pass product writes f32 (out_f32) or a (main, rest) pair of planes (out=, out_rest=; optional f32 master)")
        if any(r is not None and r.dtype != F32 for r in (res1, res2)):
            raise ValueError("gemm: a three-pass product reads f32 residuals")
    elif a_rest is not None or a2_rest is not None:
        raise ValueError("gemm: a_rest / a2_rest without w_rest")
    if out_rest is not None and (P.SPLIT or out_f32 or geglu or out is None):
        raise ValueError("gemm: out_rest goes with an explicit 16-bit `out` outside split precision (no GEGLU)")
    x2 = P.SPLIT or planes
    if x2 and mode == PLAIN and a2 is not None and a.shape[1] % 64 != 0:
        # the LDS-direct loader switches source buffers per 64-wide K tile and the register-staged tiles have no split form:
        # a concat split inside a K tile (tiny test networks: 160-channel skips) runs as two launches, the first leaving its
        # un-rounded f32 partial for the second's epilogue — the same sum
        if res1 is not None and res2 is not None:
            raise ValueError("gemm: split precision with an unaligned two-source A supports one residual")
        if gelu or geglu or rowstats is not None or ln is not None or stats_pos is not None or ln_pos is not None or (master is not None and not planes) or n is not None \
                or k is not None or m is not None or tile or split_k != 1:
            # (GELU would be applied to the second partial alone; the other options are not forwarded by this two-launch form)
            raise ValueError("gemm: split precision with an unaligned two-source A takes bias / rowvec / one residual / alpha only")
        k1 = a.shape[1]
        part = gemm(a, w[:, :k1], alpha=alpha, out_f32=True, a_rest=a_rest, w_rest=None if w_rest is None else w_rest[:, :k1])
        return gemm(a2, w[:, k1:], bias=bias, rowvec=rowvec, rows_per_vec=rows_per_vec, res1=part, res2=res1 if res1 is not None else res2,
                    alpha=alpha, out_f32=out_f32, out=out, out_rest=out_rest, master=master, a_rest=a2_rest, w_rest=None if w_rest is None else w_rest[:, k1:])
    d = GemmDesc()
    N = w.shape[0] if n is None else n
    lda = _ld(a)
    if mode == PLAIN:
        M = a.shape[0] if m is None else m
        K1 = a.shape[1]
        K = K1 + (a2.shape[1] if a2 is not None else 0)
        if a2 is not None:
            _req(a2, P.ACT, "a2")
            d.A2, d.lda2, d.k_split = _p(a2), _ld(a2), K1
        else:
            d.k_split = K
        if k is not None:
            K = k
    elif mode == TMIX:
        hw, frames = tmix
        M = a.shape[0]
        cseg = a.shape[1]
        K = 3 * cseg
        d.hw, d.frames, d.cseg = hw, frames, cseg
    elif mode == CONV3:
        n_img, hs, ws, stride, ups = conv[:5]
        pad = conv[5] if len(conv) > 5 else 1
        cin = a.shape[1]
        if a2 is not None:              # two-source input: only the LDS-resident tiles read it (conv3r.hip)
            _req(a2, P.ACT, "a2")
            d.A2, d.lda2, d.k_split = _p(a2), _ld(a2), cin
            cin += a2.shape[1]
        if ups == 2:
            # nearest-2x upsample + 3x3 convolution as four per-parity 2x2 convolutions on the original image (weights.subpixel_conv3x3):
            # GEMM rows = INPUT pixels, columns = (parity, cout), K = 4 taps x cin; the epilogue scatters row m = (n, y, x), column
            # (dy, dx, co) to output pixel (n, 2 y + dy, 2 x + dx) of the [n_img * 2 hs * 2 ws, cout] result
            if stride != 1 or pad != 1 or a2 is not None or N % 4 or w.shape[1] != 4 * cin or cin % 64 or (N // 4) % 64:
                raise ValueError("gemm: ups = 2 (sub-pixel upsample convolution) needs stride 1, pad 1, one source, w [4 cout, 4 cin], cin % 64 == 0, cout % 64 == 0")
            if geglu or gelu or rowstats is not None or ln is not None or res1 is not None or res2 is not None or rowvec is not None:
                raise ValueError("gemm: ups = 2 takes bias / master / out_rest only")
            ho, wo = hs, ws
            M = n_img * hs * ws
            K = 4 * cin
        else:
            hin, win = hs << ups, ws << ups
            ho, wo = (hin + 2 - 3) // stride + 1, (win + 2 - 3) // stride + 1
            M = n_img * ho * wo
            K = 9 * cin
        d.hs, d.ws, d.ho, d.wo, d.cin, d.stride, d.ups, d.pad = hs, ws, ho, wo, cin, stride, ups, pad
    else:
        raise ValueError(f"unknown gemm mode {mode}")
    n_out = N // 2 if geglu else N
    m_out = M
    if mode == CONV3 and d.ups == 2:
        m_out, n_out = 4 * M, N // 4
    if out is None:
        out = torch.empty((m_out, n_out), dtype=F32, device=a.device) if out_f32 else alloc16((m_out, n_out), a.device)
    else:
        _req(out, F32 if out_f32 else P.ACT, "out")
        if tuple(out.shape) != (m_out, n_out) and mode == CONV3 and d.ups == 2:
            raise ValueError(f"gemm: ups = 2 writes [{m_out}, {n_out}], got {tuple(out.shape)}")
    d.A, d.W, d.out = _p(a), _p(w), _p(out)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldw, d.ldc = lda, _ld(w), _ld(out)
    if bias is not None:
        _req(bias, F32, "bias")
        d.bias = _p(bias)
    if rowvec is not None:
        _req(rowvec, F32, "rowvec")
        d.rowvec, d.rows_per_vec, d.ldv = _p(rowvec), rows_per_vec, _ld(rowvec)
    d.flags = (GEGLU if geglu else 0) | (OUT_F32 if out_f32 else 0) | (GELU if gelu else 0)
    if res1 is not None:            # residuals: 16-bit, or f32 (the f32 residual stream)
        _req(res1, F32 if res1.dtype == F32 else P.ACT, "res1")
        d.res1, d.ldr1 = _p(res1), _ld(res1)
        d.flags |= RES1_F32 if res1.dtype == F32 else 0
    if res2 is not None:
        _req(res2, F32 if res2.dtype == F32 else P.ACT, "res2")
        d.res2, d.ldr2 = _p(res2), _ld(res2)
        d.flags |= RES2_F32 if res2.dtype == F32 else 0
    if master is not None:
        _req(master, F32, "master")
        if master.shape != (m_out, n_out) or geglu:
            raise ValueError("gemm: master must be f32 [M, N] (not available with GEGLU)")
        d.out_master, d.ldm = _p(master), _ld(master)
    d.alpha = alpha
    d.mode = mode
    if rowstats is not None:
        _req(rowstats, F32, "rowstats")
        if not rowstats.is_contiguous() or rowstats.numel() != M * (N // 32) * 2:
            raise ValueError("gemm: rowstats must be contiguous f32 [M, N/32, 2]")
        d.rowstats = _p(rowstats)
        d.flags |= ROWSTATS
    # A-resident N-streaming tile (csrc/nstream.hip): decided HERE, once — the caller passes the fragment-ordered copy and the raw K / 32
    # statistics; when the tile does not apply (planes / split mode, M too small, a non-contiguous copy) the product runs on the table's
    # tiles, and the GEGLU projection's statistics are pre-folded (avsd_ln_fold) so that they are not re-folded in every column tile
    x2_now = P.SPLIT or planes
    use_nstream = False
    if tile == NSTREAM_TILE or (tile == 0 and _NSTREAM and w_frag is not None):
        use_nstream = (w_frag is not None and not x2_now and mode == PLAIN and a2 is None and split_k == 1 and K in (320, 640) and N % 32 == 0
                       and (tile == NSTREAM_TILE or nstream_supported(M, N, K))
                       and stats_pos is None and ln_pos is None and w_frag.dtype == P.ACT and w_frag.is_contiguous() and w_frag.numel() == N * K)
        if tile == NSTREAM_TILE and not use_nstream:
            raise ValueError("gemm: tile 70 needs w_frag (weights.pack_frag) and a PLAIN single-source product with K = 320 / 640, N % 32 == 0")
    if w_frag is not None and not use_nstream and ln is not None and geglu and ln[0].shape[-2] != 1 and ln_pos is None:
        ln = (ln_fold(ln[0]), ln[1], ln[2])
    if ln is not None:
        st, colsum, eps = ln
        _req(st, F32, "ln stats")
        _req(colsum, F32, "ln colsum")
        if not st.is_contiguous() or st.shape[-2:] not in ((K // 32, 2), (1, 2)) or colsum.numel() != N:
            raise ValueError("gemm: ln = (stats [rows, K/32, 2], colsum [N], eps)")
        d.ln_stats, d.ln_colsum, d.ln_nblk, d.ln_eps = _p(st), _p(colsum), st.shape[-2], float(eps)
        d.flags |= LNFUSE
    for name, arg in (("stats_pos", stats_pos), ("ln_pos", ln_pos)):
        if arg is None:
            continue
        tbl, hw_, fr_ = arg
        _req(tbl, F32, name)
        if P.SPLIT or not tbl.is_contiguous() or tbl.shape != (fr_, N) or hw_ <= 0 or (rowstats is None if name == "stats_pos" else ln is None):
            raise ValueError(f"gemm: {name} = (f32 [frames, N], hw, frames) goes with " + ("rowstats" if name == "stats_pos" else "ln") +
                             "; not in split precision")
        if name == "stats_pos":
            d.stats_pos = _p(tbl)
        else:
            d.ln_rowvec = _p(tbl)
        d.pos_hw, d.pos_frames = int(hw_), int(fr_)
    # XCD banding: the 8 L2s are not shared, so whichever operand is NOT banded is fetched by all 8 of them
    a_bytes = M * (K // (4 if d.ups == 2 else 9) if mode == CONV3 else K // 3 if mode == TMIX else K)
    if _XCD_MODE == "n" or (_XCD_MODE == "auto" and N * K > a_bytes):
        d.flags |= XCD_N
    d.batch = 1
    d.raster_g = _RASTER_G
    ws = None
    key_flags, key_master = None, int(master is not None)
    if planes:
        d.flags |= X2
        d.a_lo, d.a2_lo, d.w_lo = _rest_off(a, a_rest), (_rest_off(a2, a2_rest) if a2 is not None else 0), _rest_off(w, w_rest)
        if out_rest is not None:
            # (main, rest) planes (+ f32 master) instead of f32: the same main loop — the tile table keeps ONE entry per three-pass shape
            d.out_lo = _rest_off(out, out_rest)
            key_flags, key_master = d.flags | OUT_F32, 0
    elif out_rest is not None:
        key_flags = d.flags                 # (the rest-plane store does not change which tile is fastest: not part of the table key)
        d.flags |= OUT_REST
        d.out_lo = _rest_off(out, out_rest)
    elif P.SPLIT:
        if master is not None:
            raise ValueError("gemm: split precision has no f32 master (the planes carry 16 bits)")
        d.flags |= X2
        d.a_lo, d.a2_lo, d.w_lo = _lo(a), _lo(a2), _lo(w)
        d.out_lo = 0 if out_f32 else _lo(out)
        d.res1_lo = _lo(res1) if (res1 is not None and res1.dtype != F32) else 0
        d.res2_lo = _lo(res2) if (res2 is not None and res2.dtype != F32) else 0

    if use_nstream:
        tile = NSTREAM_TILE
        d.W = _p(w_frag)
        d.flags |= W_FRAG

    def _set(t, sk):
        nonlocal ws
        d.tile, d.split_k = t, sk
        krot = _KROT and (t in ASM_TILES or not _KROT_ASM_ONLY) and 2 * N * K <= _KROT_MAX_W
        d.flags = (d.flags | KROT) if krot else (d.flags & ~KROT)          # (set after the table key was formed: not part of it)
        if sk > 1:
            if ws is None or ws.numel() < sk * M * N:
                ws = torch.empty((sk * M * N,), dtype=F32, device=a.device)
            d.splitk_ws = _p(ws)

    if tile == 0:
        def _launch(t, sk):
            _set(t, sk)
            check(_lib.lib().avsd_gemm_bf16(C.byref(d), _stream()), "avsd_gemm_bf16")

        cands = X2_TILE_CANDIDATES if x2 else TILE_CANDIDATES
        nk = (K + 63) // 64
        two_src_unaligned = a2 is not None and (a.shape[1] % 64 != 0)     # C falls back to register-staged tiles
        if not geglu and not two_src_unaligned and ((M + 127) // 128) * ((N + 127) // 128) < 256 and nk >= 16:
            cands = cands + tuple(c for c in (X2_SPLITK_CANDIDATES if x2 else SPLITK_CANDIDATES) if nk // c[1] >= 4)
        splitk_ok = not geglu and not two_src_unaligned
        if mode == CONV3 and d.ups == 2:       # the tiles built for this loader whose column tiles do not straddle two output-pixel parities
            ok_tiles = SUBPIX_X2_TILES if x2 else SUBPIX_TILES
            cands = tuple(c for c in cands if c[0] in ok_tiles and (N // 4) % TILE_BN[c[0]] == 0)
        two_src_conv = mode == CONV3 and a2 is not None           # only the LDS-resident tiles read a second source
        if two_src_conv:
            cands = conv3r_candidates(d.hs, d.ws, d.cin, M, N)
            if not cands:
                raise ValueError("gemm: no LDS-resident convolution tile takes this two-source geometry")
        elif (_CONV3R and mode == CONV3 and not x2 and d.stride == 1 and d.ups == 0 and d.pad == 1 and ln is None and out_rest is None):
            cands = cands + conv3r_candidates(d.hs, d.ws, d.cin, M, N)
        # 16-bit convolutions are keyed by the image geometry too: which LDS-resident tiles apply depends on (hs, ws)
        key = (mode, M, N, K, d.flags if key_flags is None else key_flags, d.stride, d.ups, d.pad, key_master)
        if mode == CONV3 and not x2:
            key = key + (d.hs, d.ws)
        if two_src_conv:
            key = key + ("a2", d.k_split)                         # a table entry of the one-source shape may name a tile that never reads A2
        asm = ()
        if (_ASM_TILES and a2 is None and K % 64 == 0 and M * lda < (1 << 29) and N * _ld(w) < (1 << 29) and
                (mode == PLAIN or (mode == TMIX and d.cseg % 64 == 0 and ln is None))):
            asm = ASM_X2_CANDIDATES if x2 else tuple(c for c in ASM_CANDIDATES if mode == PLAIN or c[0] != 60)
            if splitk_ok and ((M + 127) // 128) * ((N + 127) // 128) < 256 and nk >= 8:
                asm = asm + tuple(c for c in (ASM_X2_SPLITK_CANDIDATES if x2 else ASM_SPLITK_CANDIDATES)
                                  if nk // c[1] >= 4 and (c[1] - 1) * -(-nk // c[1]) < nk)      # (no empty K slice: the asm tiles refuse it)
        if RECORD_KEYS is not None:          # tools/tune_in_step.py: which table keys a forward uses, how often, and what could run them
            rec = RECORD_KEYS.setdefault(key, {"n": 0, "cands": tuple(dict.fromkeys(tuple(cands) + tuple(asm))), "flops": 2.0 * M * N * K})
            rec["n"] += 1
        picked = _pick_tile(key, _launch, cands, warm=(a, a2, res1, res2), challengers=asm)
        if picked is not None and picked[0] in ASM_TILES and not asm:        # (a two-source call shares the key of the one-source shape)
            picked = None
        if picked is not None and picked[0] in CONV3R_TILES and not ((_CONV3R or two_src_conv) and _lib.lib().avsd_gemm_conv3r_supported(picked[0], d.hs, d.ws, d.cin)):
            picked = None
        if picked is not None and picked[0] in CONV3R2D_TILES and not ((_CONV3R or two_src_conv) and _lib.lib().avsd_gemm_conv3r2d_supported(picked[0], d.hs, d.ws, d.cin)):
            picked = None
        if picked is not None and two_src_conv and picked not in cands:
            picked = None
        if picked is not None and out_rest is not None and not planes and (picked[0] in CONV3R_TILES or picked[0] in CONV3R2D_TILES):
            picked = None            # (the LDS-resident convolution tiles have no rest-plane store)
        heur = _heuristic_tile_x2 if x2 else _heuristic_tile
        if picked is None and two_src_conv:
            picked = _heuristic_conv3r(cands, M, N)
        tile, split_k = picked if picked is not None else heur(M, N, K, geglu, splitk_ok)
        if mode == CONV3 and d.ups == 2 and (tile not in (SUBPIX_X2_TILES if x2 else SUBPIX_TILES) or (N // 4) % TILE_BN[tile]):
            tile = {14: 20, 12: 24}.get(tile, 11)      # (the rule named a tile this loader is not built for: its nearest built relative)
    _set(tile, split_k)
    ev = _TIMER.start() if _TIMER is not None else None
    check(_lib.lib().avsd_gemm_bf16(C.byref(d), _stream()), "avsd_gemm_bf16")
    if ev is not None:
        fam = ("gemm_plain", "gemm_tmix", "gemm_conv3")[mode]
        dc = GemmDesc.from_buffer_copy(d)
        _TIMER.add_replay(fam, lambda dc=dc: check(_lib.lib().avsd_gemm_bf16(C.byref(dc), _stream()), "avsd_gemm_bf16"),
                          (a, a2, w, out, bias, rowvec, res1, res2, ws, rowstats, ln, master, stats_pos, ln_pos, a_rest, a2_rest, w_rest, w_frag, out_rest))
        subpix = mode == CONV3 and d.ups == 2
        _TIMER.stop(ev, fam, 2.0 * M * N * K * (3 if x2 else 1), 2.0 * M * K * (1.0 / (4 if subpix else 9) if mode == CONV3 else 1.0 / 3 if mode == TMIX else 1.0)
                    + 2.0 * N * K + _nbytes(out, res1, res2), flops_once=2.0 * M * N * K, flops_ref=2.0 * M * N * K * (2.25 if subpix else 1.0))
    return out


def gemm_f32(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """exact-f32 yardstick: out = a . w^T + bias on v_mfma_f32_32x32x2_f32 (avsd_gemm_f32); a [M, K], w [N, K] f32"""
    _req(a, F32, "a")
    _req(w, F32, "w")
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=F32, device=a.device)
    check(_lib.lib().avsd_gemm_f32(_p(a), _ld(a), _p(w), _ld(w), _p(bias), _p(out), _ld(out), M, N, K, _stream()), "avsd_gemm_f32")
    return out


def gemm_batched(a: torch.Tensor, w: torch.Tensor, *, alpha: float = 1.0, out_f32: bool = False,
                 bias: Optional[torch.Tensor] = None, tile: int = 0, ln: Optional[tuple] = None) -> torch.Tensor:
    """out[b] = alpha * a[b] . w[b]^T for 3-D a [B, M, K], w [B, N, K] (VAE mid-block attention)."""
    _req(a, P.ACT, "a")
    _req(w, P.ACT, "w")
    B, M, K = a.shape
    N = w.shape[1]
    out = torch.empty((B, M, N), dtype=F32, device=a.device) if out_f32 else alloc16((B, M, N), a.device)
    d = GemmDesc()
    d.A, d.W, d.out = _p(a), _p(w), _p(out)
    d.M, d.N, d.K, d.k_split = M, N, K, K
    d.lda, d.ldw, d.ldc = a.stride(1), w.stride(1), out.stride(1)
    d.alpha, d.mode, d.flags, d.batch, d.tile = alpha, PLAIN, (OUT_F32 if out_f32 else 0), B, tile
    d.batch_stride_a, d.batch_stride_w, d.batch_stride_out = a.stride(0), w.stride(0), out.stride(0)
    if bias is not None:
        _req(bias, F32, "bias")
        d.bias = _p(bias)
    if ln is not None:          # statistics are indexed by the row of the full tensor the batches are views of
        st, colsum, eps = ln
        _req(st, F32, "ln stats")
        _req(colsum, F32, "ln colsum")
        d.ln_stats, d.ln_colsum, d.ln_nblk, d.ln_eps = _p(st), _p(colsum), K // 32, float(eps)
        d.flags |= LNFUSE
    if P.SPLIT:
        d.flags |= X2
        d.a_lo, d.w_lo, d.out_lo = _lo(a), _lo(w), (0 if out_f32 else _lo(out))
    if tile == 0:
        def _launch(t):
            d.tile = t
            check(_lib.lib().avsd_gemm_bf16(C.byref(d), _stream()), "avsd_gemm_bf16(batched)")

        picked = _pick_tile(("batched", B, M, N, K, d.flags), lambda t, sk: _launch(t), X2_TILE_CANDIDATES if P.SPLIT else TILE_CANDIDATES)
        d.tile = picked[0] if picked is not None else (_heuristic_tile_x2 if P.SPLIT else _heuristic_tile)(B * M, N, K, False, False)[0]
    dc = GemmDesc.from_buffer_copy(d) if _TIMER is not None else d
    _timed("gemm_plain", 2.0 * B * M * N * K * (3 if P.SPLIT else 1), 2.0 * B * (M * K + N * K) + _nbytes(out), (a, w, out, bias, ln),
           lambda: check(_lib.lib().avsd_gemm_bf16(C.byref(dc), _stream()), "avsd_gemm_bf16(batched)"))
    return out


def linear_small_m(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, act_in: bool = False,
                   act_out: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x, F32, "x")
    _req(w, P.ACT, "w")
    M, K = x.shape
    N = w.shape[0]
    if not x.is_contiguous():
        raise ValueError("linear_small_m: x must be contiguous")
    if out is None:
        out = torch.empty((M, N), dtype=F32, device=x.device)
    if P.SPLIT:
        check(_lib.lib().avsd_linear_small_m_x2(_p(x), _p(w), _lo(w), _p(bias), _p(out), M, N, K, _ld(w), int(act_in), int(act_out),
                                                _stream()), "avsd_linear_small_m_x2")
        return out
    check(_lib.lib().avsd_linear_small_m(_p(x), _p(w), _p(bias), _p(out), M, N, K, _ld(w), int(act_in), int(act_out),
                                         _stream()), "avsd_linear_small_m")
    return out


# one-launch GroupNorm for small batches (csrc/groupnorm_fused.hip); AVSD_GN_FUSED=0: always the pair
_GN_FUSED = True

def groupnorm(x1: torch.Tensor, x2: Optional[torch.Tensor], nb: int, rows_per_batch: int, groups: int,
              gamma: torch.Tensor, beta: torch.Tensor, eps: float, act: bool,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm(+SiLU) of the channel concat [x1 | x2]; statistics pooled over each run of
    `rows_per_batch` rows.  One launch for small batches (avsd_groupnorm_fused: the ResBlock norms at 4 x 4, the per-frame
    norms up to 16 x 16), else two: partial sums, then reduce + apply."""
    _req(x1, P.ACT, "x1")
    c1 = x1.shape[1]
    c2 = 0
    if x2 is not None:
        _req(x2, P.ACT, "x2")
        c2 = x2.shape[1]
    _req(gamma, F32, "gamma")
    _req(beta, F32, "beta")
    L = _lib.lib()
    rows = nb * rows_per_batch
    if out is None:
        out = alloc16((rows, c1 + c2), x1.device)
    ld2 = _ld(x2) if x2 is not None else 0
    keep = (x1, x2, gamma, beta, out)
    if _GN_FUSED and L.avsd_groupnorm_fused_supported(nb, rows_per_batch, groups, c1, c2, int(P.SPLIT)):
        if P.SPLIT:
            a_ = (_p(x1), _ld(x1), c1, _lo(x1), _p(x2), ld2, c2, _lo(x2), nb, rows_per_batch, groups, _p(gamma), _p(beta), float(eps), int(act),
                  _p(out), _ld(out), _lo(out))
            _timed("groupnorm", 0.0, _nbytes(x1, x2) + _nbytes(out), keep, lambda: check(L.avsd_groupnorm_fused_x2(*a_, _stream()), "avsd_groupnorm_fused_x2"))
        else:
            a_ = (_p(x1), _ld(x1), c1, _p(x2), ld2, c2, nb, rows_per_batch, groups, _p(gamma), _p(beta), float(eps), int(act), _p(out), _ld(out))
            _timed("groupnorm", 0.0, _nbytes(x1, x2) + _nbytes(out), keep, lambda: check(L.avsd_groupnorm_fused(*a_, _stream()), "avsd_groupnorm_fused"))
        return out
    nchunks = L.avsd_groupnorm_nchunks(nb, rows_per_batch, c1 + c2)
    partial = torch.empty((L.avsd_groupnorm_scratch_floats(nb, nchunks, groups, c1 + c2),), dtype=F32, device=x1.device)
    keep = keep + (partial,)
    if P.SPLIT:
        s_ = (_p(x1), _ld(x1), c1, _lo(x1), _p(x2), ld2, c2, _lo(x2), nb, rows_per_batch, groups, _p(partial), nchunks)
        a_ = (_p(x1), _ld(x1), c1, _lo(x1), _p(x2), ld2, c2, _lo(x2), nb, rows_per_batch, groups, _p(gamma), _p(beta), float(eps), _p(partial),
              nchunks, int(act), _p(out), _ld(out), _lo(out))
        _timed("groupnorm", 0.0, 2.0 * _nbytes(x1, x2) + _nbytes(out), keep,
               lambda: check(L.avsd_groupnorm_stats_x2(*s_, _stream()), "avsd_groupnorm_stats_x2"),
               lambda: check(L.avsd_groupnorm_apply_x2(*a_, _stream()), "avsd_groupnorm_apply_x2"))
    else:
        s_ = (_p(x1), _ld(x1), c1, _p(x2), ld2, c2, nb, rows_per_batch, groups, _p(partial), nchunks)
        a_ = (_p(x1), _ld(x1), c1, _p(x2), ld2, c2, nb, rows_per_batch, groups, _p(gamma), _p(beta), float(eps), _p(partial), nchunks, int(act),
              _p(out), _ld(out))
        _timed("groupnorm", 0.0, 2.0 * _nbytes(x1, x2) + _nbytes(out), keep,
               lambda: check(L.avsd_groupnorm_stats(*s_, _stream()), "avsd_groupnorm_stats"),
               lambda: check(L.avsd_groupnorm_apply(*a_, _stream()), "avsd_groupnorm_apply"))
    return out


def groupnorm_planes(x1: torch.Tensor, x1_rest: torch.Tensor, nb: int, rows_per_batch: int, groups: int, gamma: torch.Tensor,
                     beta: torch.Tensor, eps: float, act: bool):
    """GroupNorm(+SiLU) of a two-plane tensor into two planes, whatever the process-wide mode: the input of a three-pass product under
    the per-layer precision plan (conv_out reads conv_norm_out's result).  -> (main, rest)"""
    _req(x1, P.ACT, "x1")
    L = _lib.lib()
    c1 = x1.shape[1]
    out, out_rest = alloc_planes((nb * rows_per_batch, c1), x1.device)
    lo_in, lo_out = _rest_off(x1, x1_rest), _rest_off(out, out_rest)
    nchunks = L.avsd_groupnorm_nchunks(nb, rows_per_batch, c1)
    partial = torch.empty((L.avsd_groupnorm_scratch_floats(nb, nchunks, groups, c1),), dtype=F32, device=x1.device)
    s_ = (_p(x1), _ld(x1), c1, lo_in, None, 0, 0, 0, nb, rows_per_batch, groups, _p(partial), nchunks)
    a_ = (_p(x1), _ld(x1), c1, lo_in, None, 0, 0, 0, nb, rows_per_batch, groups, _p(gamma), _p(beta), float(eps), _p(partial), nchunks, int(act),
          _p(out), _ld(out), lo_out)
    _timed("groupnorm", 0.0, 4.0 * _nbytes(x1) + 2.0 * _nbytes(out), (x1, x1_rest, gamma, beta, out, out_rest, partial),
           lambda: check(L.avsd_groupnorm_stats_x2(*s_, _stream()), "avsd_groupnorm_stats_x2"),
           lambda: check(L.avsd_groupnorm_apply_x2(*a_, _stream()), "avsd_groupnorm_apply_x2"))
    return out, out_rest


def ncfhw_to_rows_planes(x: torch.Tensor, cpad: int, rep: int = 1, scale: float = 1.0):
    """ncfhw_to_rows into (main, rest) planes, whatever the process-wide mode (conv_in as a three-pass product)"""
    _req(x, F32, "x")
    if not x.is_contiguous():
        raise ValueError("ncfhw_to_rows: x must be contiguous")
    B, Cc, Fr, H, W = x.shape
    out, rest = alloc_planes((rep * B * Fr * H * W, cpad), x.device)
    check(_lib.lib().avsd_ncfhw_to_rows_x2(_p(x), _p(out), _rest_off(out, rest), B, Cc, Fr, H * W, cpad, rep, float(scale), _stream()),
          "avsd_ncfhw_to_rows_x2")
    return out, rest


def ln_fold(stats: torch.Tensor) -> torch.Tensor:
    """row statistics [M, K/32, 2] of a ROWSTATS producer -> [M, 1, 2]: the pair a LayerNorm-folding consumer (gemm(ln=...)) would fold
    itself in every column tile, folded once (same order, same bits)"""
    _req(stats, F32, "stats")
    M, nblk = stats.shape[0], stats.shape[1]
    out = torch.empty((M, 1, 2), dtype=F32, device=stats.device)
    _timed("ln_fold", 0.0, _nbytes(stats, out), (stats, out), lambda: check(_lib.lib().avsd_ln_fold(_p(stats), M, nblk, _p(out), _stream()), "avsd_ln_fold"))
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              pos: Optional[torch.Tensor] = None, hw: int = 1, frames: int = 1,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x, P.ACT, "x")
    _req(gamma, F32, "gamma")
    _req(beta, F32, "beta")
    M, Cc = x.shape
    if pos is not None:
        _req(pos, F32, "pos")
        if not pos.is_contiguous() or pos.shape != (frames, Cc):
            raise ValueError("layernorm: pos must be contiguous [frames, C]")
    if out is None:
        out = alloc16((M, Cc), x.device)
    if P.SPLIT:
        a_ = (_p(x), _ld(x), _lo(x), _p(out), _ld(out), _lo(out), M, Cc, _p(gamma), _p(beta), float(eps), _p(pos), hw, frames)
        call = lambda: check(_lib.lib().avsd_layernorm_x2(*a_, _stream()), "avsd_layernorm_x2")   # noqa: E731
    else:
        a_ = (_p(x), _ld(x), _p(out), _ld(out), M, Cc, _p(gamma), _p(beta), float(eps), _p(pos), hw, frames)
        call = lambda: check(_lib.lib().avsd_layernorm(*a_, _stream()), "avsd_layernorm")   # noqa: E731
    _timed("layernorm", 0.0, _nbytes(x, out), (x, out, gamma, beta, pos), call)
    return out


def softmax_rows(s: torch.Tensor) -> torch.Tensor:
    _req(s, F32, "s")
    rows, L = s.shape
    out = alloc16((rows, L), s.device)
    check(_lib.lib().avsd_softmax_rows_x2(_p(s), _ld(s), _p(out), _ld(out), _lo(out), rows, L, _stream()), "avsd_softmax_rows_x2")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, bq: int, lq: int, lk: int, kv_rows: int,
              heads: int, q_per_kv: int, frames: int, key_index: Optional[torch.Tensor] = None,
              scale: Optional[float] = None, out: Optional[torch.Tensor] = None, fp8: Optional[tuple] = None) -> torch.Tensor:
    """q [bq*lq, heads*d]; k, v [(bq/q_per_kv)*kv_rows, >= heads*d] (views into a fused k|v buffer are fine).
    fp8 = (q_scale, k_scale, v_scale): the e4m3 Q/K/V kernel (avsd_attention_fp8, BASELINE cfg 5) instead of the 16-bit one."""
    _req(q, P.ACT, "q")
    _req(k, P.ACT, "k")
    _req(v, P.ACT, "v")
    Cc = q.shape[1]
    d = Cc // heads
    if key_index is not None:
        if key_index.dtype != torch.int32 or not key_index.is_contiguous() or key_index.shape != (frames, lk):
            raise ValueError("attention: key_index must be contiguous int32 [frames, lk]")
    if out is None:
        out = alloc16((bq * lq, Cc), q.device)
    if scale is None:
        scale = float(d) ** -0.5
    if P.SPLIT:
        if fp8 is not None:
            raise ValueError("attention: fp8 Q/K/V and split precision are exclusive")
        a_ = (_p(q), _ld(q), _lo(q), _p(k), _ld(k), _lo(k), _p(v), _ld(v), _lo(v), _p(out), _ld(out), _lo(out), bq, lq, lk, kv_rows, heads, d,
              q_per_kv, _p(key_index), frames, float(scale))
        call = lambda: check(_lib.lib().avsd_attention_x2(*a_, _stream()), "avsd_attention_x2")   # noqa: E731
    elif fp8 is not None:
        qs, ks, vs = (float(x) for x in fp8)
        a_ = (_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(out), _ld(out), bq, lq, lk, kv_rows, heads, d, q_per_kv, _p(key_index), frames,
              float(scale), qs, ks, vs)
        call = lambda: check(_lib.lib().avsd_attention_fp8(*a_, _stream()), "avsd_attention_fp8")   # noqa: E731
    else:
        a_ = (_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(out), _ld(out), bq, lq, lk, kv_rows, heads, d, q_per_kv, _p(key_index), frames,
              float(scale))
        call = lambda: check(_lib.lib().avsd_attention(*a_, _stream()), "avsd_attention")   # noqa: E731
    _timed("attention", 4.0 * bq * heads * lq * lk * d * (3 if P.SPLIT else 1), _nbytes(q, out) + 4.0 * (bq // q_per_kv) * lk * Cc,
           (q, k, v, out, key_index), call)
    return out


def cross_attention_block_supported(C: int, heads: int, lk_pad: int, M: int, L: int) -> bool:
    return not P.SPLIT and bool(_lib.lib().avsd_cross_attention_block_supported(C, heads, lk_pad)) and M % 128 == 0 and L % 128 == 0


def cross_attention_block(h: torch.Tensor, stats: torch.Tensor, wq: torch.Tensor, q_colsum: torch.Tensor, q_bias: torch.Tensor,
                          k: torch.Tensor, vt: torch.Tensor, lk: int, wo: torch.Tensor, o_bias: torch.Tensor, *, res: torch.Tensor,
                          heads: int, L: int, q_per_kv: int, eps: float = 1e-5, scale: Optional[float] = None,
                          rowstats: Optional[torch.Tensor] = None, master: Optional[torch.Tensor] = None,
                          out: Optional[torch.Tensor] = None, stats_pos: Optional[tuple] = None) -> torch.Tensor:
    """out = res + to_out(softmax((LN(h) Wq^T) K^T scale) V) in one launch; see avsd_cross_attention_block (include/avsd.h).
    k [nkv, lk_pad, C], vt [nkv, C, lk_pad] are the cached, padded (and for audio mask-gathered) K / V^T."""
    if P.SPLIT:
        raise ValueError("cross_attention_block: not built for split precision (use the separate kernels)")
    _req(h, P.ACT, "h")
    _req(wq, P.ACT, "wq")
    _req(wo, P.ACT, "wo")
    _req(k, P.ACT, "k")
    _req(vt, P.ACT, "vt")
    _req(stats, F32, "stats")
    M, Cc = h.shape
    lk_pad = k.shape[1]
    if not k.is_contiguous() or not vt.is_contiguous() or vt.shape != (k.shape[0], Cc, lk_pad) or k.shape[2] != Cc:
        raise ValueError("cross_attention_block: k must be contiguous [nkv, lk_pad, C] and vt [nkv, C, lk_pad]")
    if not stats.is_contiguous() or stats.shape != (M, Cc // 32, 2):
        raise ValueError("cross_attention_block: stats must be contiguous f32 [M, C/32, 2]")
    if (M // L) // q_per_kv > k.shape[0] or (M // L) % q_per_kv:
        raise ValueError("cross_attention_block: K/V blocks do not cover the query batches")
    if out is None:
        out = torch.empty((M, Cc), dtype=P.ACT, device=h.device)
    d = XAttnDesc()
    d.h, d.ldh = _p(h), _ld(h)
    _req(res, F32 if res.dtype == F32 else P.ACT, "res")
    d.res, d.ldres, d.res_f32 = _p(res), _ld(res), int(res.dtype == F32)
    d.M, d.C, d.heads, d.L = M, Cc, heads, L
    d.ln_stats, d.ln_eps = _p(stats), float(eps)
    d.scale = float(scale) if scale is not None else float(Cc // heads) ** -0.5
    d.wq, d.ldwq, d.q_colsum, d.q_bias = _p(wq), _ld(wq), _p(q_colsum), _p(q_bias)
    d.k, d.vt, d.lk, d.lk_pad, d.q_per_kv = _p(k), _p(vt), lk, lk_pad, q_per_kv
    d.wo, d.ldwo, d.o_bias = _p(wo), _ld(wo), _p(o_bias)
    d.out, d.ldo = _p(out), _ld(out)
    if master is not None:
        _req(master, F32, "master")
        d.out_master, d.ldm = _p(master), _ld(master)
    if rowstats is not None:
        _req(rowstats, F32, "rowstats")
        d.rowstats = _p(rowstats)
    if stats_pos is not None:
        tbl, hw_, fr_ = stats_pos
        _req(tbl, F32, "stats_pos")
        if rowstats is None or not tbl.is_contiguous() or tbl.shape != (fr_, Cc) or hw_ <= 0:
            raise ValueError("cross_attention_block: stats_pos = (f32 [frames, C], hw, frames) goes with rowstats")
        d.stats_pos, d.pos_hw, d.pos_frames = _p(tbl), int(hw_), int(fr_)
    ev = _TIMER.start() if _TIMER is not None else None
    check(_lib.lib().avsd_cross_attention_block(C.byref(d), _stream()), "avsd_cross_attention_block")
    if ev is not None:
        dc = XAttnDesc.from_buffer_copy(d)
        _TIMER.add_replay("cross_attention_block", lambda dc=dc: check(_lib.lib().avsd_cross_attention_block(C.byref(dc), _stream()),
                                                                         "avsd_cross_attention_block"),
                          (h, stats, wq, q_colsum, q_bias, k, vt, wo, o_bias, res, out, master, rowstats, stats_pos))
        _TIMER.stop(ev, "cross_attention_block", 4.0 * M * Cc * Cc + 4.0 * M * lk * Cc, _nbytes(h, out, res, master) + 4.0 * Cc * Cc)
    return out


def temporal_attention(qkv: torch.Tensor, *, b: int, frames: int, hw: int, heads: int,
                       scale: Optional[float] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(qkv, P.ACT, "qkv")
    Cc = qkv.shape[1] // 3
    d = Cc // heads
    if out is None:
        out = alloc16((qkv.shape[0], Cc), qkv.device)
    if scale is None:
        scale = float(d) ** -0.5
    if P.SPLIT:
        a_ = (_p(qkv), _ld(qkv), _lo(qkv), _p(out), _ld(out), _lo(out), b, frames, hw, heads, d, float(scale))
        call = lambda: check(_lib.lib().avsd_temporal_attention_x2(*a_, _stream()), "avsd_temporal_attention_x2")   # noqa: E731
    else:
        a_ = (_p(qkv), _ld(qkv), _p(out), _ld(out), b, frames, hw, heads, d, float(scale))
        call = lambda: check(_lib.lib().avsd_temporal_attention(*a_, _stream()), "avsd_temporal_attention")   # noqa: E731
    _timed("temporal_attention", 4.0 * b * hw * heads * frames * frames * d, _nbytes(qkv, out), (qkv, out), call)
    return out


def ncfhw_to_rows(x: torch.Tensor, cpad: int, rep: int = 1, scale: float = 1.0) -> torch.Tensor:
    _req(x, F32, "x")
    if not x.is_contiguous():
        raise ValueError("ncfhw_to_rows: x must be contiguous")
    B, Cc, Fr, H, W = x.shape
    out = alloc16((rep * B * Fr * H * W, cpad), x.device)
    if P.SPLIT:
        check(_lib.lib().avsd_ncfhw_to_rows_x2(_p(x), _p(out), _lo(out), B, Cc, Fr, H * W, cpad, rep, float(scale), _stream()),
              "avsd_ncfhw_to_rows_x2")
        return out
    check(_lib.lib().avsd_ncfhw_to_rows(_p(x), _p(out), B, Cc, Fr, H * W, cpad, rep, float(scale), _stream()),
          "avsd_ncfhw_to_rows")
    return out


def copy(src: torch.Tensor, dst: Optional[torch.Tensor] = None, rep: int = 1) -> torch.Tensor:
    """dst = `rep` copies of the contiguous tensor `src` along dim 0 (torch.cat([src] * rep)); with dst given, a device copy into
    it.  A library launch instead of a torch op, so that it is part of a recorded launch plan (asva_amd/plan.py)."""
    if not (src.is_cuda and src.is_contiguous()):
        raise ValueError("copy: src must be a contiguous device tensor")
    split = P.SPLIT and src.dtype == P.ACT          # a split tensor: both planes move
    if dst is None:
        shape = (rep * src.shape[0],) + tuple(src.shape[1:])
        dst = alloc16(shape, src.device) if split else torch.empty(shape, dtype=src.dtype, device=src.device)
    nbytes = src.numel() * src.element_size()
    if not dst.is_contiguous() or dst.numel() * dst.element_size() != rep * nbytes or nbytes % 16:
        raise ValueError("copy: dst must be contiguous and hold rep x src (a multiple of 16 bytes)")
    check(_lib.lib().avsd_copy(_p(src), _p(dst), nbytes, rep, _stream()), "avsd_copy")
    if split:
        check(_lib.lib().avsd_copy(_p(src) + 2 * _lo(src), _p(dst) + 2 * _lo(dst), nbytes, rep, _stream()), "avsd_copy")
    return dst


def xattn_pack_kv(kv: torch.Tensor, n_kv: int, rows: int, Cc: int, idx: Optional[torch.Tensor], k_out: torch.Tensor,
                  vt_out: torch.Tensor) -> None:
    """kv [n_kv*rows, 2C] -> k_out [nb, lk_pad, C], vt_out [nb, C, lk_pad] (operands of cross_attention_block); idx [F, nk]
    int32 gathers the visible keys of each frame (nb = n_kv * F).  The padding slots lk..lk_pad are zero-filled by the same launch."""
    _req(kv, P.ACT, "kv")
    _req(k_out, P.ACT, "k_out")
    _req(vt_out, P.ACT, "vt_out")
    if idx is not None:
        _req(idx, torch.int32, "idx")
    nfr, nk = (idx.shape if idx is not None else (0, 0))
    nb = n_kv * nfr if idx is not None else n_kv
    lkp = k_out.shape[1]
    if tuple(k_out.shape) != (nb, lkp, Cc) or tuple(vt_out.shape) != (nb, Cc, lkp) or not (kv.is_contiguous() and k_out.is_contiguous() and vt_out.is_contiguous()):
        raise ValueError("xattn_pack_kv: k_out [nb, lk_pad, C] / vt_out [nb, C, lk_pad] expected, all contiguous")
    check(_lib.lib().avsd_xattn_pack_kv(_p(kv), n_kv, rows, Cc, _p(idx), nfr, nk, _p(k_out), _p(vt_out), lkp, _stream()),
          "avsd_xattn_pack_kv")


def rows_to_ncfhw(rows: torch.Tensor, B: int, Cc: int, Fr: int, H: int, W: int) -> torch.Tensor:
    _req(rows, F32, "rows")
    out = torch.empty((B, Cc, Fr, H, W), dtype=F32, device=rows.device)
    check(_lib.lib().avsd_rows_to_ncfhw(_p(rows), _ld(rows), _p(out), B, Cc, Fr, H * W, _stream()), "avsd_rows_to_ncfhw")
    return out


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    _req(t, F32, "t")
    n = t.numel()
    out = torch.empty((n, dim), dtype=F32, device=t.device)
    check(_lib.lib().avsd_timestep_embedding(_p(t), _p(out), n, dim, _stream()), "avsd_timestep_embedding")
    return out


def guided_step(noise_pred: torch.Tensor, n_branch: int, g: float, x_in: torch.Tensor, x_out: torch.Tensor,
                ca: float, cb: float, *, eps_hist: Optional[torch.Tensor] = None, store_slot: int = -1,
                w_cur: float = 1.0, hist_idx=(), w=(), g2: float = 0.0) -> None:
    _req(noise_pred, F32, "noise_pred")
    _req(x_in, F32, "x_in")
    _req(x_out, F32, "x_out")
    B, Cc, Fr, H, W = x_in.shape
    nh = len(hist_idx)
    idx = (C.c_int32 * 4)(*(list(hist_idx) + [0] * (4 - nh)))
    ws = (C.c_float * 4)(*(list(w) + [0.0] * (4 - nh)))
    check(_lib.lib().avsd_guided_step(_p(noise_pred), n_branch, float(g), float(g2), _p(eps_hist), store_slot, float(w_cur), idx, ws,
                                      nh, _p(x_in), _p(x_out), float(ca), float(cb), B, Cc, Fr, H * W, _stream()),
          "avsd_guided_step")


def vae_postprocess(rows: torch.Tensor, n_img: int, H: int, W: int) -> torch.Tensor:
    _req(rows, P.ACT, "rows")
    out = torch.empty((n_img, 3, H, W), dtype=F32, device=rows.device)
    if P.SPLIT:
        check(_lib.lib().avsd_vae_postprocess_x2(_p(rows), _ld(rows), _lo(rows), _p(out), n_img, H * W, _stream()), "avsd_vae_postprocess_x2")
        return out
    check(_lib.lib().avsd_vae_postprocess(_p(rows), _ld(rows), _p(out), n_img, H * W, _stream()), "avsd_vae_postprocess")
    return out


def vae_postprocess_u8(rows: torch.Tensor, n_img: int, H: int, W: int) -> torch.Tensor:
    _req(rows, P.ACT, "rows")
    out = torch.empty((n_img, H, W, 3), dtype=torch.uint8, device=rows.device)
    if P.SPLIT:
        check(_lib.lib().avsd_vae_postprocess_u8_x2(_p(rows), _ld(rows), _lo(rows), _p(out), n_img, H * W, _stream()),
              "avsd_vae_postprocess_u8_x2")
        return out
    check(_lib.lib().avsd_vae_postprocess_u8(_p(rows), _ld(rows), _p(out), n_img, H * W, _stream()), "avsd_vae_postprocess_u8")
    return out


# ---- audio conditioning front-end ------------------------------------------------------------------
def kaldi_fbank(wave: torch.Tensor, window: torch.Tensor, mel_fb: torch.Tensor, *, shift: int, nfft: int, t_out: int,
                preemph: float = 0.97, remove_dc: bool = True, mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    """wave (B, n_samples) f32 -> normalised log-mel (B, n_mel, t_out) f32; see avsd_kaldi_fbank."""
    _req(wave, F32, "wave")
    _req(window, F32, "window")
    _req(mel_fb, F32, "mel_fb")
    if wave.dim() != 2 or not mel_fb.is_contiguous() or mel_fb.shape[1] != nfft // 2 + 1:
        raise ValueError("kaldi_fbank: wave must be (B, n), mel_fb contiguous [n_mel][nfft/2+1]")
    B, n = wave.shape
    out = torch.empty((B, mel_fb.shape[0], t_out), dtype=F32, device=wave.device)
    check(_lib.lib().avsd_kaldi_fbank(_p(wave), B, n, wave.stride(0), _p(window), _p(mel_fb), window.numel(), shift, nfft,
                                      mel_fb.shape[0], float(preemph), int(remove_dc), _p(out), t_out, float(mean), float(std),
                                      _stream()), "avsd_kaldi_fbank")
    return out


def patchify(x: torch.Tensor, kh: int, kw: int, stride: int) -> torch.Tensor:
    """(B, C, H, W) f32 -> bf16 rows [B*ph*pw, C*kh*kw]."""
    _req(x, F32, "x")
    if not x.is_contiguous():
        raise ValueError("patchify: x must be contiguous")
    B, Cc, H, W = x.shape
    ph, pw = (H - kh) // stride + 1, (W - kw) // stride + 1
    if P.SPLIT:
        # split precision (once per clip, conditioning side): the gather is a strided device copy, the two planes come from
        # avsd_split_f32 — same values as the kernel's, with the rest plane kept
        cols = torch.nn.functional.unfold(x, (kh, kw), stride=stride)                 # (B, C*kh*kw, ph*pw), c-major then kh, kw
        return to_act(cols.transpose(1, 2).reshape(B * ph * pw, Cc * kh * kw))
    out = torch.empty((B * ph * pw, Cc * kh * kw), dtype=P.ACT, device=x.device)
    check(_lib.lib().avsd_patchify(_p(x), _p(out), B, Cc, H, W, kh, kw, stride, _stream()), "avsd_patchify")
    return out


def vit_tokens(patches: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, b: int, tail_rows: int = 0) -> torch.Tensor:
    """bf16 patch embeddings [b*np, C] + cls [C] + pos [1+np, C] (f32) -> bf16 [b*(1+np+tail_rows), C]."""
    _req(patches, P.ACT, "patches")
    _req(cls, F32, "cls")
    _req(pos, F32, "pos")
    n_p, Cc = patches.shape[0] // b, patches.shape[1]
    if not patches.is_contiguous() or pos.shape != (1 + n_p, Cc) or cls.numel() != Cc:
        raise ValueError("vit_tokens: shape mismatch")
    if P.SPLIT:          # split precision: the same f32 sums, both planes (device ops + avsd_split_f32; once per clip)
        tok = torch.zeros((b, 1 + n_p + tail_rows, Cc), dtype=F32, device=patches.device)
        tok[:, 0] = cls.reshape(1, Cc) + pos[0]
        tok[:, 1:1 + n_p] = from_act(patches).view(b, n_p, Cc) + pos[1:]
        return to_act(tok.view(-1, Cc))
    out = torch.empty((b * (1 + n_p + tail_rows), Cc), dtype=P.ACT, device=patches.device)
    check(_lib.lib().avsd_vit_tokens(_p(patches), _p(cls), _p(pos), _p(out), b, n_p, Cc, tail_rows, _stream()), "avsd_vit_tokens")
    return out
