"""Launch plans: record what the Python host asks of libavsd_hip.so, write it as a bundle any host can replay.

SURVEY.md 8(b)-3 proposed plan-level entry points (`avsd_unet_plan_create / _set_conditioning / _forward`, `avsd_vae_decode`)
so that a host without Python can run the path.  The network description (config, weight packing, tile table) lives in this
package, so the plan is produced here: a `Recorder` intercepts every call the ctypes binding makes while the model runs
once — `set_conditioning`, one `denoise_forward`, a VAE `decode` — and stores (entry point, arguments) with each device
pointer rewritten as (buffer, byte offset).  `include/avsd.h` ("launch plans") documents the bundle and the C API that loads,
binds and runs it; `tools/plan_host.cpp` is a host written against that API alone.

    rec = Recorder()
    rec.region("weights", unet.pack().blob, CONST)            # named regions: what the replaying host uploads / reads
    rec.region("x", x, INPUT); ...
    with rec.record("forward"):
        noise = unet.denoise_forward(x, t, rep=2)
    rec.region("noise_pred", noise, OUTPUT)
    rec.save("step.plan")                                     # + step.plan.d/<name>.bin for CONST regions

Memory model: the bundle's buffers are the caching allocator's SEGMENTS that the recorded calls touched, with their sizes;
every device pointer becomes (segment, offset).  The replaying host allocates each segment zero-filled, so its memory
layout equals the recording run's — temporaries that reused one another's addresses do so again, and what
"set_conditioning" leaves in a segment is what "forward" reads (the plans of a bundle share the buffer table).  Regions are
labels on that memory.  Only launches that go through the library are recorded — a torch op inside a recorded stretch would
be missing from the plan, which is why the product path uses `ops.copy` / `ops.xattn_pack_kv` there instead of `torch.cat` /
indexing; tests/test_plan_gpu.py replays every plan against fresh buffers to prove it complete.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import struct
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from . import precision as P

CONST, INPUT, OUTPUT = 1, 2, 3
MAGIC = b"AVSDPLN1"
_NOT_RECORDED = {"avsd_abi_version", "avsd_precision", "avsd_last_error", "avsd_device_info", "avsd_sizeof_gemm_desc",
                 "avsd_sizeof_xattn_desc", "avsd_cross_attention_block_supported", "avsd_groupnorm_nchunks",
                 "avsd_groupnorm_scratch_floats", "avsd_groupnorm_fused_supported", "avsd_gemm_conv3r_supported", "avsd_gemm_conv3r2d_supported"}


def _ptr_fields(struct_type) -> List[Tuple[str, int]]:
    return [(name, getattr(struct_type, name).offset) for name, tp in struct_type._fields_ if tp is C.c_void_p]


class _Proxy:
    """Stands in for the CDLL handle while recording: launching entry points are executed AND logged."""

    def __init__(self, real, sink):
        self._real, self._sink = real, sink

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if (name in _NOT_RECORDED or name.startswith("avsd_plan_") or name.startswith("avsd_unet_") or name == "avsd_vae_decode"
                or name not in _lib.SIGNATURES):
            return fn
        argtypes = _lib.SIGNATURES[name][1]

        def call(*args):
            rc = fn(*args)
            if rc == 0:
                self._sink.append((name, self._convert(name, argtypes, args)))
            return rc

        return call

    @staticmethod
    def _convert(name, argtypes, args):
        out = []
        last = len(argtypes) - 1
        for k, (tp, a) in enumerate(zip(argtypes, args)):
            if tp is C.c_void_p:
                out.append(("S",) if k == last else ("p", 0 if a is None else int(a)))
            elif tp in (C.c_int, C.c_int64, C.c_int32):
                out.append(("i", int(a)))
            elif tp is C.c_float:
                out.append(("f", float(a)))
            elif isinstance(tp, type) and issubclass(tp, C._Pointer):
                target = tp._type_
                if issubclass(target, C.Structure):           # descriptor by reference: C.byref(desc)
                    desc = a._obj
                    relocs = [(off, int(getattr(desc, f) or 0)) for f, off in _ptr_fields(target)]
                    out.append(("s", bytes(desc), relocs))
                else:                                         # small host array (scheduler history indices / weights)
                    out.append(("h", b"" if a is None else bytes(a)))
            else:
                raise TypeError(f"{name}: cannot record an argument of type {tp}")
        return out


class Recorder:
    def __init__(self):
        self.named: List[dict] = []
        self.plans: Dict[str, list] = {}
        self._live: Optional[list] = None      # allocations alive when the first recording started: (address, size)

    def region(self, name: str, tensor: torch.Tensor, kind: int, data: Optional[bool] = None) -> None:
        """Names the memory of `tensor` (contiguous, device) so that a host can find it: CONST (contents exported with the
        bundle), INPUT, OUTPUT — a region is a label on the allocator segment that holds it.  CONST and INPUT tensors must
        have been allocated BEFORE the first `record()` and stay alive until `save()`: only then is their memory never
        lent to a temporary of a recorded plan, i.e. what the host writes there survives running the plans (`save` checks).
        An OUTPUT may be produced inside a plan; it is valid from the end of that plan's run until the next run."""
        if not tensor.is_cuda or not tensor.is_contiguous():
            raise ValueError(f"{name}: plan regions are contiguous device tensors")
        if any(b["name"] == name for b in self.named):
            raise ValueError(f"region {name} declared twice")
        self.named.append(dict(name=name, ptr=tensor.data_ptr(), bytes=tensor.numel() * tensor.element_size(), kind=kind,
                               tensor=tensor, data=(kind == CONST) if data is None else data))

    @contextlib.contextmanager
    def record(self, plan: str):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("record a plan eagerly, not under graph capture")
        if self._live is None:
            self._live = []
            for seg in torch.cuda.memory_snapshot():
                addr = seg["address"]
                for blk in seg["blocks"]:
                    if blk["state"] == "active_allocated":
                        self._live.append((addr, blk["size"]))
                    addr += blk["size"]
        real = _lib.lib()
        sink = self.plans.setdefault(plan, [])
        _lib._libs[P.NAME] = _Proxy(real, sink)
        try:
            yield self
        finally:
            _lib._libs[P.NAME] = real
        torch.cuda.synchronize()

    # ---- pointer -> (buffer, offset): buffers are the caching allocator's segments -------------------------------
    def _resolve(self):
        segs = sorted((s["address"], s["total_size"]) for s in torch.cuda.memory_snapshot())
        starts = [a for a, _ in segs]
        used: Dict[int, int] = {}            # segment index -> buffer index, in order of first use
        import bisect

        def find(ptr: int) -> Tuple[int, int]:
            if ptr == 0:
                return -1, 0
            k = bisect.bisect_right(starts, ptr) - 1
            if k < 0 or ptr >= segs[k][0] + segs[k][1]:
                raise RuntimeError(f"device pointer {ptr:#x} is in no segment of the caching allocator (was it freed to the driver?)")
            if k not in used:
                used[k] = len(used)
            return used[k], ptr - segs[k][0]

        plans = {}
        for pname, calls in self.plans.items():
            out = []
            for fn, args in calls:
                conv = []
                for a in args:
                    if a[0] == "p":
                        conv.append(("p",) + find(a[1]))
                    elif a[0] == "s":
                        conv.append(("s", a[1], [(off,) + find(ptr) for off, ptr in a[2]]))
                    else:
                        conv.append(a)
                out.append((fn, conv))
            plans[pname] = out
        regions = []
        for b in self.named:
            if b["kind"] in (CONST, INPUT) and not any(a <= b["ptr"] and b["ptr"] + b["bytes"] <= a + n for a, n in (self._live or [])):
                raise RuntimeError(f"region {b['name']}: CONST / INPUT tensors must be allocated before the first record() — "
                                   "memory allocated later may have served a temporary of a recorded plan")
            buf, off = find(b["ptr"])
            regions.append(dict(name=b["name"], buf=buf, off=off, bytes=b["bytes"], kind=b["kind"], tensor=b["tensor"], data=b["data"]))
        buffers = [segs[k][1] for k, _ in sorted(used.items(), key=lambda kv: kv[1])]
        return buffers, regions, plans

    def save(self, path: str, export_inputs: bool = False) -> "Bundle":
        """Writes `path` (+ `path`.d/<region>.bin for regions whose contents travel).  Returns the bundle, loaded."""
        torch.cuda.synchronize()
        buffers, regions, plans = self._resolve()
        with open(path, "wb") as f:
            f.write(MAGIC)
            f.write(struct.pack("<I", _lib.lib().avsd_abi_version()))
            f.write(P.NAME.encode().ljust(8, b"\0"))
            f.write(struct.pack("<I", len(buffers)))
            for nbytes in buffers:
                f.write(struct.pack("<q", nbytes))
            f.write(struct.pack("<I", len(regions)))
            for r in regions:
                nm = r["name"].encode()
                f.write(struct.pack("<I", len(nm)) + nm + struct.pack("<iqqI", r["buf"], r["off"], r["bytes"], r["kind"]))
            f.write(struct.pack("<I", len(plans)))
            for pname, calls in plans.items():
                nm = pname.encode()
                f.write(struct.pack("<I", len(nm)) + nm + struct.pack("<I", len(calls)))
                for fn, args in calls:
                    nm = fn.encode()
                    f.write(struct.pack("<I", len(nm)) + nm + struct.pack("<I", len(args)))
                    for a in args:
                        f.write(a[0].encode())
                        if a[0] == "i":
                            f.write(struct.pack("<q", a[1]))
                        elif a[0] == "f":
                            f.write(struct.pack("<d", a[1]))
                        elif a[0] == "p":
                            f.write(struct.pack("<iq", a[1], a[2]))
                        elif a[0] == "h":
                            f.write(struct.pack("<I", len(a[1])) + a[1])
                        elif a[0] == "s":
                            f.write(struct.pack("<I", len(a[1])) + a[1] + struct.pack("<I", len(a[2])))
                            for off, buf, boff in a[2]:
                                f.write(struct.pack("<Iiq", off, buf, boff))
        ddir = path + ".d"
        os.makedirs(ddir, exist_ok=True)
        for r in regions:
            if r["data"] or (export_inputs and r["kind"] == INPUT):
                r["tensor"].detach().reshape(-1).view(torch.uint8).cpu().numpy().tofile(os.path.join(ddir, r["name"] + ".bin"))
        return Bundle(path, {k: len(v) for k, v in plans.items()})


def export_steps(path: str, timesteps, plans) -> None:
    """The scalar schedule of a denoising run (schedulers.plan_step: DDIM / PLMS coefficients of avsd_guided_step) as the
    binary table tools/plan_host.cpp reads: per step  f32 t, ca, cb, w_cur; i32 store_slot, n_hist, save_sample, use_saved;
    i32 hist_idx[4]; f32 hist_w[4]."""
    with open(path, "wb") as f:
        for t, p in zip(timesteps, plans):
            idx = list(p.hist_idx) + [0] * (4 - len(p.hist_idx))
            w = list(p.hist_w) + [0.0] * (4 - len(p.hist_w))
            f.write(struct.pack("<4f4i4i4f", float(t), float(p.ca), float(p.cb), float(p.w_cur), int(p.store_slot), len(p.hist_idx),
                                int(p.save_sample), int(p.use_saved_sample), *[int(i) for i in idx], *[float(x) for x in w]))


class Bundle:
    """A saved bundle, replayed in-process through the C API (what tools/plan_host.cpp does from C++)."""

    def __init__(self, path: str, n_calls=None):
        self.path, self.n_calls = path, n_calls
        self._h = C.c_void_p()
        _lib.check(_lib.lib().avsd_plan_bundle_load(path.encode(), C.byref(self._h)), "avsd_plan_bundle_load")
        self._bufs: List[torch.Tensor] = []

    def close(self):
        if self._h:
            _lib.lib().avsd_plan_bundle_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass

    def buffer_sizes(self) -> List[int]:
        L = _lib.lib()
        return [L.avsd_plan_bundle_buffer_bytes(self._h, i) for i in range(L.avsd_plan_bundle_num_buffers(self._h))]

    def regions(self) -> Dict[str, Tuple[int, int, int, int]]:
        """name -> (buffer, offset, bytes, kind)"""
        L = _lib.lib()
        out = {}
        for j in range(L.avsd_plan_bundle_num_regions(self._h)):
            nm, bf, off, nb, kd = C.c_char_p(), C.c_int(), C.c_int64(), C.c_int64(), C.c_int()
            _lib.check(L.avsd_plan_bundle_region(self._h, j, C.byref(nm), C.byref(bf), C.byref(off), C.byref(nb), C.byref(kd)),
                       "avsd_plan_bundle_region")
            out[nm.value.decode()] = (bf.value, off.value, nb.value, kd.value)
        return out

    def bind_fresh(self, device) -> None:
        """Every buffer gets a new zero-filled allocation; CONST regions are loaded from the bundle's data directory —
        exactly what a replaying host starts from."""
        import numpy as np

        self._bufs = [torch.zeros(n, dtype=torch.uint8, device=device) for n in self.buffer_sizes()]
        for i, t in enumerate(self._bufs):
            _lib.check(_lib.lib().avsd_plan_bundle_bind(self._h, i, t.data_ptr()), "avsd_plan_bundle_bind")
        for name, (_, _, _, kind) in self.regions().items():
            if kind == CONST:
                self.view(name).copy_(torch.from_numpy(np.fromfile(os.path.join(self.path + ".d", name + ".bin"), dtype=np.uint8)))

    def view(self, name: str) -> torch.Tensor:
        """the bytes of region `name` inside the bound buffers"""
        buf, off, nb, _ = self.regions()[name]
        return self._bufs[buf][off:off + nb]

    def run(self, plan: str) -> None:
        k = _lib.lib().avsd_plan_bundle_find_plan(self._h, plan.encode())
        if k < 0:
            raise KeyError(plan)
        _lib.check(_lib.lib().avsd_plan_run(self._h, k, torch.cuda.current_stream().cuda_stream), "avsd_plan_run")
