"""A/B of the graph-replayed cfg-2 step with some table keys swapped to another tile: alternating measurements on one box.
    python tools/tile_swap_ab.py --tile 67 --where "k[2] == 320 and k[1] == 24576" [--rounds 4] [--reps 40]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from asva_amd import ops
from asva_amd.conditioning import audio_segment_mask
from asva_amd.engine import DenoiseEngine
from asva_amd.schedulers import DDIMScheduler

ap = argparse.ArgumentParser()
ap.add_argument("--tile", type=int, default=67)
ap.add_argument("--split-k", type=int, default=1)
ap.add_argument("--where", default="k[2] == 320 and k[1] == 24576", help="python expression over the table key k = (mode, M, N, K, flags, ...)")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda", 0)
unet = bench.build_unet(dev, 0, 1)
lat, text, audio, null_audio = bench.synthetic_clip(dev, 1000, n=1)
ops.RECORD_KEYS = {}
eng = DenoiseEngine(unet, DDIMScheduler(), 4.0, use_graph=True)
eng.set_conditioning(text, audio, null_audio, audio_segment_mask(12), 12)
x = lat.clone()
eng.prepare(x, 50)
unet.denoise_forward(x, torch.full((1,), 501.0, device=dev), rep=2)
keys, ops.RECORD_KEYS = ops.RECORD_KEYS, None
table = ops.tile_cache()
sel = [k for k in keys if k in table and k[0] in (0, 1) and (a.tile, a.split_k) in keys[k]["cands"] and eval(a.where, {"k": k})]
for k in sel:
    print(k, table[k], "n =", keys[k]["n"])
orig = {k: table[k] for k in sel}


def step_ms():
    eng._graph = None
    for i in range(3):
        eng.step(x, i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.reps):
            eng.step(x, i % 50)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / a.reps)
    x.copy_(lat)
    return best


for r in range(a.rounds):
    for k in sel:
        table[k] = orig[k]
    t0 = step_ms()
    for k in sel:
        table[k] = (a.tile, a.split_k)
    t1 = step_ms()
    print(f"round {r}: table {t0:.4f} ms   swapped {t1:.4f} ms   ({(t0 - t1) * 1e3:+.1f} us)", flush=True)
