"""In-step tile tuner: coordinate descent on the committed table, scored by the graph-replayed denoising STEP itself.

Isolated timing (tools/tune_tiles.py: hot replay + one cold launch) ranks tiles by how fast they run alone; inside a step the weights
stream from HBM, the previous kernel's output sits in another XCD's L2 and ~600 kernels share the clocks — several tiles that win
alone lose in the step (round 4: the asm tiles of the low-resolution layers were 20 % faster alone and 0 % in the step until their K
walk was rotated).  This tool takes every table key the BASELINE cfg-2 forward uses (most launches x FLOPs first), tries each
candidate tile for that key with everything else fixed — re-captures the step graph, times REPS replays — and keeps a candidate only
if the whole step gets faster by more than the noise floor.

    python tools/tune_in_step.py [--out gpurun_out/tiles_instep.json] [--min-launches 2] [--reps 30] [--max-keys 60]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from asva_amd import ops  # noqa: E402
from asva_amd.conditioning import audio_segment_mask  # noqa: E402
from asva_amd.engine import DenoiseEngine  # noqa: E402
from asva_amd.schedulers import DDIMScheduler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/tiles_instep.json")
    ap.add_argument("--min-launches", type=int, default=4)
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--max-keys", type=int, default=80)
    ap.add_argument("--clips", type=int, default=1)
    ap.add_argument("--modes", default="0,1,2", help="A-loader modes of the keys to visit: 0 plain, 1 temporal mix, 2 conv3x3")
    ap.add_argument("--only-tiles", default="", help="try only these tile ids as candidates (e.g. 67: a newly added tile against the table)")
    ap.add_argument("--split", action="store_true", help="tune the split-precision step (AVSD_GEMM_X2 keys)")
    a = ap.parse_args()
    if a.split:
        from asva_amd import precision as P

        P.set_split(True)
    dev = torch.device("cuda", 0)
    unet = bench.build_unet(dev, 0, 1)
    lat, text, audio, null_audio = bench.synthetic_clip(dev, 1000, n=a.clips)
    ops.RECORD_KEYS = {}
    eng = DenoiseEngine(unet, DDIMScheduler(), 4.0, use_graph=True)
    eng.set_conditioning(text, audio, null_audio, audio_segment_mask(12), 12)
    x = lat.clone()
    eng.prepare(x, 50)
    unet.denoise_forward(x, torch.full((1,), 501.0, device=dev), rep=2)        # eager: records the keys (capture would record them thrice)
    keys = ops.RECORD_KEYS
    ops.RECORD_KEYS = None
    table = ops.tile_cache()

    def step_ms(reps):
        eng._graph = None                      # re-capture with the current table
        for i in range(3):
            eng.step(x, i)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                eng.step(x, i % 50)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        x.copy_(lat)
        return best

    base = step_ms(a.reps)
    noise = abs(step_ms(a.reps) - base)
    print(f"{len(keys)} table keys in one forward; step {base:.4f} ms (repeat differs by {noise * 1e3:.1f} us)", flush=True)
    modes = {int(m) for m in a.modes.split(",")}
    order = sorted((k for k in keys if keys[k]["n"] >= a.min_launches and k in table and k[0] in modes), key=lambda k: -keys[k]["n"] * keys[k]["flops"])[:a.max_keys]
    t0 = time.time()
    changed = []
    for k in order:
        inc = table[k]
        cur = base
        best_c, best_t = inc, cur
        tried = 0
        # shortlist: the asm tiles and the staple LDS-direct tiles (the isolated tuner already ranked the rest)
        # (+ every resident-convolution tile / split the geometry admits; split precision: its own short tile list)
        short = [c for c in keys[k]["cands"] if c[0] >= 40 or (c[0] in (7, 11, 13, 20, 24, 25, 30, 34, 35, 36) and c[1] <= (8 if a.split else 2))]
        if a.only_tiles:
            only = {int(v) for v in a.only_tiles.split(",")}
            short = [c for c in keys[k]["cands"] if c[0] in only]
        for c in short:
            if c == inc:
                continue
            table[k] = c
            try:
                t = step_ms(a.reps)
            except Exception as e:  # noqa: BLE001   (a candidate the geometry does not admit)
                continue
            tried += 1
            if t < best_t:
                best_c, best_t = c, t
        table[k] = inc
        gain = cur - best_t
        if best_c != inc and gain > max(2.5 * noise, 0.004):          # > 4 us per step and clear of the repeat noise
            table[k] = best_c
            t_chk = step_ms(a.reps)                                     # confirm against a fresh measurement of the incumbent state
            if t_chk < base - max(1.5 * noise, 0.002):
                changed.append((k, inc, best_c, round((base - t_chk) * 1e3, 1)))
                base = t_chk
                print(f"  {k}: {inc} -> {best_c}  step {base:.4f} ms (-{changed[-1][3]} us)  [{keys[k]['n']} launches, {tried} candidates, {time.time() - t0:.0f} s]", flush=True)
            else:
                table[k] = inc
    print(f"step {base:.4f} ms after {len(changed)} replacements over {len(order)} keys, {time.time() - t0:.0f} s")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    ops.save_tile_cache(a.out)
    for row in changed:
        print("replaced", row)


if __name__ == "__main__":
    main()
