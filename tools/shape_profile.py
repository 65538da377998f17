"""Per-(family, shape) time of one instrumented eager UNet step (HIP events around every launch)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from asva_amd import ops
from asva_amd.conditioning import audio_segment_mask
from asva_amd.engine import DenoiseEngine
from asva_amd.schedulers import DDIMScheduler

dev = torch.device("cuda", 0)
unet = bench.build_unet(dev, 0, 1)
lat, text, audio, null_audio = bench.synthetic_clip(dev, 1000)
eng = DenoiseEngine(unet, DDIMScheduler(), 4.0, use_graph=False)
eng.set_conditioning(text, audio, null_audio, audio_segment_mask(12), 12)
eng.prepare(lat, 50)
for i in range(3): eng.step(lat, i)
recs = []
orig_stop = ops.KernelTimer.stop
class T(ops.KernelTimer):
    pass
timer = ops.KernelTimer()
# capture shapes by wrapping gemm
orig_gemm = ops.gemm
shapes = []
def gemm(a, w, **k):
    out = orig_gemm(a, w, **k)
    mode = k.get("mode", 0)
    K = {0: a.shape[1] + (k["a2"].shape[1] if k.get("a2") is not None else 0), 1: 3 * a.shape[1], 2: 9 * a.shape[1]}[mode]
    shapes.append((("plain", "tmix", "conv3")[mode] + ("+geglu" if k.get("geglu") else "") + ("+res" if k.get("res1") is not None else ""), out.shape[0], w.shape[0], K))
    return out
import asva_amd.unet as U
class P:
    def __getattr__(self, n):
        return gemm if n == "gemm" else getattr(ops, n)
U.ops = P()
for rep in range(3):
    shapes.clear()
    timer.records.clear()
    ops.set_timer(timer)
    unet.denoise_forward(lat, torch.full((1,), 501.0, device=dev), rep=2)
    ops.set_timer(None)
torch.cuda.synchronize()
g = [r for r in timer.records if r[0].startswith("gemm") ]
# batched gemm calls are not in `shapes` (gemm_batched) -> align by skipping them
it = iter(shapes)
agg = collections.OrderedDict()
gi = 0
for r in timer.records:
    if not r[0].startswith("gemm"): continue
    ms = r[3].elapsed_time(r[4])
    if r[0] == "gemm_plain" and abs(r[1] - 0) >= 0:
        pass
    try_shape = None
    agg_key = None
    # match flops to next shape; batched ones won't match
    peek = shapes[gi] if gi < len(shapes) else None
    if peek is not None and abs(2.0 * peek[1] * peek[2] * peek[3] - r[1]) < 1:
        agg_key = peek; gi += 1
    else:
        agg_key = ("batched", 0, 0, int(r[1]))
    d = agg.setdefault(agg_key, [0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += r[1]
tot = sum(v[1] for v in agg.values())
print(f"{'kind':22s} {'M':>6s} {'N':>6s} {'K':>6s} {'n':>4s} {'ms':>8s} {'us/launch':>10s} {'TF/s':>7s} {'tile':>8s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    mode = {"plain": 0, "tmix": 1, "conv3": 2}.get(k[0].split("+")[0], 0)
    tile = [vv for kk, vv in ops.tile_cache().items() if kk[0] == mode and kk[1:4] == k[1:4]]
    print(f"{k[0]:22s} {k[1]:6d} {k[2]:6d} {k[3]:6d} {v[0]:4d} {v[1]:8.3f} {v[1] / v[0] * 1e3:10.1f} {v[2] / v[1] / 1e9:7.1f} {str(tile[:2]):>8s}")
print("total gemm ms", tot)
