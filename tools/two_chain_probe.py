"""Does the GPU run two INDEPENDENT UNet forwards (the two classifier-free-guidance branches as separate B=1 chains on two
streams inside one captured graph) faster than one batched B=2 forward?  Each chain's kernels are latency-bound
(SQ_WAIT_ANY ~60 %), so a second chain could fill the idle issue slots; it also halves the rows per launch and streams the
weights twice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from asva_amd.conditioning import audio_segment_mask

dev = torch.device("cuda", 0)
unet = bench.build_unet(dev, 0, 1)
lat, text, audio, null_audio = bench.synthetic_clip(dev, 1000)
mask = audio_segment_mask(12)
t = torch.full((1,), 501.0, device=dev)


def timed_graph(fn, reps=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps, out


# (a) one batched forward, B = 2
unet.set_conditioning(torch.cat([text, text]), torch.cat([null_audio, audio]), mask, 12)
ms_a, _ = timed_graph(lambda: unet.denoise_forward(lat, t, rep=2))
print(f"batched B=2 forward: {ms_a:.3f} ms")

# (b) B = 1 forward alone, then two of them on two streams
unet.set_conditioning(text, audio, mask, 12)
ms_1, _ = timed_graph(lambda: unet.denoise_forward(lat, t, rep=1))
print(f"single  B=1 forward: {ms_1:.3f} ms")
side = torch.cuda.Stream()


def two():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        o2 = unet.denoise_forward(lat, t, rep=1)
    o1 = unet.denoise_forward(lat, t, rep=1)
    cur.wait_stream(side)
    return o1, o2


ms_2, _ = timed_graph(two)
print(f"two B=1 forwards on two streams in one graph: {ms_2:.3f} ms  (sequential would be {2 * ms_1:.3f})")
