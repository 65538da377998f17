"""Where does a level-0 layer find its input inside a step?  The temporal-mix layer (24576 x 320 x 960 + residual) timed as the CONSUMER of a
tensor another kernel has just written, per producer:
  alone      the consumer replayed back to back (what the tile tuner's hot timing sees: its input is never rewritten)
  copy       an elementwise copy rewrites the input (a different rows -> XCD mapping)
  gemm67     a 128 x 320-tile GEMM writes the input: row band b on the same XCD that reads it back (gemm4.hip's XCD-contiguous items)
  gemm13     the 256 x 160 LDS-direct tile writes it (another mapping)
  flush      400 MB streamed through the caches first (input from HBM)
Time of the consumer = (producer + consumer) pairs minus the producer alone, graph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
B, Fr, hw, C = 2, 12, 1024, 320
M = B * Fr * hw
src, y, out = rnd(M, C), rnd(M, C), torch.empty(M, C, dtype=torch.bfloat16, device=dev)
w3, w1 = rnd(C, 3 * C, sc=(3 * C) ** -0.5), rnd(C, C, sc=C ** -0.5)
big_a, big_b = torch.empty(100 << 20, dtype=torch.float32, device=dev), torch.empty(100 << 20, dtype=torch.float32, device=dev)


def seq_us(fns, reps=8):
    for f in fns:
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            for f in fns:
                f()
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    gr.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3


producers = {"copy": lambda: y.copy_(src),
             "gemm67": lambda: ops.gemm(src, w1, out=y, tile=67),
             "gemm13": lambda: ops.gemm(src, w1, out=y, tile=13),
             "gemm64": lambda: ops.gemm(src, w1, out=y, tile=64),
             "flush": lambda: big_b.copy_(big_a)}
for ct in (67, 13, 64):
    cons = lambda: ops.gemm(y, w3, res1=y, out=out, mode=ops.TMIX, tmix=(hw, Fr), tile=ct)
    cells = [f"alone {seq_us([cons]):6.1f}"]
    for name, p in producers.items():
        tp = seq_us([p])
        cells.append(f"{name} {seq_us([p, cons]) - tp:6.1f} (producer {tp:5.1f})")
    print(f"consumer tmix+res tile {ct}: " + "   ".join(cells), flush=True)
# the plain layer that follows a temporal mix (1 x 1, K = 320) and the 3 x 3 convolution reading GroupNorm's output
for ct in (67, 38, 64):
    cons = lambda: ops.gemm(y, w1, res1=src, out=out, tile=ct)
    cells = [f"alone {seq_us([cons]):6.1f}"]
    for name, p in producers.items():
        tp = seq_us([p])
        cells.append(f"{name} {seq_us([p, cons]) - tp:6.1f}")
    print(f"consumer plain+res 24576x320x320 tile {ct}: " + "   ".join(cells), flush=True)
