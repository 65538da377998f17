import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import filled_unet, load_golden
from asva_amd import ops
import asva_amd.unet as U

g = load_golden("unet_tiny_e2e.pt")
unet = filled_unet(g["config"]).to("cuda")
B, Fr = g["sample"].shape[0], g["sample"].shape[2]
text = g["text"][:, None].expand(B, Fr, -1, -1).cuda(); audio = g["audio"][:, None].expand(B, Fr, -1, -1).cuda(); mask = g["mask"][None].expand(B, -1, -1)
x = g["sample"].cuda()
unet(x, 501, text, audio, audio_attention_mask=mask); torch.cuda.synchronize()

arena = [torch.zeros(256 << 20, dtype=torch.uint8, device="cuda") for _ in range(2)]
log = [[], []]
cur = [0]
off = [0]
class Traced:
    def __getattr__(self, name):
        fn = getattr(ops, name)
        if not callable(fn): return fn
        def w(*a, **k):
            out = fn(*a, **k)
            if torch.is_tensor(out) and out.is_contiguous():
                b = out.view(-1).view(torch.uint8)
                n = b.numel()
                arena[cur[0]][off[0]:off[0] + n].copy_(b)
                log[cur[0]].append((name, tuple(out.shape), str(out.dtype), off[0], n, out.data_ptr(),
                                    {kk: vv for kk, vv in k.items() if not torch.is_tensor(vv)},
                                    [(tuple(t.shape), t.data_ptr() % 256) for t in list(a) + list(k.values()) if torch.is_tensor(t)]))
                off[0] += (n + 255) // 256 * 256
            return out
        return w
U.ops = Traced()
outs = []
for r in range(2):
    cur[0] = r; off[0] = 0
    outs.append(unet(x, 501).sample)      # keep alive like the previous script did
    torch.cuda.synchronize()
print("final equal:", torch.equal(outs[0], outs[1]), "ops:", len(log[0]), len(log[1]))
for i, (a, b) in enumerate(zip(log[0], log[1])):
    sa = arena[0][a[3]:a[3] + a[4]]; sb = arena[1][b[3]:b[3] + b[4]]
    if not torch.equal(sa, sb):
        nd = int((sa != sb).sum())
        print(f"first differing op #{i}: {a[0]} {a[1]} {a[2]} kwargs={a[6]} bytes differing {nd}/{a[4]}")
        print("  ptrs:", hex(a[5]), hex(b[5]), " inputs(run A):", a[7], " inputs(run B):", b[7])
        # where do they differ?
        if a[2] == "torch.bfloat16":
            ta = sa.view(torch.bfloat16).view(a[1]).float(); tb = sb.view(torch.bfloat16).view(a[1]).float()
        else:
            ta = sa.view(torch.float32).view(a[1]); tb = sb.view(torch.float32).view(a[1])
        d = (ta != tb)
        rows = d.any(-1).nonzero().flatten()
        cols = d.any(0).nonzero().flatten() if d.dim() == 2 else None
        print("  rows differing:", rows[:20].tolist(), "n", rows.numel(), " cols:", None if cols is None else (cols[:20].tolist(), cols.numel()))
        print("  max abs diff", float((ta - tb).abs().max()), "prev op:", log[0][i-1][:3] if i else None)
        break
else:
    print("no per-op difference")
