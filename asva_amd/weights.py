"""Weight packing: reference state_dict tensors -> the layouts the gfx950 kernels read.

All GEMM-class weights become bf16 [N][K] matrices (K contiguous):
  * nn.Linear weight [out, in]                      -> as is
  * 1x1 nn.Conv2d weight [out, in, 1, 1]            -> [out, in]
  * 3x3 nn.Conv2d weight [out, in, 3, 3]            -> [out, 9*cin_pad], k = (kh*3 + kw)*cin_pad + c
    (cin padded to a multiple of 8 with zeros: conv_in has 4 input channels)
  * GEGLU proj weight [2*Nh, K] (diffusers: value rows then gate rows) -> per 32-row block
    [16 value rows | 16 gate rows] so one MFMA accumulator fragment holds value and gate of the
    same output feature in the same lane (see gemm.hip epilogue)
  * output rows are padded to a multiple of 4 where the layer has fewer (conv_out: 4 -> 4 ok).
Biases and norm parameters stay f32.
"""
from __future__ import annotations

import torch

from . import precision as P


def to_act(w: torch.Tensor) -> torch.Tensor:
    """f32 values -> the 16-bit storage type.  In split-precision mode (precision.SPLIT) the result is the main plane of a
    twin allocation [2, ...]: plane 0 = round16(w), plane 1 = round16(w - plane 0); the rest plane travels with every view
    of the result (same offset in the second half of the storage)."""
    w = w.detach().float().contiguous()
    if not P.SPLIT or w.device.type == "meta":
        return w.to(P.ACT)
    n = w.numel()
    npad = (n + 7) // 8 * 8
    base = torch.zeros((2, npad), dtype=P.ACT, device=w.device)
    main = w.reshape(-1).to(P.ACT)
    base[0, :n] = main
    base[1, :n] = (w.reshape(-1) - main.float()).to(P.ACT)
    return base[0, :n].view(w.shape)


def to_planes(w: torch.Tensor):
    """f32 values -> (main, rest) as two separate 16-bit tensors, whatever the process-wide mode: the weights of a three-pass
    product under the per-layer precision plan (precision.py)"""
    w = w.detach().float().contiguous()
    main = w.to(P.ACT)
    if w.device.type == "meta":
        return main, torch.empty_like(main)
    return main, (w - main.float()).to(P.ACT)


def is_twin(t: torch.Tensor) -> bool:
    """True for a view into the first half of a twin allocation (see to_act)"""
    nb = t.untyped_storage().nbytes()
    if t.numel() == 0 or nb % 32:
        return False
    last = sum((s - 1) * st for s, st in zip(t.shape, t.stride()))
    return (t.storage_offset() + last + 1) * t.element_size() <= nb // 2


def rest_of(t: torch.Tensor) -> torch.Tensor:
    """the rest plane of a twin tensor, as a view with the same shape and strides"""
    half = t.untyped_storage().nbytes() // 2 // t.element_size()
    return torch.as_strided(t, t.shape, t.stride(), t.storage_offset() + half)


def from_act(t: torch.Tensor) -> torch.Tensor:
    """the f32 values a kernel reconstructs from a packed 16-bit tensor (main + rest in split mode)"""
    if P.SPLIT and is_twin(t):
        return t.float() + rest_of(t).float()
    return t.float()


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    return to_act(w)


def pack_conv1x1(w: torch.Tensor) -> torch.Tensor:
    return to_act(w.detach().reshape(w.shape[0], w.shape[1]))


def pack_conv3x3(w: torch.Tensor, cin_pad: int | None = None, cout_pad: int | None = None) -> torch.Tensor:
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cin_pad = cin_pad or (cin + 7) // 8 * 8
    cout_pad = cout_pad or (cout + 3) // 4 * 4
    p = torch.zeros((cout_pad, 3, 3, cin_pad), dtype=torch.float32, device=w.device)
    p[:cout, :, :, :cin] = w.detach().float().permute(0, 2, 3, 1)
    return to_act(p.reshape(cout_pad, 9 * cin_pad))


def subpixel_conv3x3(wf: torch.Tensor) -> torch.Tensor:
    """3x3 pad-1 convolution behind a nearest 2x upsample (FFSpatioTempResUpsample3D, ff_spatio_temp_resnet_3d.py:48-55) as FOUR 2x2
    convolutions on the ORIGINAL image, one per output-pixel parity (dy, dx): output pixel (2y + dy, 2x + dx) of the upsampled
    convolution sees input rows y + dy - 1 and y + dy (columns likewise), because upsampled row 2y + dy + ky - 1 is input row
    y + ((dy + ky - 1) >> 1) — so the 3 kernel rows fold onto 2 input rows with summed weights: dy = 0: {w0 | w1 + w2}, dy = 1:
    {w0 + w1 | w2}.  Exactly the same function (the sums are formed in f32 before the one rounding to storage), 4/9 of the multiplies.
    wf [cout, 3, 3, cin] f32 (kernel layout, channels last) -> [4 * cout, 2 * 2 * cin]: row (2 dy + dx) * cout + co, column
    (2 i + j) * cin + ci = the weight of input pixel (y + dy - 1 + i, x + dx - 1 + j) (AVSD_GEMM_CONV3 with ups = 2, include/avsd.h)."""
    cout, kh, kw, cin = wf.shape
    assert kh == 3 and kw == 3
    sets = (((0,), (1, 2)), ((0, 1), (2,)))            # sets[d][i]: kernel rows (columns) that land on input row (column) offset d - 1 + i
    out = torch.zeros((2, 2, cout, 2, 2, cin), dtype=torch.float32, device=wf.device)
    for dy in range(2):
        for dx in range(2):
            for i in range(2):
                for j in range(2):
                    for ky in sets[dy][i]:
                        for kx in sets[dx][j]:
                            out[dy, dx, :, i, j, :] += wf[:, ky, kx, :]
    return out.reshape(4 * cout, 4 * cin)


def geglu_row_order(nh: int) -> torch.Tensor:
    """Packed row r of a GEGLU projection with nh output features reads source row order[r]."""
    assert nh % 16 == 0, "GEGLU packing needs the inner dim to be a multiple of 16"
    blk = torch.arange(nh // 16).repeat_interleave(32)
    p = torch.arange(32).repeat(nh // 16)
    return torch.where(p < 16, blk * 16 + p, nh + blk * 16 + p - 16)


def pack_geglu(w: torch.Tensor, b: torch.Tensor | None):
    nh = w.shape[0] // 2
    order = geglu_row_order(nh).to(w.device)
    wp = to_act(w.detach()[order])
    bp = None if b is None else b.detach()[order].float().contiguous()
    return wp, bp


def pack_frag(wp: torch.Tensor) -> torch.Tensor:
    """packed 16-bit weights [N, K] -> MFMA-fragment order [N / 32, K / 16, 64, 8] (AVSD_GEMM_W_FRAG, include/avsd.h): element
    (f, s, l, e) = W[32 f + (l & 31)][16 s + 8 (l >> 5) + e] — what lane l of a wave feeds the 32x32x16 MFMA of k-step s for the
    32 columns of fragment f; one k-step of one fragment is 1 KB, contiguous (csrc/nstream.hip streams it straight into registers)"""
    N, K = wp.shape
    assert N % 32 == 0 and K % 16 == 0
    return wp.reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(N // 32, K // 16, 64, 8)


def pad_rows(w: torch.Tensor, mult: int = 4) -> torch.Tensor:
    n = w.shape[0]
    if n % mult == 0:
        return w
    pad = torch.zeros((mult - n % mult,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    return torch.cat([w, pad], 0).contiguous()
