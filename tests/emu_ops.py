"""CPU emulation of the kernel CONTRACTS of asva_amd.ops (same signatures, torch fp32 math, bf16
storage rounding at the same points).  TEST INFRASTRUCTURE ONLY: it lets the host orchestration
(weight packing, layer sequencing, conditioning cache, scheduler glue) be checked against the oracle
in the GPU-less build container by monkeypatching `asva_amd.unet.ops`.  The product never imports it;
the real parity tests (-m gpu) run the HIP kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from asva_amd import precision as P

EMULATED = True
PLAIN, TMIX, CONV3 = 0, 1, 2
F32 = torch.float32


def gemm(a, w, *, n=None, k=None, a2=None, bias=None, rowvec=None, rows_per_vec=0, res1=None, res2=None, alpha=1.0,
         geglu=False, gelu=False, out_f32=False, out=None, mode=PLAIN, tmix=None, conv=None, m=None, tile=0, split_k=1,
         rowstats=None, ln=None, master=None, stats_pos=None, ln_pos=None, a_rest=None, a2_rest=None, w_rest=None, out_rest=None, w_frag=None):
    assert a.dtype == P.ACT and w.dtype == P.ACT
    planes = w_rest is not None       # an explicit three-pass product (per-layer precision plan): operands are main + rest
    if planes:
        assert a_rest is not None and (a2 is None) == (a2_rest is None) and out_f32 == (out_rest is None) and not (out_f32 and master is not None)
        assert all(r is None or r.dtype == F32 for r in (res1, res2))
        COUNTS["three_pass"] += 1
    else:
        assert a_rest is None and a2_rest is None
    if out_rest is not None:
        assert out is not None and not out_f32 and not geglu
    wf = w.float() + (w_rest.float() if planes else 0.0)
    if planes:
        a = a.float() + a_rest.float()
        a2 = None if a2 is None else a2.float() + a2_rest.float()
    if mode == PLAIN:
        x = a.float() if a2 is None else torch.cat([a.float(), a2.float()], 1)
    elif mode == TMIX:
        hw, frames = tmix
        C = a.shape[1]
        y = a.float().reshape(-1, frames, hw, C)
        prev = torch.cat([y[:, :1], y[:, :-1]], 1)
        x = torch.cat([y[:, :1].expand_as(y), prev, y], -1).reshape(-1, 3 * C)
    elif conv[4] == 2:
        x = None
    else:
        n_img, hs, ws, stride, ups = conv[:5]
        pad = conv[5] if len(conv) > 5 else 1
        cin = a.shape[1]
        xi = a.float().reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2)
        if ups:
            xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
        xi = F.pad(xi, (pad, 1, pad, 1))                  # top/left = pad, bottom/right: reads past the image are zero
        cols = F.unfold(xi, 3, padding=0, stride=stride)
        ho = ((hs << ups) - 1) // stride + 1
        wo = ((ws << ups) - 1) // stride + 1
        full_w = (xi.shape[-1] - 3) // stride + 1
        cols = cols.reshape(n_img, cin * 9, -1, full_w)[:, :, :ho, :wo].reshape(n_img, cin * 9, ho * wo)                      # [n, cin*9, L] (c-major, tap-minor)
        L = cols.shape[-1]
        x = cols.reshape(n_img, cin, 9, L).permute(0, 3, 2, 1).reshape(n_img * L, 9 * cin)   # tap-major, c-minor
    if mode == CONV3 and conv[4] == 2:
        # sub-pixel form of nearest-2x upsample + 3x3 convolution (include/avsd.h, AVSD_GEMM_CONV3 with ups = 2): w [4 cout, 4 cin] holds one
        # 2x2 kernel per output-pixel parity (dy, dx); output pixel (2y + dy, 2x + dx) reads input pixels (y + dy - 1 + i, x + dx - 1 + j)
        n_img, hs, ws = conv[:3]
        cin, cout = a.shape[1], wf.shape[0] // 4
        assert wf.shape[1] == 4 * cin and a2 is None and rowvec is None and res1 is None and res2 is None and not geglu and not gelu and ln is None
        xp = F.pad(a.float().reshape(n_img, hs, ws, cin), (0, 0, 1, 1, 1, 1))
        o = torch.zeros(n_img, 2 * hs, 2 * ws, cout)
        for dy in range(2):
            for dx in range(2):
                taps = torch.cat([xp[:, dy + i:dy + i + hs, dx + j:dx + j + ws] for i in range(2) for j in range(2)], -1).reshape(-1, 4 * cin)
                par = 2 * dy + dx
                o[:, dy::2, dx::2] = (taps @ wf[par * cout:(par + 1) * cout].T).reshape(n_img, hs, ws, cout)
        x = None
        acc = alpha * o.reshape(-1, cout)
        if bias is not None:
            assert torch.equal(bias[:cout], bias[cout:2 * cout])
            bias = bias[:cout]
    else:
        assert x.shape[1] == wf.shape[1], (x.shape, wf.shape)
        acc = alpha * (x @ wf.T)
    if ln is not None:          # LayerNorm(A) folded in: W carries gamma, bias carries beta . W^T (see avsd.h)
        if ln_pos is not None:  # LayerNorm(A + pos[frame]): pos . W'^T joins the product inside the rstd scaling
            tbl, hw_, fr_ = ln_pos
            acc = acc + tbl[(torch.arange(x.shape[0]) // hw_) % fr_]
        acc = _ln_fold(acc, ln, torch.arange(x.shape[0]), x.shape[1])
    if not geglu:
        v = acc
        if bias is not None:
            v = v + bias
        if rowvec is not None:
            v = v + rowvec.repeat_interleave(rows_per_vec, 0)[: v.shape[0]]
        if gelu:
            v = F.gelu(v)
        if res1 is not None:
            v = v + res1.float()
        if res2 is not None:
            v = v + res2.float()
    else:
        if bias is not None:
            acc = acc + bias
        blk = acc.reshape(acc.shape[0], -1, 32)
        v = (blk[:, :, :16] * F.gelu(blk[:, :, 16:])).reshape(acc.shape[0], -1)
    if master is not None:
        master.copy_(v)
    if rowstats is not None:
        r = v.to(P.ACT).float()
        if stats_pos is not None:
            tbl, hw_, fr_ = stats_pos
            r = r + tbl[(torch.arange(v.shape[0]) // hw_) % fr_]
        r = r.reshape(v.shape[0], -1, 32)
        rowstats.copy_(torch.stack([r.sum(-1), (r * r).sum(-1)], -1))
    if out is not None:
        out.copy_(v.to(out.dtype))
        if out_rest is not None:        # AVSD_GEMM_OUT_REST / the second output plane of a three-pass product
            out_rest.copy_((v - out.float()).to(out_rest.dtype))
        return out
    return v if out_f32 else v.to(P.ACT)


COUNTS = {"three_pass": 0, "split_planes": 0}


def alloc_planes(shape, device):
    buf = torch.empty((2,) + tuple(shape), dtype=P.ACT, device=device)
    return buf[0], buf[1]


def split_planes(x):
    COUNTS["split_planes"] += 1
    main = x.to(P.ACT)
    return main, (x - main.float()).to(P.ACT)


def ncfhw_to_rows_planes(x, cpad, rep=1, scale=1.0):
    B, C, Fr, H, W = x.shape
    r = (x * scale).permute(0, 2, 3, 4, 1).reshape(-1, C)
    out = torch.zeros(r.shape[0], cpad)
    out[:, :C] = r
    out = out.repeat(rep, 1)
    main = out.to(P.ACT)
    return main, (out - main.float()).to(P.ACT)


def groupnorm_planes(x1, x1_rest, nb, rows_per_batch, groups, gamma, beta, eps, act):
    x = x1.float() + x1_rest.float()
    C = x.shape[1]
    y = F.group_norm(x.reshape(nb, rows_per_batch, C).permute(0, 2, 1), groups, gamma, beta, eps)
    y = (F.silu(y) if act else y).permute(0, 2, 1).reshape(-1, C)
    main = y.to(P.ACT)
    return main, (y - main.float()).to(P.ACT)


def _ln_fold(acc, ln, rows, k):
    st, colsum, eps = ln
    st = st.reshape(-1, st.shape[-2], 2)[rows]
    mean = st[..., 0].sum(-1) / k
    rstd = torch.rsqrt((st[..., 1].sum(-1) / k - mean * mean).clamp_min(0) + eps)
    return rstd[:, None] * (acc - mean[:, None] * colsum[None, :])


def to_act(x):
    return x.to(P.ACT).contiguous()


def from_act(t):
    return t.float()


def gemm_batched(a, w, *, alpha=1.0, out_f32=False, bias=None, tile=0, ln=None):
    v = alpha * torch.einsum("bmk,bnk->bmn", a.float(), w.float())
    if ln is not None:
        # `a` is a batch of row-views into the tensor the statistics belong to
        assert a.stride(-1) == 1 and a.stride(0) % a.stride(1) == 0
        rows = (torch.arange(a.shape[0])[:, None] * (a.stride(0) // a.stride(1)) + torch.arange(a.shape[1])[None, :]).reshape(-1)
        v = _ln_fold(v.reshape(-1, v.shape[-1]), ln, rows, a.shape[-1]).reshape(v.shape)
    if bias is not None:
        v = v + bias
    return v if out_f32 else v.to(P.ACT)


def linear_small_m(x, w, bias, *, act_in=False, act_out=False, out=None):
    assert x.dtype == F32 and w.dtype == P.ACT
    v = (F.silu(x) if act_in else x) @ w.float().T
    if bias is not None:
        v = v + bias
    return F.silu(v) if act_out else v


def groupnorm(x1, x2, nb, rows_per_batch, groups, gamma, beta, eps, act, out=None):
    x = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    C = x.shape[1]
    y = F.group_norm(x.reshape(nb, rows_per_batch, C).permute(0, 2, 1), groups, gamma, beta, eps)
    y = F.silu(y) if act else y
    return y.permute(0, 2, 1).reshape(-1, C).to(P.ACT)


def ln_fold(stats):
    return stats.sum(1, keepdim=True)


def layernorm(x, gamma, beta, eps=1e-5, pos=None, hw=1, frames=1, out=None):
    v = x.float()
    if pos is not None:
        f = (torch.arange(v.shape[0]) // hw) % frames
        v = v + pos[f]
    return F.layer_norm(v, (v.shape[1],), gamma, beta, eps).to(P.ACT)


def softmax_rows(s):
    return torch.softmax(s, -1).to(P.ACT)


def attention(q, k, v, *, bq, lq, lk, kv_rows, heads, q_per_kv, frames, key_index=None, scale=None, out=None, fp8=None):
    C = q.shape[1]
    d = C // heads
    scale = scale if scale is not None else d ** -0.5
    qh = q.float().reshape(bq, lq, heads, d).transpose(1, 2)
    bk = bq // q_per_kv
    kk = k.float().reshape(bk, kv_rows, -1)[..., :C]
    vv = v.float().reshape(bk, kv_rows, -1)[..., :C]
    outs = []
    for qb in range(bq):
        kb = qb // q_per_kv
        if key_index is not None:
            rows = key_index[qb % frames].long()
        else:
            rows = torch.arange(lk)
        kh = kk[kb, rows].reshape(-1, heads, d).transpose(0, 1)
        vh = vv[kb, rows].reshape(-1, heads, d).transpose(0, 1)
        p = torch.softmax(qh[qb] @ kh.transpose(1, 2) * scale, -1).to(P.ACT).float()
        outs.append((p @ vh).transpose(0, 1).reshape(lq, C))
    return torch.cat(outs, 0).to(P.ACT)


def cross_attention_block_supported(C, heads, lk_pad, M, L):
    return C == 320 and heads == 8 and lk_pad in (32, 64, 96) and M % 128 == 0 and L % 128 == 0


def cross_attention_block(h, stats, wq, q_colsum, q_bias, k, vt, lk, wo, o_bias, *, res, heads, L, q_per_kv, eps=1e-5, scale=None,
                          rowstats=None, master=None, out=None, stats_pos=None):
    """same rounding points as the fused kernel: q, P and o are rounded to the storage type"""
    M, C = h.shape
    d = C // heads
    scale = d ** -0.5 if scale is None else scale
    q = gemm(h, wq, bias=q_bias, ln=(stats, q_colsum, eps)).float().reshape(M // L, L, heads, d)
    kv = torch.arange(M // L) // q_per_kv
    kk = k[kv, :lk].float().reshape(M // L, lk, heads, d)
    vv = vt[kv, :, :lk].float().reshape(M // L, heads, d, lk)
    s = torch.einsum("blhd,bkhd->bhlk", q, kk) * scale
    pr = torch.softmax(s, -1).to(P.ACT).float()
    o = torch.einsum("bhlk,bhdk->blhd", pr, vv).reshape(M, C).to(P.ACT)
    return gemm(o, wo, bias=o_bias, res1=res, rowstats=rowstats, master=master, out=out, stats_pos=stats_pos)


def temporal_attention(qkv, *, b, frames, hw, heads, scale=None, out=None):
    C = qkv.shape[1] // 3
    d = C // heads
    x = qkv.float().reshape(b, frames, hw, 3, heads, d).permute(3, 0, 2, 4, 1, 5)
    o = F.scaled_dot_product_attention(x[0], x[1], x[2], scale=scale)
    return o.permute(0, 3, 1, 2, 4).reshape(b * frames * hw, C).to(P.ACT)


def ncfhw_to_rows(x, cpad, rep=1, scale=1.0):
    B, C, Fr, H, W = x.shape
    r = (x * scale).permute(0, 2, 3, 4, 1).reshape(-1, C)
    out = torch.zeros(r.shape[0], cpad)
    out[:, :C] = r
    return out.repeat(rep, 1).to(P.ACT)


def copy(src, dst=None, rep=1):
    out = torch.cat([src] * rep) if rep > 1 else src.clone()
    if dst is None:
        return out
    dst.copy_(out.reshape(dst.shape))
    return dst


def xattn_pack_kv(kv, n_kv, rows, C, idx, k_out, vt_out):
    kv3 = kv.view(n_kv, rows, 2 * C)
    if idx is not None:
        kv3 = kv3[:, idx.long()].reshape(n_kv * idx.shape[0], idx.shape[1], 2 * C)
    lk = kv3.shape[1]
    k_out.zero_()
    vt_out.zero_()
    k_out[:, :lk].copy_(kv3[..., :C])
    vt_out[:, :, :lk].copy_(kv3[..., C:].transpose(1, 2))


def rows_to_ncfhw(rows, B, C, Fr, H, W):
    return rows[:, :C].reshape(B, Fr, H, W, C).permute(0, 4, 1, 2, 3).contiguous()


def timestep_embedding(t, dim):
    half = dim // 2
    w = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None] * w[None]
    return torch.cat([torch.cos(a), torch.sin(a)], -1)


def guided_step(noise_pred, n_branch, g, x_in, x_out, ca, cb, *, eps_hist=None, store_slot=-1, w_cur=1.0, hist_idx=(), w=(),
                g2=0.0):
    B = x_in.shape[0]
    eps = noise_pred[:B]
    if n_branch >= 2:
        e1 = noise_pred[B:2 * B]
        eps = eps + g * (e1 - eps)
        if n_branch == 3:
            eps = eps + g2 * (noise_pred[2 * B:] - e1)
    if store_slot >= 0:
        eps_hist[store_slot] = eps
    e = w_cur * eps
    for i, wk in zip(hist_idx, w):
        e = e + wk * eps_hist[i]
    new = ca * x_in + cb * e
    new[:, :, 0] = x_in[:, :, 0]
    x_out.copy_(new)


def vae_postprocess(rows, n_img, H, W):
    return (rows[:, :3].float().reshape(n_img, H, W, 3).permute(0, 3, 1, 2) / 2 + 0.5).clamp(0, 1).contiguous()


def vae_postprocess_u8(rows, n_img, H, W):
    v = (rows[:, :3].float().reshape(n_img, H, W, 3) / 2 + 0.5).clamp(0, 1)
    return (v * 255).to(torch.uint8)


def kaldi_fbank(wave, window, mel_fb, *, shift, nfft, t_out, preemph=0.97, remove_dc=True, mean=0.0, std=1.0):
    B, n = wave.shape
    win = window.numel()
    nfr = 0 if n < win else 1 + (n - win) // shift
    out = torch.full((B, mel_fb.shape[0], t_out), (0.0 - mean) / std, dtype=F32)
    if nfr:
        fr = wave.unfold(1, win, shift)[:, :nfr].to(torch.float64)
        if remove_dc:
            fr = fr - fr.mean(-1, keepdim=True)
        prev = torch.cat([fr[..., :1], fr[..., :-1]], -1)
        fr = (fr - preemph * prev) * window.to(torch.float64)
        spec = torch.fft.rfft(fr, n=nfft).abs() ** 2
        mel = torch.clamp(spec @ mel_fb.to(torch.float64).T, min=1.1920928955078125e-07).log()
        k = min(nfr, t_out)
        out[:, :, :k] = ((mel[:, :k] - mean) / std).transpose(1, 2).float()
    return out


def patchify(x, kh, kw, stride):
    B, Cc = x.shape[:2]
    cols = F.unfold(x, (kh, kw), stride=stride)                  # [B, C*kh*kw, L]
    return cols.transpose(1, 2).reshape(-1, Cc * kh * kw).to(P.ACT)


def vit_tokens(patches, cls, pos, b, tail_rows=0):
    n_p, Cc = patches.shape[0] // b, patches.shape[1]
    x = patches.float().reshape(b, n_p, Cc) + pos[1:]
    c = (cls.reshape(1, 1, Cc) + pos[:1]).expand(b, 1, Cc)
    return torch.cat([c, x, torch.zeros(b, tail_rows, Cc)], 1).reshape(-1, Cc).to(P.ACT)
