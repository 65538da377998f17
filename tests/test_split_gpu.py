"""Split-precision ("x2") kernels: every entry point of the mode against an fp64 PyTorch statement of the same op evaluated on
the values the kernel reconstructs from its (main, rest) planes.  The planes carry 16 significant bits (bf16 library), products
run as three MFMA passes, so the tolerances are two orders below the single-plane ones (tests/test_ops_gpu.py):
  * 16-bit (two-plane) outputs: 3e-5 — the re-split of the result alone is 2^-17 / sqrt(3) ~ 4.4e-6 RMS
  * f32 outputs: 2e-5 — the dropped rest.rest term (2^-18 per product) plus f32 summation order
Reference call sites as in tests/test_ops_gpu.py; scripts/animation_gen.py:43-44 is why the mode exists (the reference runs fp32).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL16 = 3e-5
TOL32 = 2e-5


def dev():
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rndf(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


@pytest.fixture(scope="module")
def ops():
    from asva_amd import ops as _ops, precision as P

    P.set_precision("bf16")
    P.set_split(True)
    yield _ops
    P.set_split(False)


def act(ops, *shape, seed=0, scale=1.0):
    """random split tensor + the f64 values it holds"""
    t = ops.to_act(rndf(*shape, seed=seed, scale=scale))
    return t, ops.from_act(t).double()


def test_split_roundtrip_carries_16_bits(ops):
    x = rndf(1000, 328, seed=1)
    t = ops.to_act(x)
    assert t.dtype == torch.bfloat16 and t.shape == x.shape
    v = ops.from_act(t)
    assert rel_l2(v, x) < 2.0 ** -17          # vs 2^-9 for one bf16 plane
    assert rel_l2(t, x) > 1e-3                # the main plane alone is plain bf16
    # views carry their rest plane
    assert torch.equal(ops.from_act(t[100:200, 8:16]), v[100:200, 8:16])
    with pytest.raises(ValueError):
        ops._lo(torch.zeros(8, 8, dtype=torch.bfloat16, device=dev()))      # not a twin allocation


@pytest.mark.parametrize("tile", [7, 11, 13, 24, 25, 34, 35, 36, 0])
@pytest.mark.parametrize("M,N,K", [(384, 320, 320), (1000, 640, 1280), (128, 64, 64), (77, 132, 200), (2048, 1280, 768)])
def test_gemm_plain(ops, tile, M, N, K):
    a, av = act(ops, M, K, seed=1)
    w, wv = act(ops, N, K, seed=2, scale=K ** -0.5)
    bias = rndf(N, seed=3)
    res, rv = act(ops, M, N, seed=4)
    ref = av @ wv.T + bias.double() + rv
    out = ops.gemm(a, w, bias=bias, res1=res, tile=tile)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    assert rel_l2(ops.from_act(out), ref) < TOL16
    assert rel_l2(out, ref) > 5e-4            # ... and the main plane alone would not do
    out32 = ops.gemm(a, w, bias=bias, res1=res, out_f32=True, tile=tile)
    assert rel_l2(out32, ref) < TOL32


def test_gemm_asymmetric_transpose_detect(ops):
    M = N = K = 128
    a = ops.to_act(torch.eye(M, device=dev()))
    wf = (torch.arange(N * K, device=dev()).reshape(N, K) % 6553).float() / 16.0       # needs both planes
    w = ops.to_act(wf)
    for tile in (7, 11, 13, 24, 25, 34, 35, 36):
        out = ops.gemm(a, w, out_f32=True, tile=tile)
        assert torch.equal(out, ops.from_act(w).T.contiguous())


def test_gemm_two_residuals_rowvec_alpha_strided(ops):
    M, N, K = 768, 320, 640
    big, bigv = act(ops, M, 3 * K, seed=5)
    a, av = big[:, K:2 * K], bigv[:, K:2 * K]
    w, wv = act(ops, N, K, seed=6, scale=K ** -0.5)
    r1, r1v = act(ops, M, N, seed=7)
    r2f = rndf(M, N, seed=8)                                   # an f32 residual next to a split one
    rpv = 192
    rvec = rndf(M // rpv, N + 64, seed=9)[:, 32:32 + N]
    out = ops.gemm(a, w, res1=r1, res2=r2f, rowvec=rvec, rows_per_vec=rpv, alpha=0.5)
    ref = 0.5 * (av @ wv.T) + r1v + r2f.double() + rvec.double().repeat_interleave(rpv, 0)
    assert rel_l2(ops.from_act(out), ref) < TOL16


@pytest.mark.parametrize("tile,split", [(0, 1), (11, 1), (25, 4), (7, 2), (24, 8)])
def test_gemm_two_source_and_splitk(ops, tile, split):
    M, N, K1, K2 = 384, 1280, 1280, 640
    a1, a1v = act(ops, M, K1, seed=1)
    a2, a2v = act(ops, M, K2, seed=2)
    w, wv = act(ops, N, K1 + K2, seed=3, scale=(K1 + K2) ** -0.5)
    bias = rndf(N, seed=4)
    res, rv = act(ops, M, N, seed=5)
    ref = torch.cat([a1v, a2v], 1) @ wv.T + bias.double() + rv
    out = ops.gemm(a1, w, a2=a2, bias=bias, res1=res, tile=tile, split_k=split)
    assert rel_l2(ops.from_act(out), ref) < TOL16


@pytest.mark.parametrize("M,N,K", [(512, 2560, 320), (200, 640, 320)])
def test_gemm_geglu(ops, M, N, K):
    from asva_amd.weights import pack_geglu

    a, av = act(ops, M, K, seed=1)
    wf, b = rndf(N, K, seed=2, scale=K ** -0.5), rndf(N, seed=3)
    wp, bp = pack_geglu(wf, b)
    wv = torch.cat([ops.from_act(wp).double().reshape(-1, 2, 16, K)[:, 0].reshape(-1, K),
                    ops.from_act(wp).double().reshape(-1, 2, 16, K)[:, 1].reshape(-1, K)], 0)   # value rows, then gate rows
    h = av @ wv.T + b.double()
    ref = h[:, :N // 2] * F.gelu(h[:, N // 2:])
    out = ops.gemm(a, wp, bias=bp, geglu=True)
    assert out.shape == (M, N // 2)
    assert rel_l2(ops.from_act(out), ref) < TOL16 * 2       # erf from a 1.5e-7 polynomial


def test_gemm_layernorm_fold_and_rowstats(ops):
    """the producer writes (sum, sumsq) of main + rest per 32 columns; the consumer folds LayerNorm(h) into its epilogue"""
    from asva_amd.unet import Packer, _Affine

    M, C, N = 640, 320, 960
    x, xv = act(ops, M, C, seed=1)
    w0, w0v = act(ops, C, C, seed=2, scale=C ** -0.5)
    stats = torch.empty(M, C // 32, 2, device=dev())
    h = ops.gemm(x, w0, rowstats=stats)
    hv = ops.from_act(h).double()
    blk = hv.reshape(M, C // 32, 32)
    assert rel_l2(stats[..., 0], blk.sum(-1)) < 1e-5 and rel_l2(stats[..., 1], (blk * blk).sum(-1)) < 1e-5
    norm = _Affine(C)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.1 * rndf(C, seed=3).cpu())
        norm.bias.copy_(0.1 * rndf(C, seed=4).cpu())
    norm = norm.to(dev())
    w1 = rndf(N, C, seed=5, scale=C ** -0.5)
    b1 = rndf(N, seed=6)
    wf, cs, cb = Packer.lnfold(w1, norm, b1)
    out = ops.gemm(h, wf, bias=cb, ln=(stats, cs, 1e-5))
    ln = F.layer_norm(hv, (C,), norm.weight.double(), norm.bias.double(), 1e-5)
    # the folded weight is rounded to 16 bits AFTER the gain went in: compare with the weight the kernel multiplies by
    g = norm.weight.double()
    ref = ((hv - hv.mean(-1, keepdim=True)) * torch.rsqrt(hv.var(-1, unbiased=False, keepdim=True) + 1e-5)) @ ops.from_act(wf).double().T \
        + (w1.double() @ norm.bias.double() + b1.double())
    assert rel_l2(ops.from_act(out), ref) < 5e-5
    assert rel_l2(ops.from_act(out), ln @ w1.double().T + b1.double()) < 2e-4     # vs the unfolded statement: the 16-bit weights differ
    del g


@pytest.mark.parametrize("tile", [0, 11, 25, 34, 36])
def test_gemm_tmix(ops, tile):
    B, Fr, hw, C = 2, 4, 48, 320
    M = B * Fr * hw
    y, yv = act(ops, M, C, seed=1)
    w, wv = act(ops, C, 3 * C, seed=2, scale=(3 * C) ** -0.5)
    bias = rndf(C, seed=3)
    res, rv = act(ops, M, C, seed=4)
    y4 = yv.reshape(B, Fr, hw, C)
    prev = torch.cat([y4[:, :1], y4[:, :-1]], 1)
    x = torch.cat([y4[:, :1].expand_as(y4), prev, y4], -1).reshape(M, 3 * C)
    ref = x @ wv.T + bias.double() + yv + rv
    out = ops.gemm(y, w, bias=bias, res1=y, res2=res, mode=ops.TMIX, tmix=(hw, Fr), tile=tile)
    assert rel_l2(ops.from_act(out), ref) < TOL16


@pytest.mark.parametrize("stride,ups", [(1, 0), (2, 0), (1, 1)])
@pytest.mark.parametrize("cin,cout", [(320, 320), (8, 320), (640, 320)])
def test_gemm_conv3(ops, stride, ups, cin, cout):
    from asva_amd.weights import pack_conv3x3

    n_img, hs, ws = 3, 16, 8
    x, xv = act(ops, n_img * hs * ws, cin, seed=1)
    wf = rndf(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    b = rndf(cout, seed=3)
    wp = pack_conv3x3(wf)
    wv = ops.from_act(wp).double().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    xi = xv.reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2)
    if ups:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xi, wv, b.double(), stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    out = ops.gemm(x, wp, bias=b, mode=ops.CONV3, conv=(n_img, hs, ws, stride, ups))
    assert out.shape == ref.shape
    assert rel_l2(ops.from_act(out), ref) < TOL16


def test_gemm_batched_shared_weight(ops):
    B, Mfull, L, C = 3, 96, 32, 320
    h, hv = act(ops, B * Mfull, C, seed=1)
    w, wv = act(ops, 2 * C, C, seed=2, scale=C ** -0.5)
    out = ops.gemm_batched(h.view(B, Mfull, C)[:, :L], w.unsqueeze(0).expand(B, 2 * C, C))
    ref = hv.reshape(B, Mfull, C)[:, :L] @ wv.T
    assert out.shape == (B, L, 2 * C)
    assert rel_l2(ops.from_act(out), ref) < TOL16


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("two", [False, True])
@pytest.mark.parametrize("actv", [False, True])
@pytest.mark.parametrize("rows,c1", [(768, 320), (192, 1280), (192, 640), (64, 960)])
def test_groupnorm(ops, two, actv, rows, c1, fused, monkeypatch):
    monkeypatch.setattr(ops, "_GN_FUSED", fused)
    nb, c2, G = 2, (320 if two else 0), 32
    x1, v1 = act(ops, nb * rows, c1, seed=1)
    x2, v2 = act(ops, nb * rows, c2, seed=2) if two else (None, None)
    C = c1 + c2
    g, b = 1.0 + 0.1 * rndf(C, seed=3), 0.1 * rndf(C, seed=4)
    out = ops.groupnorm(x1, x2, nb, rows, G, g, b, 1e-5, actv)
    xv = v1 if not two else torch.cat([v1, v2], 1)
    ref = F.group_norm(xv.reshape(nb, rows, C).permute(0, 2, 1), G, g.double(), b.double(), 1e-5)
    ref = (F.silu(ref) if actv else ref).permute(0, 2, 1).reshape(-1, C)
    assert rel_l2(ops.from_act(out), ref) < TOL16


@pytest.mark.parametrize("C", [320, 640, 1280])
def test_layernorm_with_positions(ops, C):
    Fr, hw = 4, 24
    M = 2 * Fr * hw
    x, xv = act(ops, M, C, seed=1)
    g, b = 1.0 + 0.1 * rndf(C, seed=2), 0.1 * rndf(C, seed=3)
    pos = rndf(Fr, C, seed=4)
    out = ops.layernorm(x, g, b, pos=pos, hw=hw, frames=Fr)
    f = (torch.arange(M, device=dev()) // hw) % Fr
    ref = F.layer_norm(xv + pos.double()[f], (C,), g.double(), b.double(), 1e-5)
    assert rel_l2(ops.from_act(out), ref) < TOL16


def _sdpa(q, k, v, heads, scale):
    Bq, Lq, C = q.shape
    d = C // heads
    qh = q.reshape(Bq, Lq, heads, d).transpose(1, 2)
    kh = k.reshape(k.shape[0], -1, heads, d).transpose(1, 2)
    vh = v.reshape(v.shape[0], -1, heads, d).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    return (p @ vh).transpose(1, 2).reshape(Bq, Lq, C)


@pytest.mark.parametrize("d", [40, 80, 160, 64, 128])
def test_first_frame_attention(ops, d):
    heads, Fr, B, L = 8, 3, 2, 200
    C = heads * d
    q, qv = act(ops, B * Fr * L, C, seed=1)
    kv, kvv = act(ops, B * L, 2 * C, seed=2)
    out = ops.attention(q, kv[:, :C], kv[:, C:], bq=B * Fr, lq=L, lk=L, kv_rows=L, heads=heads, q_per_kv=Fr, frames=Fr)
    k3 = kvv[:, :C].reshape(B, L, C).repeat_interleave(Fr, 0)
    v3 = kvv[:, C:].reshape(B, L, C).repeat_interleave(Fr, 0)
    ref = _sdpa(qv.reshape(B * Fr, L, C), k3, v3, heads, d ** -0.5).reshape(-1, C)
    assert rel_l2(ops.from_act(out), ref) < TOL16 * 1.5


def test_gathered_cross_attention(ops):
    from asva_amd.conditioning import audio_segment_mask, mask_to_key_index

    heads, d, Fr, B, L, rows = 8, 40, 12, 2, 64, 229
    C = heads * d
    mask = audio_segment_mask(Fr)
    idx = mask_to_key_index(mask).to(dev())
    q, qv = act(ops, B * Fr * L, C, seed=1)
    kv, kvv = act(ops, B * rows, 2 * C, seed=2)
    out = ops.attention(q, kv[:, :C], kv[:, C:], bq=B * Fr, lq=L, lk=idx.shape[1], kv_rows=rows, heads=heads, q_per_kv=Fr,
                        frames=Fr, key_index=idx)
    k3 = kvv[:, :C].reshape(B, rows, C)
    v3 = kvv[:, C:].reshape(B, rows, C)
    refs = []
    for b in range(B):
        for f in range(Fr):
            sel = idx[f].long()
            refs.append(_sdpa(qv.reshape(B * Fr, L, C)[b * Fr + f:b * Fr + f + 1], k3[b:b + 1, sel], v3[b:b + 1, sel], heads, d ** -0.5))
    assert rel_l2(ops.from_act(out), torch.cat(refs, 0).reshape(-1, C)) < TOL16 * 1.5


def test_vae_wide_head_attention(ops):
    n, L, C = 2, 320, 512
    q, qv = act(ops, n * L, C, seed=1)
    k, kv_ = act(ops, n * L, C, seed=2)
    v, vv = act(ops, n * L, C, seed=3)
    out = ops.attention(q, k, v, bq=n, lq=L, lk=L, kv_rows=L, heads=1, q_per_kv=1, frames=1, scale=C ** -0.5)
    ref = _sdpa(qv.reshape(n, L, C), kv_.reshape(n, L, C), vv.reshape(n, L, C), 1, C ** -0.5).reshape(-1, C)
    assert rel_l2(ops.from_act(out), ref) < TOL16 * 1.5


@pytest.mark.parametrize("d", [40, 80, 160])
def test_temporal_attention(ops, d):
    heads, Fr, B, hw = 8, 12, 2, 16
    C = heads * d
    qkv, v = act(ops, B * Fr * hw, 3 * C, seed=1)
    out = ops.temporal_attention(qkv, b=B, frames=Fr, hw=hw, heads=heads)
    x = v.reshape(B, Fr, hw, 3 * C).permute(0, 2, 1, 3).reshape(B * hw, Fr, 3 * C)
    ref = _sdpa(x[..., :C], x[..., C:2 * C], x[..., 2 * C:], heads, d ** -0.5)
    ref = ref.reshape(B, hw, Fr, C).permute(0, 2, 1, 3).reshape(-1, C)
    assert rel_l2(ops.from_act(out), ref) < TOL16


def test_small_kernels(ops):
    # skinny linear with split weights
    x = rndf(4, 1280, seed=1)
    w, wv = act(ops, 640, 1280, seed=2, scale=1280 ** -0.5)
    b = rndf(640, seed=3)
    out = ops.linear_small_m(x, w, b, act_in=True, act_out=True)
    ref = F.silu(F.silu(x.double()) @ wv.T + b.double())
    assert rel_l2(out, ref) < 1e-5
    # latents -> rows, replicated, padded channels zero
    lat = rndf(1, 4, 3, 8, 8, seed=4)
    rows = ops.ncfhw_to_rows(lat, cpad=8, rep=2, scale=0.5)
    v = ops.from_act(rows)
    ref = (0.5 * lat).permute(0, 2, 3, 4, 1).reshape(-1, 4)
    assert rel_l2(v[:192, :4], ref) < 2.0 ** -17 and torch.equal(v[:192], v[192:]) and not v[:, 4:].any()
    # replication moves both planes
    t, tv = act(ops, 64, 320, seed=5)
    r = ops.copy(t, rep=3)
    assert torch.equal(ops.from_act(r), tv.float().repeat(3, 1))
    # VAE post-processing reads both planes
    img, iv = act(ops, 2 * 16, 8, seed=6)
    out = ops.vae_postprocess(img, 2, 4, 4)
    ref = (iv[:, :3].reshape(2, 16, 3).permute(0, 2, 1).reshape(2, 3, 4, 4) * 0.5 + 0.5).clamp(0, 1)
    assert rel_l2(out, ref) < 1e-6
    u8 = ops.vae_postprocess_u8(img, 2, 4, 4)
    assert (u8.float() - (ref.permute(0, 2, 3, 1) * 255).floor().float()).abs().max() <= 1


# ---- the exact-f32 yardstick (avsd_gemm_f32, v_mfma_f32_32x32x2_f32) and what it says about the other two GEMMs ------------
@pytest.mark.parametrize("M,N,K", [(384, 320, 320), (1000, 644, 1284), (2048, 1280, 768), (130, 36, 40)])
def test_f32_yardstick_gemm_is_exact_f32(ops, M, N, K):
    a, w, b = rndf(M, K, seed=1), rndf(N, K, seed=2, scale=K ** -0.5), rndf(N, seed=3)
    out = ops.gemm_f32(a, w, b)
    ref = a.double() @ w.double().T + b.double()
    assert rel_l2(out, ref) < 1e-6          # f32 round-off of a K-long fmaf chain (measured 1-5e-7 for K = 40 .. 1284)


def test_three_gemms_on_the_same_f32_operands(ops):
    """storage rounding vs kernel arithmetic, separated: f32 operands through (1) the exact-f32 MFMA GEMM, (2) the split-precision
    GEMM (operands rounded to 16 bits, three passes), (3) the plain bf16 GEMM (operands rounded to 8 bits)"""
    from asva_amd import precision as P

    M, N, K = 1536, 1280, 1280
    a, w = rndf(M, K, seed=1), rndf(N, K, seed=2, scale=K ** -0.5)
    ref = a.double() @ w.double().T
    e_f32 = rel_l2(ops.gemm_f32(a, w), ref)
    e_x2 = rel_l2(ops.gemm(ops.to_act(a), ops.to_act(w), out_f32=True), ref)
    P.set_split(False)
    try:
        e_bf16 = rel_l2(ops.gemm(a.bfloat16(), w.bfloat16(), out_f32=True), ref)
    finally:
        P.set_split(True)
    print(f"GEMM {M}x{N}x{K} on f32 operands vs fp64: exact-f32 MFMA {e_f32:.2e}, split precision {e_x2:.2e}, bf16 {e_bf16:.2e}")
    assert e_f32 < 1e-6 and e_x2 < 1.5e-5 and 1e-3 < e_bf16 < 5e-3
