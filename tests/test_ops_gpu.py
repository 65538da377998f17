"""Per-kernel parity: every C-ABI entry point against a plain PyTorch fp32 statement of the same op on
the same bf16-representable inputs.  Tolerances (rel-L2 over the whole output):
  * bf16 outputs: 4e-3  (bf16 rounding of the result alone is ~1.1e-3 RMS; attention adds the bf16
    rounding of P before P.V)
  * f32 outputs:  2e-5  (same products, different f32 summation order)
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_BF16 = 4e-3
TOL_F32 = 2e-5


def dev():
    return torch.device("cuda:0")


def rel_l2(a, b):
    a = a.float()
    b = b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev())


def rndf(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


@pytest.fixture(scope="module")
def ops():
    from asva_amd import ops as _ops

    return _ops


def test_device_is_gfx950():
    import ctypes

    from asva_amd import _lib

    name = ctypes.create_string_buffer(64)
    cu = ctypes.c_int(0)
    rc = _lib.lib().avsd_device_info(name, 64, ctypes.byref(cu))
    assert rc == 0, _lib.lib().avsd_last_error()
    assert name.value.startswith(b"gfx950") and cu.value > 0


# ---- GEMM ------------------------------------------------------------------------------------------
# tile ids built into the library (csrc/gemm.hip dispatch_tile): 1-3 register-staged, the rest LDS-direct; 0 = table / rule
V1_TILES = [1, 2, 3]
V2_TILES = [4, 6, 7, 9, 11, 12, 13, 14, 17, 19, 20, 24, 25, 30, 31, 38]
@pytest.mark.parametrize("tile", V1_TILES + V2_TILES + [0])
@pytest.mark.parametrize("M,N,K", [(384, 320, 320), (1000, 640, 1280), (128, 64, 64), (77, 132, 200), (2048, 1280, 768)])
def test_gemm_plain(ops, tile, M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias = rndf(N, seed=3)
    res = rnd(M, N, seed=4)
    out = ops.gemm(a, w, bias=bias, res1=res, tile=tile)
    ref = a.float() @ w.float().T + bias + res.float()
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    assert rel_l2(out, ref) < TOL_BF16
    out32 = ops.gemm(a, w, bias=bias, res1=res, out_f32=True, tile=tile)
    assert rel_l2(out32, ref) < TOL_F32


def test_gemm_asymmetric_transpose_detect(ops):
    # A = I with an asymmetric W catches a swapped row/col in the accumulator layout
    M = N = K = 128
    a = torch.eye(M, dtype=torch.bfloat16, device=dev())
    w = (torch.arange(N * K, device=dev()).reshape(N, K) % 251).to(torch.bfloat16)
    for tile in V1_TILES + V2_TILES:
        out = ops.gemm(a, w, out_f32=True, tile=tile)
        assert torch.equal(out, w.float().T.contiguous())


def test_gemm_strided_views_two_residuals_rowvec_alpha(ops):
    M, N, K = 768, 320, 640
    big = rnd(M, 3 * K, seed=5)
    a = big[:, K:2 * K]                      # lda = 3K
    w = rnd(N, K, seed=6, scale=K ** -0.5)
    r1, r2 = rnd(M, N, seed=7), rnd(M, N, seed=8)
    rows_per_vec = 192
    rv = rndf(M // rows_per_vec, N + 64, seed=9)[:, 32:32 + N]   # ldv = N + 64, offset view
    out = ops.gemm(a, w, res1=r1, res2=r2, rowvec=rv, rows_per_vec=rows_per_vec, alpha=0.5)
    ref = 0.5 * (a.float() @ w.float().T) + r1.float() + r2.float() + rv.repeat_interleave(rows_per_vec, 0)
    assert rel_l2(out, ref) < TOL_BF16


@pytest.mark.parametrize("tile", [0, 2, 4, 7, 9, 19, 20, 24, 25])
@pytest.mark.parametrize("M,N,K1,K2", [(512, 320, 640, 320), (8, 160, 160, 160), (200, 80, 160, 80), (1000, 1280, 1280, 640)])
def test_gemm_two_source_concat(ops, tile, M, N, K1, K2):
    a1, a2 = rnd(M, K1, seed=1), rnd(M, K2, seed=2)
    w = rnd(N, K1 + K2, seed=3, scale=(K1 + K2) ** -0.5)
    out = ops.gemm(a1, w, a2=a2, out_f32=True, tile=tile)
    ref = torch.cat([a1, a2], 1).float() @ w.float().T
    assert rel_l2(out, ref) < TOL_F32


@pytest.mark.parametrize("tile", V1_TILES + [t for t in V2_TILES if t != 38])
def test_gemm_geglu(ops, tile):
    from asva_amd.weights import pack_geglu

    M, C = 640, 320
    x = rnd(M, C, seed=1)
    w = rnd(8 * C, C, seed=2, scale=C ** -0.5)
    b = rndf(8 * C, seed=3)
    wp, bp = pack_geglu(w, b)
    out = ops.gemm(x, wp, bias=bp, geglu=True, tile=tile)
    h = x.float() @ w.float().T + b
    val, gate = h.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    assert out.shape == (M, 4 * C)
    assert rel_l2(out, ref) < TOL_BF16


def _tmix_ref(y, w, b, B, Fr, hw):
    C = y.shape[1]
    y5 = y.float().reshape(B, Fr, hw, C)
    prev = torch.clamp(torch.arange(Fr) - 1, min=0)
    cat = torch.cat([y5[:, [0] * Fr], y5[:, prev], y5], dim=-1)
    return (y5 + cat @ w.float().T + b).reshape(B * Fr * hw, C)


@pytest.mark.parametrize("tile", [0, 2] + V2_TILES)
@pytest.mark.parametrize("B,Fr,hw,C", [(2, 12, 64, 320), (1, 4, 16, 80), (2, 3, 100, 640)])
def test_gemm_tmix(ops, B, Fr, hw, C, tile):
    y = rnd(B * Fr * hw, C, seed=1)
    w = rnd(C, 3 * C, seed=2, scale=(3 * C) ** -0.5)
    b = rndf(C, seed=3)
    out = ops.gemm(y, w, bias=b, res1=y, mode=ops.TMIX, tmix=(hw, Fr), tile=tile)
    assert rel_l2(out, _tmix_ref(y, w, b, B, Fr, hw)) < TOL_BF16


@pytest.mark.parametrize("tile", [0, 1] + V2_TILES)
@pytest.mark.parametrize("stride,ups", [(1, 0), (2, 0), (1, 1)])
@pytest.mark.parametrize("n_img,hs,ws,cin,cout", [(3, 16, 16, 64, 128), (2, 8, 12, 320, 320), (4, 5, 7, 8, 4), (2, 32, 32, 4, 320)])
def test_gemm_conv3x3(ops, stride, ups, n_img, hs, ws, cin, cout, tile):
    from asva_amd.weights import pack_conv3x3

    if stride == 2 and (hs % 2 or ws % 2):
        pytest.skip("stride-2 case uses even sizes")
    cin_pad = (cin + 7) // 8 * 8
    x = torch.zeros(n_img * hs * ws, cin_pad, dtype=torch.bfloat16, device=dev())
    x[:, :cin] = rnd(n_img * hs * ws, cin, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    b = rndf(cout, seed=3)
    out = ops.gemm(x, pack_conv3x3(w, cin_pad), bias=b, mode=ops.CONV3, conv=(n_img, hs, ws, stride, ups), tile=tile)
    xi = x[:, :cin].float().reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2)
    if ups:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xi, w.float(), b, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < TOL_BF16


@pytest.mark.parametrize("tile", [40, 42, 43, 44, 48])
@pytest.mark.parametrize("n_img,hs,ws,cin,cout,split", [
    (3, 32, 32, 64, 320, 1),      # bands of image rows, one chunk
    (2, 32, 32, 320, 192, 1),     # five chunks: the A double buffer turns over; N tail for 128- / 160- / 256-wide tiles
    (5, 16, 16, 128, 132, 2),     # N not a multiple of 32, one chunk per slice
    (7, 8, 8, 192, 128, 3),       # whole images per tile, M tail (448 rows), uneven... 3 chunks / 3 slices
    (6, 4, 4, 320, 192, 2),       # M = 96 < one tile; 5 chunks over 2 slices (3 + 2)
    (3, 8, 16, 64, 64, 1),        # non-square image
    (2, 16, 32, 640, 320, 5),     # ten chunks, five slices
])
def test_gemm_conv3_resident(ops, n_img, hs, ws, cin, cout, split, tile):
    """conv3r.hip: the 3x3 stride-1 convolution with the input tile resident in LDS, every tile id the geometry admits, against
    F.conv2d in f32 and against the tap-major tile 9 (same products, other f32 order); full epilogue; split over channel chunks"""
    from asva_amd import _lib
    from asva_amd.weights import pack_conv3x3

    bm = _lib.lib().avsd_gemm_conv3r_supported(tile, hs, ws, cin)
    M = n_img * hs * ws
    x = rnd(M, cin, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    b = rndf(cout, seed=3)
    res = rnd(M, cout, seed=4)
    wp = pack_conv3x3(w)
    if bm == 0:
        with pytest.raises(RuntimeError):
            ops.gemm(x, wp, bias=b, mode=ops.CONV3, conv=(n_img, hs, ws, 1, 0), tile=tile)
        return
    assert bm % ws == 0 and ((hs * ws) % bm == 0 or bm % (hs * ws) == 0)
    xi = x.float().reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xi, w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    out = ops.gemm(x, wp, bias=b, res1=res, mode=ops.CONV3, conv=(n_img, hs, ws, 1, 0), tile=tile, split_k=split)
    assert rel_l2(out, ref + res.float()) < TOL_BF16
    o32 = ops.gemm(x, wp, bias=b, out_f32=True, mode=ops.CONV3, conv=(n_img, hs, ws, 1, 0), tile=tile, split_k=split)
    assert rel_l2(o32, ref) < TOL_F32
    assert rel_l2(o32, ops.gemm(x, wp, bias=b, out_f32=True, mode=ops.CONV3, conv=(n_img, hs, ws, 1, 0), tile=9)) < TOL_F32
    # run to run identical (no atomics, fixed slice order)
    assert torch.equal(o32, ops.gemm(x, wp, bias=b, out_f32=True, mode=ops.CONV3, conv=(n_img, hs, ws, 1, 0), tile=tile, split_k=split))


@pytest.mark.parametrize("tile", [51, 52, 53, 54])
@pytest.mark.parametrize("n_img,hs,ws,cin,cout,split", [
    (2, 64, 64, 64, 128, 1),        # 2 x 2 (x 8 rows) tiles per image: every border and corner case of the zero-filled halo
    (1, 32, 96, 128, 132, 2),       # non-square, three tiles per row band, N tail, one chunk per slice
    (3, 8, 32, 192, 320, 3),        # one tile per image (TH = 8) / two (TH = 4)
    (1, 128, 128, 64, 64, 1),       # a VAE-sized image
    (2, 12, 64, 64, 64, 1),         # height 12: only the 4-row tile fits
])
def test_gemm_conv3_resident_2d(ops, n_img, hs, ws, cin, cout, split, tile):
    """conv3r.hip, rectangular tiles (TH image rows x 32 pixels): against F.conv2d in f32 and the tap-major tile 9; full epilogue,
    split over channel chunks, a two-source input"""
    from asva_amd import _lib
    from asva_amd.weights import pack_conv3x3

    bm = _lib.lib().avsd_gemm_conv3r2d_supported(tile, hs, ws, cin)
    M = n_img * hs * ws
    x = rnd(M, cin, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    b = rndf(cout, seed=3)
    res = rnd(M, cout, seed=4)
    wp = pack_conv3x3(w)
    conv = (n_img, hs, ws, 1, 0)
    if bm == 0:
        with pytest.raises(RuntimeError):
            ops.gemm(x, wp, bias=b, mode=ops.CONV3, conv=conv, tile=tile)
        return
    xi = x.float().reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xi, w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    out = ops.gemm(x, wp, bias=b, res1=res, mode=ops.CONV3, conv=conv, tile=tile, split_k=split)
    assert rel_l2(out, ref + res.float()) < TOL_BF16
    o32 = ops.gemm(x, wp, bias=b, out_f32=True, mode=ops.CONV3, conv=conv, tile=tile, split_k=split)
    assert rel_l2(o32, ref) < TOL_F32
    assert rel_l2(o32, ops.gemm(x, wp, bias=b, out_f32=True, mode=ops.CONV3, conv=conv, tile=9)) < TOL_F32
    assert torch.equal(o32, ops.gemm(x, wp, bias=b, out_f32=True, mode=ops.CONV3, conv=conv, tile=tile, split_k=split))
    if cin >= 128:      # the same input handed over as two sources
        c1 = 64
        o2 = ops.gemm(x[:, :c1].contiguous(), wp, a2=x[:, c1:].contiguous(), bias=b, out_f32=True, mode=ops.CONV3, conv=conv, tile=tile, split_k=split)
        assert torch.equal(o2, o32)


def test_gemm_conv3_resident_refuses_other_convolutions(ops):
    from asva_amd.weights import pack_conv3x3

    x = rnd(2 * 16 * 16, 64, seed=1)
    wp = pack_conv3x3(rnd(64, 64, 3, 3, seed=2))
    for conv in [(2, 16, 16, 2, 0), (2, 16, 16, 1, 1), (2, 16, 16, 2, 0, 0)]:      # stride 2, upsample fold, asymmetric pad
        with pytest.raises(RuntimeError):
            ops.gemm(x, wp, mode=ops.CONV3, conv=conv, tile=40)
    with pytest.raises(RuntimeError):                                                # cin % 64 != 0
        ops.gemm(rnd(2 * 16 * 16, 32, seed=1), pack_conv3x3(rnd(64, 32, 3, 3, seed=2)), mode=ops.CONV3, conv=(2, 16, 16, 1, 0), tile=44)
    with pytest.raises(RuntimeError):                                                # more slices than chunks
        ops.gemm(x, wp, mode=ops.CONV3, conv=(2, 16, 16, 1, 0), tile=44, split_k=2)
    # a two-source convolution on a tile that reads one buffer (tap-major LDS-direct, asm): refused at the C ABI, not half-read
    x2, w2 = rnd(2 * 16 * 16, 128, seed=3), pack_conv3x3(rnd(64, 128, 3, 3, seed=4))
    for t in (9, 25):
        with pytest.raises((RuntimeError, ValueError), match="two-source"):
            ops.gemm(x2[:, :64].contiguous(), w2, a2=x2[:, 64:].contiguous(), mode=ops.CONV3, conv=(2, 16, 16, 1, 0), tile=t)


@pytest.mark.parametrize("tile,split", [(0, 1), (3, 1), (6, 1), (13, 1), (17, 1), (20, 1), (25, 1), (7, 2), (24, 4), (63, 1), (65, 2)])
def test_gemm_layernorm_fusion(ops, tile, split):
    """LayerNorm folded into the GEMMs around it (AVSD_GEMM_ROWSTATS / AVSD_GEMM_LNFUSE): the producer's per-32-column
    (sum, sumsq) pairs are exact for its rounded output, and consumer(raw h) == Linear(LayerNorm(h))."""
    torch.manual_seed(0)
    M, C, N = 24 * 40, 640, 1280
    a = rnd(M, C, seed=1)
    wp = (0.05 * torch.randn(C, C, device=dev())).bfloat16()
    res = rnd(M, C, seed=2) * 3 + 1.5                       # a residual stream with a clear mean
    stats = torch.empty(M, C // 32, 2, device=dev())
    h = ops.gemm(a, wp, bias=torch.randn(C, device=dev()), res1=res, rowstats=stats, tile=tile, split_k=split)
    hb = h.float().reshape(M, C // 32, 32)
    assert torch.allclose(stats[..., 0], hb.sum(-1), atol=2e-3, rtol=1e-5) and torch.allclose(stats[..., 1], (hb * hb).sum(-1), rtol=1e-5, atol=1e-3)
    # consumer: y = LN(h; gamma, beta) @ W^T + b  with W' = W * gamma, colsum = sum_k W', bias' = beta @ W^T + b
    g, be = 1 + 0.2 * torch.randn(C, device=dev()), 0.3 * torch.randn(C, device=dev())
    w = 0.05 * torch.randn(N, C, device=dev())
    b = torch.randn(N, device=dev())
    wf = (w * g).bfloat16()
    colsum = wf.float().sum(1)
    bias2 = w @ be + b
    y = ops.gemm(h, wf, bias=bias2, ln=(stats, colsum, 1e-5), tile=tile, split_k=split)
    ref = F.layer_norm(h.float(), (C,), g, be, 1e-5) @ w.T + b
    assert rel_l2(y, ref) < TOL_BF16
    # statistics pre-folded to one pair per row (avsd_ln_fold): the same bits
    folded = ops.ln_fold(stats)
    assert folded.shape == (M, 1, 2)
    assert torch.equal(y, ops.gemm(h, wf, bias=bias2, ln=(folded, colsum, 1e-5), tile=tile, split_k=split))
    # GEGLU consumer (norm3 -> ff.net.0) and a batched consumer reading a strided subset of rows (norm1 -> to_k/to_v of frame 0)
    from asva_amd.weights import pack_geglu
    w1 = 0.05 * torch.randn(2 * N, C, device=dev())
    b1 = torch.randn(2 * N, device=dev())
    wpk, bpk = pack_geglu((w1 * g), w1 @ be + b1)
    yg = ops.gemm(h, wpk, bias=bpk, geglu=True, ln=(stats, wpk.float().sum(1), 1e-5), tile=tile if split == 1 else 0)
    t = F.layer_norm(h.float(), (C,), g, be, 1e-5) @ w1.T + b1
    assert rel_l2(yg, t[:, :N] * F.gelu(t[:, N:])) < TOL_BF16
    if split == 1:
        hv = h.view(2, 12 * 40, C)[:, :40]
        yb = ops.gemm_batched(hv, wf.unsqueeze(0).expand(2, N, C), bias=bias2, ln=(stats, colsum, 1e-5), tile=tile if tile in (0, 3, 6, 13) else 0)
        refb = (F.layer_norm(hv.float(), (C,), g, be, 1e-5) @ w.T + b)
        assert rel_l2(yb, refb) < TOL_BF16


@pytest.mark.parametrize("tile,split", [(0, 1), (3, 1), (6, 1), (11, 1), (14, 1), (20, 1), (7, 2), (24, 4), (61, 1), (63, 1), (66, 1), (65, 2)])
def test_gemm_layernorm_fusion_with_frame_positions(ops, tile, split):
    """LayerNorm(h + pos[frame]) folded into the GEMMs around it (norm_temp of the temporal attention,
    ff_spatio_audio_temp_transformer_3d.py:346-356): the producer's statistics are those of (rounded h + stats_pos[frame]) while h itself
    is stored unchanged, and the consumer reads the raw h with pos . W'^T added inside the rstd scaling (ln_rowvec)."""
    torch.manual_seed(0)
    B, Fr, hw, C, N = 2, 12, 40, 640, 1920
    M = B * Fr * hw
    a = rnd(M, C, seed=1)
    wp = (0.05 * torch.randn(C, C, device=dev())).bfloat16()
    res = rnd(M, C, seed=2) * 3 + 1.5
    pos = torch.randn(Fr, C, device=dev()) * 0.7 + 0.2
    frame = (torch.arange(M, device=dev()) // hw) % Fr
    stats, plain = torch.empty(M, C // 32, 2, device=dev()), torch.empty(M, C // 32, 2, device=dev())
    bias = torch.randn(C, device=dev())
    h = ops.gemm(a, wp, bias=bias, res1=res, rowstats=stats, stats_pos=(pos, hw, Fr), tile=tile, split_k=split)
    assert torch.equal(h, ops.gemm(a, wp, bias=bias, res1=res, rowstats=plain, tile=tile, split_k=split))      # the output does not see pos
    u = (h.float() + pos[frame]).reshape(M, C // 32, 32)
    assert torch.allclose(stats[..., 0], u.sum(-1), atol=2e-3, rtol=1e-5) and torch.allclose(stats[..., 1], (u * u).sum(-1), rtol=1e-5, atol=1e-3)
    assert not torch.equal(stats, plain)
    g, be = 1 + 0.2 * torch.randn(C, device=dev()), 0.3 * torch.randn(C, device=dev())
    w = 0.05 * torch.randn(N, C, device=dev())
    wf = (w * g).bfloat16()
    colsum = wf.float().sum(1)
    bias2 = w @ be
    posw = ops.linear_small_m(pos, wf, None)
    assert rel_l2(posw, pos @ wf.float().T) < 1e-5
    y = ops.gemm(h, wf, bias=bias2, ln=(stats, colsum, 1e-5), ln_pos=(posw, hw, Fr), tile=tile, split_k=split)
    ref = F.layer_norm(h.float() + pos[frame], (C,), g, be, 1e-5) @ w.T
    assert rel_l2(y, ref) < TOL_BF16
    # and it is the LayerNorm kernel + plain GEMM path within the 16-bit rounding of the normalised tensor
    nt = ops.layernorm(h, g, be, pos=pos, hw=hw, frames=Fr)
    assert rel_l2(y, ops.gemm(nt, w.bfloat16())) < TOL_BF16
    with pytest.raises((RuntimeError, ValueError)):
        ops.gemm(h, wf, bias=bias2, ln_pos=(posw, hw, Fr), tile=tile)            # ln_rowvec without the LayerNorm fold
    with pytest.raises((RuntimeError, ValueError)):
        ops.gemm(a, wp, stats_pos=(pos, hw, Fr), tile=tile)                      # stats_pos without rowstats


@pytest.mark.parametrize("tile,split", [(4, 2), (6, 4), (7, 3), (9, 2), (14, 5), (20, 4), (25, 3), (19, 2), (24, 3), (30, 3), (31, 2), (38, 2)])
def test_gemm_split_k(ops, tile, split):
    from asva_amd.weights import pack_conv3x3

    # conv at low resolution (the shapes split-K exists for), uneven K-tile split, full epilogue
    n_img, hs, ws, cin, cout = 6, 4, 4, 320, 192
    x = rnd(n_img * hs * ws, cin, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    b = rndf(cout, seed=3)
    res = rnd(n_img * hs * ws, cout, seed=4)
    out = ops.gemm(x, pack_conv3x3(w), bias=b, res1=res, mode=ops.CONV3, conv=(n_img, hs, ws, 1, 0), tile=tile, split_k=split)
    xi = x.float().reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xi, w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + res.float()
    assert rel_l2(out, ref) < TOL_BF16
    # temporal mix + plain two-source, f32 out
    B, Fr, hw, Cc = 2, 4, 16, 320
    y = rnd(B * Fr * hw, Cc, seed=5)
    wt = rnd(Cc, 3 * Cc, seed=6, scale=(3 * Cc) ** -0.5)
    out = ops.gemm(y, wt, res1=y, mode=ops.TMIX, tmix=(hw, Fr), tile=tile, split_k=split)
    assert rel_l2(out, _tmix_ref(y, wt, torch.zeros(Cc, device=dev()), B, Fr, hw)) < TOL_BF16
    a1, a2 = rnd(200, 640, seed=7), rnd(200, 320, seed=8)
    w2 = rnd(132, 960, seed=9, scale=960 ** -0.5)
    out = ops.gemm(a1, w2, a2=a2, out_f32=True, tile=tile, split_k=split)
    assert rel_l2(out, torch.cat([a1, a2], 1).float() @ w2.float().T) < TOL_F32


@pytest.mark.parametrize("tile", [0, 3, 4, 6, 13, 17])
def test_gemm_conv3x3_stride2_asymmetric_pad(ops, tile):
    """VAE encoder Downsample2D: F.pad(x, (0, 1, 0, 1)) then 3x3 stride-2 conv with padding 0."""
    from asva_amd.weights import pack_conv3x3

    n_img, hs, ws, cin, cout = 2, 16, 24, 128, 128
    x = rnd(n_img * hs * ws, cin, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    b = rndf(cout, seed=3)
    out = ops.gemm(x, pack_conv3x3(w), bias=b, mode=ops.CONV3, conv=(n_img, hs, ws, 2, 0, 0), tile=tile)
    xi = F.pad(x.float().reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2), (0, 1, 0, 1))
    ref = F.conv2d(xi, w.float(), b, stride=2).permute(0, 2, 3, 1).reshape(-1, cout)
    assert out.shape == ref.shape and rel_l2(out, ref) < TOL_BF16


def test_gemm_batched_f32(ops):
    B, M, N, K = 3, 256, 192, 512
    a, w = rnd(B, M, K, seed=1), rnd(B, N, K, seed=2)
    out = ops.gemm_batched(a, w, alpha=K ** -0.5, out_f32=True)
    ref = torch.einsum("bmk,bnk->bmn", a.float(), w.float()) * K ** -0.5
    assert rel_l2(out, ref) < TOL_F32


def test_gemm_rejects_bad_arguments(ops):
    from asva_amd._lib import AvsdError

    a, w = rnd(64, 36), rnd(64, 36)   # K not a multiple of 8
    with pytest.raises(AvsdError):
        ops.gemm(a, w)


def test_linear_small_m(ops):
    for M, N, K, ai, ao in [(2, 1280, 320, False, True), (2, 9000, 1280, True, False), (12, 640, 640, False, False), (24, 320, 320, True, True)]:
        x = rndf(M, K, seed=1)
        w = rnd(N, K, seed=2, scale=K ** -0.5)
        b = rndf(N, seed=3)
        out = ops.linear_small_m(x, w, b, act_in=ai, act_out=ao)
        xin = F.silu(x) if ai else x
        ref = xin @ w.float().T + b
        ref = F.silu(ref) if ao else ref
        assert rel_l2(out, ref) < TOL_F32 * 5


# ---- normalisation -----------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [True, False])     # one launch (csrc/groupnorm_fused.hip) where the batch fits, else / or stats + apply
@pytest.mark.parametrize("nb,rows,c1,c2,groups,act", [
    (2, 12 * 64, 320, 0, 32, True),      # 5-D resnet norm
    (24, 64, 640, 0, 32, False),         # per-frame transformer norm
    (2, 12 * 16, 1280, 640, 32, True),   # concat with a group straddling the two sources (60 ch/group)
    (2, 4 * 64, 80, 80, 16, True),       # tiny config, 10 ch/group
    (1, 7 * 9, 2560, 0, 32, True),
    (2, 12 * 16, 640, 320, 32, True),    # 30 channels per group: four groups = 15 vectors per row
    (2, 12 * 16, 1280, 1280, 32, True),
    (2, 12 * 64, 1280, 1280, 32, True),
    (24, 256, 320, 0, 32, False),        # per-frame norm: four 10-channel groups per workgroup
    (24, 16, 1280, 0, 32, False),
    (12, 256, 512, 0, 32, True),         # 16 channels per group
    (2, 12 * 1024, 320, 0, 32, True),    # 32 x 32 level: stats + apply either way
    (2, 12 * 1024, 640, 320, 32, True),  # ... its widest skip concat (30 channels per group, 12 vectors per thread)
    (2, 12 * 256, 1280, 640, 32, True),
    (24, 1024, 320, 0, 32, False),       # per-frame norm at 32 x 32: the 960-thread one-launch form
    (3, 50, 32, 0, 8, True),             # 4 channels per group: a vector holds two whole groups
])
def test_groupnorm(ops, nb, rows, c1, c2, groups, act, fused, monkeypatch):
    monkeypatch.setattr(ops, "_GN_FUSED", fused)
    # sources and output are column slices of wider buffers (ld > channels), as the UNet's skip / fused-output views are
    x1 = (rnd(nb * rows, c1 + 16, seed=1) + 0.5)[:, 8:8 + c1]
    x2 = (rnd(nb * rows, c2 + 8, seed=2) * 2)[:, :c2] if c2 else None
    C = c1 + c2
    gamma, beta = rndf(C, seed=3) + 1.0, rndf(C, seed=4)
    eps = 1e-5
    buf = torch.zeros(nb * rows, C + 8, dtype=x1.dtype, device=x1.device)
    out = ops.groupnorm(x1, x2, nb, rows, groups, gamma, beta, eps, act, out=buf[:, :C])
    assert not buf[:, C:].any()
    x = torch.cat([x1, x2], 1) if c2 else x1
    xr = x.float().reshape(nb, rows, C).permute(0, 2, 1)
    ref = F.group_norm(xr, groups, gamma, beta, eps)
    ref = (F.silu(ref) if act else ref).permute(0, 2, 1).reshape(nb * rows, C)
    assert rel_l2(out, ref) < TOL_BF16
    # each group separately (a wrong group boundary inside a straddling vector would hide in the whole-tensor norm)
    cg = C // groups
    e = ((out.float() - ref) ** 2).reshape(nb, rows, groups, cg).sum((1, 3)).sqrt() / (ref ** 2).reshape(nb, rows, groups, cg).sum((1, 3)).sqrt()
    assert float(e.max()) < 2 * TOL_BF16


def test_groupnorm_one_launch_geometry():
    from asva_amd import _lib

    L = _lib.lib()
    # the small batches are one launch: ResBlock norms at 4 x 4, per-frame norms up to 16 x 16
    for nb, rows, c1, c2 in [(2, 192, 1280, 0), (2, 192, 1280, 1280), (24, 256, 640, 0), (24, 64, 1280, 0), (24, 16, 1280, 0)]:
        assert L.avsd_groupnorm_fused_supported(nb, rows, 32, c1, c2, 0) == 1 and L.avsd_groupnorm_fused_supported(nb, rows, 32, c1, c2, 1) == 1
    for nb, rows, c1, c2 in [(2, 12288, 320, 0), (2, 3072, 640, 0), (2, 768, 1280, 0), (12, 16384, 256, 0)]:
        assert L.avsd_groupnorm_fused_supported(nb, rows, 32, c1, c2, 0) == 0
    # many batches keep the chip busy with bigger slabs (960 threads x <= 8 vectors): the per-frame norm at 32 x 32; 16-bit storage only
    assert L.avsd_groupnorm_fused_supported(24, 1024, 32, 320, 0, 0) == 1 and L.avsd_groupnorm_fused_supported(24, 1024, 32, 320, 0, 1) == 0
    assert L.avsd_groupnorm_fused(None, 320, 320, None, 0, 0, 2, 12288, 32, None, None, 1e-5, 1, None, 320, None) != 0     # refused, not launched


@pytest.mark.parametrize("M,C", [(1000, 320), (257, 640), (64, 1280), (33, 80), (16, 2048)])
def test_layernorm(ops, M, C):
    x = rnd(M, C, seed=1) * 3 + 1
    g, b = rndf(C, seed=2) + 1, rndf(C, seed=3)
    out = ops.layernorm(x, g, b, 1e-5)
    assert rel_l2(out, F.layer_norm(x.float(), (C,), g, b, 1e-5)) < TOL_BF16


def test_layernorm_with_frame_pos(ops):
    B, Fr, hw, C = 2, 12, 16, 320
    x = rnd(B * Fr * hw, C, seed=1)
    pos = rndf(Fr, C, seed=2)
    g, b = rndf(C, seed=3) + 1, rndf(C, seed=4)
    out = ops.layernorm(x, g, b, 1e-5, pos=pos, hw=hw, frames=Fr)
    xp = x.float().reshape(B, Fr, hw, C) + pos[None, :, None, :]
    assert rel_l2(out, F.layer_norm(xp, (C,), g, b, 1e-5).reshape(-1, C)) < TOL_BF16


def test_softmax_rows(ops):
    s = rndf(300, 1024, seed=1) * 4
    out = ops.softmax_rows(s)
    assert rel_l2(out, torch.softmax(s, -1)) < TOL_BF16


# ---- attention -----------------------------------------------------------------------------------------
def _sdpa_ref(q, k, v, heads):
    # q [Bq, Lq, C], k/v [Bq, Lk, C]
    Bq, Lq, C = q.shape
    d = C // heads
    qh = q.float().reshape(Bq, Lq, heads, d).transpose(1, 2)
    kh = k.float().reshape(Bq, -1, heads, d).transpose(1, 2)
    vh = v.float().reshape(Bq, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(Bq * Lq, C)


@pytest.mark.parametrize("d", [40, 64, 80, 128, 160])
@pytest.mark.parametrize("Lq,Lk", [(256, 256), (64, 64), (16, 16), (200, 77)])
def test_attention_first_frame_layout(ops, d, Lq, Lk):
    heads, B, Fr = 4, 2, 3
    C = heads * d
    q = rnd(B * Fr * Lq, C, seed=1)
    kv = rnd(B * Lk, 2 * C, seed=2)          # fused k|v rows, one set per clip branch
    k, v = kv[:, :C], kv[:, C:]
    out = ops.attention(q, k, v, bq=B * Fr, lq=Lq, lk=Lk, kv_rows=Lk, heads=heads, q_per_kv=Fr, frames=Fr)
    kk = k.reshape(B, 1, Lk, C).expand(B, Fr, Lk, C).reshape(B * Fr, Lk, C)
    vv = v.reshape(B, 1, Lk, C).expand(B, Fr, Lk, C).reshape(B * Fr, Lk, C)
    ref = _sdpa_ref(q.reshape(B * Fr, Lq, C), kk, vv, heads)
    assert rel_l2(out, ref) < TOL_BF16


def test_attention_key_gather_matches_bool_mask(ops):
    # audio cross-attention: 229 keys, per-frame boolean mask == gather of the unmasked keys
    from asva_amd.conditioning import audio_segment_mask, mask_to_key_index

    heads, d, B, Fr, Lq = 8, 40, 2, 12, 64
    C = heads * d
    q = rnd(B * Fr * Lq, C, seed=1)
    k, v = rnd(B * 229, C, seed=2), rnd(B * 229, C, seed=3)
    mask = audio_segment_mask(Fr)                                  # [Fr, 229] bool
    idx = mask_to_key_index(mask).to(dev())                        # [Fr, 25] int32
    out = ops.attention(q, k, v, bq=B * Fr, lq=Lq, lk=idx.shape[1], kv_rows=229, heads=heads, q_per_kv=Fr,
                        frames=Fr, key_index=idx)
    qh = q.float().reshape(B, Fr, Lq, heads, d).permute(0, 1, 3, 2, 4)
    kh = k.float().reshape(B, 1, 229, heads, d).permute(0, 1, 3, 2, 4).expand(B, Fr, heads, 229, d)
    vh = v.float().reshape(B, 1, 229, heads, d).permute(0, 1, 3, 2, 4).expand(B, Fr, heads, 229, d)
    m = mask.to(dev())[None, :, None, None, :]
    ref = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=m).permute(0, 1, 3, 2, 4).reshape(B * Fr * Lq, C)
    assert rel_l2(out, ref) < TOL_BF16


@pytest.mark.parametrize("d", [40, 64, 80, 160])
def test_attention_online_softmax_rescale(ops, d):
    # keys far above the rest in LATER tiles force the (rare, deferred) running-max rescale: one spike on each
    # half of the wave (keys 4..7 mod 8 live on lanes 32..63), each aimed at a different query
    heads, Lq, Lk = 1, 32, 160
    q = rnd(Lq, d, seed=1)
    k = rnd(Lk, d, seed=2)
    v = rnd(Lk, d, seed=3)
    k[70] = q[5] * 4
    k[97] = q[11] * 6
    k[130] = q[5] * 8
    out = ops.attention(q, k, v, bq=1, lq=Lq, lk=Lk, kv_rows=Lk, heads=heads, q_per_kv=1, frames=1)
    ref = _sdpa_ref(q[None], k[None], v[None], heads)
    assert rel_l2(out, ref) < TOL_BF16
    assert (out.float() - ref).abs().max() < 0.05      # no row is left at a stale scale


@pytest.mark.parametrize("d", [40, 64])
def test_attention_slowly_growing_max(ops, d):
    # the row maximum creeps up by less than the rescale threshold per tile: the stale-max path must stay exact
    heads, Lq, Lk = 1, 64, 256
    q = rnd(Lq, d, seed=1)
    k = rnd(Lk, d, seed=2) * torch.linspace(0.5, 3.0, Lk, device=dev())[:, None].to(torch.bfloat16)
    v = rnd(Lk, d, seed=3)
    out = ops.attention(q, k, v, bq=1, lq=Lq, lk=Lk, kv_rows=Lk, heads=heads, q_per_kv=1, frames=1)
    ref = _sdpa_ref(q[None], k[None], v[None], heads)
    assert rel_l2(out, ref) < TOL_BF16


@pytest.mark.parametrize("d", [40, 80, 160, 64])
@pytest.mark.parametrize("Fr", [12, 24, 4])
def test_temporal_attention(ops, d, Fr):
    heads, B, hw = 8, 2, 24
    if heads * Fr > 256:
        heads = 256 // Fr
    C = heads * d
    qkv = rnd(B * Fr * hw, 3 * C, seed=1)
    out = ops.temporal_attention(qkv, b=B, frames=Fr, hw=hw, heads=heads)
    x = qkv.float().reshape(B, Fr, hw, 3, heads, d).permute(3, 0, 2, 4, 1, 5)   # [3, B, hw, heads, Fr, d]
    o = F.scaled_dot_product_attention(x[0], x[1], x[2])                         # [B, hw, heads, Fr, d]
    ref = o.permute(0, 3, 1, 2, 4).reshape(B * Fr * hw, C)
    assert rel_l2(out, ref) < TOL_BF16


# ---- elementwise ------------------------------------------------------------------------------------------
def test_layout_roundtrip_and_replication(ops):
    B, C, Fr, H, W = 2, 4, 3, 8, 8
    x = rndf(B, C, Fr, H, W, seed=1)
    rows = ops.ncfhw_to_rows(x, cpad=8, rep=2, scale=0.5)
    ref = (0.5 * x).permute(0, 2, 3, 4, 1).reshape(-1, C).to(torch.bfloat16)
    assert rows.shape == (2 * B * Fr * H * W, 8)
    assert torch.equal(rows[: B * Fr * H * W, :C], ref) and torch.equal(rows[B * Fr * H * W:, :C], ref)
    assert torch.count_nonzero(rows[:, C:]) == 0
    r32 = rndf(B * Fr * H * W, 8, seed=2)
    back = ops.rows_to_ncfhw(r32, B, C, Fr, H, W)
    assert torch.equal(back, r32[:, :C].reshape(B, Fr, H, W, C).permute(0, 4, 1, 2, 3))


def test_timestep_embedding(ops):
    t = torch.tensor([0.0, 1.0, 981.0, 501.0], device=dev())
    out = ops.timestep_embedding(t, 320)
    half = 160
    w = torch.exp(-math.log(10000.0) * torch.arange(half, device=dev(), dtype=torch.float32) / half)
    arg = t[:, None] * w[None]
    ref = torch.cat([torch.cos(arg), torch.sin(arg)], -1)
    assert torch.allclose(out, ref, atol=2e-4)
    assert torch.equal(out[0], torch.cat([torch.ones(half), torch.zeros(half)]).to(dev()))


def test_guided_step(ops):
    B, C, Fr, H, W = 1, 4, 5, 8, 8
    npred = rndf(2 * B, C, Fr, H, W, seed=1)
    x = rndf(B, C, Fr, H, W, seed=2)
    hist = rndf(4, B, C, Fr, H, W, seed=3)
    xo = torch.empty_like(x)
    hist2 = hist.clone()
    ops.guided_step(npred, 2, 4.0, x, xo, 0.9, -0.3, eps_hist=hist2, store_slot=2, w_cur=0.0, hist_idx=(2, 1, 0),
                    w=(23 / 12, -16 / 12, 5 / 12))
    eps = npred[:B] + 4.0 * (npred[B:] - npred[:B])
    e = 23 / 12 * eps - 16 / 12 * hist[1] + 5 / 12 * hist[0]
    ref = 0.9 * x - 0.3 * e
    ref[:, :, 0] = x[:, :, 0]
    assert torch.allclose(xo, ref, atol=1e-5)
    assert torch.allclose(hist2[2], eps, atol=1e-6) and torch.equal(hist2[3], hist[3])
    # dual guidance (pipeline_audio_cond_animation.py:349-353): [uncond, text, text+audio]
    np3 = rndf(3 * B, C, Fr, H, W, seed=4)
    xo3 = torch.empty_like(x)
    ops.guided_step(np3, 3, 7.5, x, xo3, 0.9, -0.3, g2=4.0)
    eps3 = np3[:B] + 7.5 * (np3[B:2 * B] - np3[:B]) + 4.0 * (np3[2 * B:] - np3[B:2 * B])
    ref3 = 0.9 * x - 0.3 * eps3
    ref3[:, :, 0] = x[:, :, 0]
    assert torch.allclose(xo3, ref3, atol=2e-5)
    # DDIM form, no guidance, in place
    x2 = x.clone()
    ops.guided_step(npred[:B].contiguous(), 1, 1.0, x2, x2, 1.1, 0.2)
    ref2 = 1.1 * x + 0.2 * npred[:B]
    ref2[:, :, 0] = x[:, :, 0]
    assert torch.allclose(x2, ref2, atol=1e-5)


def test_vae_postprocess(ops):
    n, H, W = 3, 16, 8
    rows = rnd(n * H * W, 4, seed=1)
    u8 = ops.vae_postprocess_u8(rows, n, H, W)
    want = ((rows[:, :3].float().reshape(n, H, W, 3) / 2 + 0.5).clamp(0, 1) * 255).to(torch.uint8)
    assert u8.dtype == torch.uint8 and torch.equal(u8, want)
    out = ops.vae_postprocess(rows, n, H, W)
    ref = (rows[:, :3].float().reshape(n, H, W, 3).permute(0, 3, 1, 2) / 2 + 0.5).clamp(0, 1)
    assert torch.allclose(out, ref, atol=1e-6)


# ---- fused cross-attention block ------------------------------------------------------------------------------------
@pytest.mark.parametrize("f32_res", [False, True])
@pytest.mark.parametrize("kind,lk,per_frame", [("audio", 25, True), ("audio4", 61, True), ("text", 77, False)])
def test_cross_attention_block_matches_separate_kernels_and_fp32(ops, kind, lk, per_frame, f32_res):
    """avsd_cross_attention_block (one launch) against (a) the three kernels it replaces — LayerNorm-folded Q projection,
    attention over the gathered keys, output projection + residual — and (b) a plain fp32 statement of
    h + to_out(softmax(LN(h) Wq K^T / sqrt(d)) V).  C = 320, 8 heads of 40, 2 clips x 3 frames x 256 rows."""
    torch.manual_seed(0)
    B, Fr, L, C, heads = 2, 3, 256, 320, 8
    d = C // heads
    M = B * Fr * L
    a0 = rnd(M, C, seed=1)
    w0 = rnd(C, C, seed=2, scale=C ** -0.5)
    res0 = rnd(M, C, seed=3) + 0.5
    stats = torch.empty(M, C // 32, 2, device=dev())
    h = ops.gemm(a0, w0, res1=res0, rowstats=stats)                     # residual stream + its LayerNorm statistics
    gamma, beta = 1 + 0.1 * rndf(C, seed=4), 0.1 * rndf(C, seed=5)
    wq = rndf(C, C, seed=6, scale=C ** -0.5)
    wq_f = (wq * gamma[None, :]).to(torch.bfloat16)
    q_colsum, q_bias = wq_f.float().sum(1), wq @ beta
    wo, bo = rnd(C, C, seed=7, scale=C ** -0.5), rndf(C, seed=8)
    nkv = B * Fr if per_frame else B
    lkp = (lk + 31) // 32 * 32
    kk, vv = rnd(nkv, lk, C, seed=9), rnd(nkv, lk, C, seed=10)
    k_pad = torch.zeros(nkv, lkp, C, dtype=torch.bfloat16, device=dev())
    vt_pad = torch.zeros(nkv, C, lkp, dtype=torch.bfloat16, device=dev())
    k_pad[:, :lk] = kk
    vt_pad[:, :, :lk] = vv.transpose(1, 2)
    q_per_kv = 1 if per_frame else Fr
    master_in = h.float() + 1e-3 * rndf(M, C, seed=11) if f32_res else None      # an f32 master that differs from its 16-bit copy
    res = master_in if f32_res else h
    stats_out = torch.empty_like(stats)
    master = torch.empty(M, C, device=dev()) if f32_res else None
    out = ops.cross_attention_block(h, stats, wq_f, q_colsum, q_bias, k_pad, vt_pad, lk, wo, bo, res=res, heads=heads, L=L,
                                    q_per_kv=q_per_kv, rowstats=stats_out, master=master)
    # (a) the separate kernels
    q = ops.gemm(h, wq_f, bias=q_bias, ln=(stats, q_colsum, 1e-5))
    o = ops.attention(q, kk.reshape(nkv * lk, C), vv.reshape(nkv * lk, C), bq=B * Fr, lq=L, lk=lk, kv_rows=lk, heads=heads,
                      q_per_kv=q_per_kv, frames=Fr)
    stats_sep = torch.empty_like(stats)
    sep = ops.gemm(o, wo, bias=bo, res1=res, rowstats=stats_sep)
    # (b) fp32
    hn = F.layer_norm(h.float(), (C,), gamma, beta, 1e-5)
    qf = (hn @ wq.T).reshape(B * Fr, L, heads, d).transpose(1, 2)
    kv_of = torch.arange(B * Fr, device=dev()) // q_per_kv
    kf = kk.float()[kv_of].reshape(B * Fr, lk, heads, d).transpose(1, 2)
    vf = vv.float()[kv_of].reshape(B * Fr, lk, heads, d).transpose(1, 2)
    of = F.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(M, C)
    ref = of @ wo.float().T + bo + res.float()
    e_sep, e_ref = rel_l2(out, sep), rel_l2(out, ref)
    print(f"cross_attention_block {kind} lk={lk} f32_res={f32_res}: vs separate kernels {e_sep:.3e}, vs fp32 {e_ref:.3e}")
    assert e_sep < 3e-3 and e_ref < TOL_BF16
    # attention-only part: subtracting the residual leaves to_out(attention) — compare that too (the residual dominates `out`)
    assert rel_l2(out.float() - res.float(), ref - res.float()) < 2.5e-2
    ob = out.float().reshape(M, C // 32, 32)
    assert torch.allclose(stats_out[..., 0], ob.sum(-1), atol=1e-3, rtol=1e-5) and torch.allclose(stats_out[..., 1], (ob * ob).sum(-1), atol=1e-2, rtol=1e-5)
    if f32_res:
        assert rel_l2(master, ref) < 2e-3 and rel_l2(master.to(torch.bfloat16), out) < 1e-6
    # statistics of out + pos[frame] (the LayerNorm of the temporal attention, folded): same output bits, shifted statistics
    pos = rndf(Fr, C, seed=12)
    stats_pos = torch.empty_like(stats)
    out2 = ops.cross_attention_block(h, stats, wq_f, q_colsum, q_bias, k_pad, vt_pad, lk, wo, bo, res=res, heads=heads, L=L,
                                     q_per_kv=q_per_kv, rowstats=stats_pos, stats_pos=(pos, L, Fr))
    assert torch.equal(out2, out)
    up = (out.float() + pos[(torch.arange(M, device=dev()) // L) % Fr]).reshape(M, C // 32, 32)
    assert torch.allclose(stats_pos[..., 0], up.sum(-1), atol=1e-3, rtol=1e-5) and torch.allclose(stats_pos[..., 1], (up * up).sum(-1), atol=1e-2, rtol=1e-5)


# ---- FP8 (e4m3) Q/K/V attention: BASELINE cfg 5 -----------------------------------------------------------------------
TOL_FP8 = 8e-2      # e4m3 keeps 3 mantissa bits (2^-4 relative rounding) on q, k, v and the probabilities; measured 4-6e-2


@pytest.mark.parametrize("d,heads", [(40, 8), (64, 12), (80, 8), (128, 4), (160, 8)])
@pytest.mark.parametrize("case", ["first_frame", "text", "audio_gather"])
def test_attention_fp8_matches_bf16_kernel_and_fp32(ops, d, heads, case):
    """avsd_attention_fp8 against the 16-bit kernel and an fp32 SDPA on the same inputs: first-frame layout (K/V of frame 0
    shared by the frames of a clip), text cross-attention (77 keys) and the audio key gather (25 of 229 keys per frame)."""
    B, Fr, C = 2, 3, d * heads
    if case == "first_frame":
        L, lk, kv_rows, q_per_kv, idx = 200, 200, 200, Fr, None
    elif case == "text":
        L, lk, kv_rows, q_per_kv, idx = 160, 77, 77, Fr, None
    else:
        L, lk, kv_rows, q_per_kv = 160, 25, 229, Fr
        g = torch.Generator().manual_seed(5)
        idx = torch.stack([torch.randperm(229, generator=g)[:lk].sort().values for _ in range(Fr)]).to(torch.int32).to(dev())
    q = rnd(B * Fr * L, C, seed=1)
    k, v = rnd(B * kv_rows, C, seed=2), rnd(B * kv_rows, C, seed=3)
    kw = dict(bq=B * Fr, lq=L, lk=lk, kv_rows=kv_rows, heads=heads, q_per_kv=q_per_kv, frames=Fr, key_index=idx)
    o16 = ops.attention(q, k, v, **kw)
    o8 = ops.attention(q, k, v, fp8=(1.0, 1.0, 1.0), **kw)
    qf = q.float().reshape(B, Fr, L, heads, d).permute(0, 1, 3, 2, 4)
    kf = k.float().reshape(B, kv_rows, heads, d).permute(0, 2, 1, 3)
    vf = v.float().reshape(B, kv_rows, heads, d).permute(0, 2, 1, 3)
    outs = []
    for f in range(Fr):
        sel = idx[f].long() if idx is not None else torch.arange(lk, device=dev())
        outs.append(F.scaled_dot_product_attention(qf[:, f], kf[:, :, sel], vf[:, :, sel]))
    ref = torch.stack(outs, 1).permute(0, 1, 3, 2, 4).reshape(B * Fr * L, C)
    e16, e8, e816 = rel_l2(o16, ref), rel_l2(o8, ref), rel_l2(o8, o16)
    print(f"attention fp8 d={d} {case}: bf16 kernel vs fp32 {e16:.3e} | fp8 vs fp32 {e8:.3e} | fp8 vs bf16 kernel {e816:.3e}")
    assert e16 < TOL_BF16 and e8 < TOL_FP8 and e816 < TOL_FP8
    # per-tensor scales that are powers of two only shift exponents: same roundings, same result
    o8s = ops.attention(q, k, v, fp8=(4.0, 0.5, 2.0), **kw)
    assert rel_l2(o8s, o8) < 1e-2      # (values that fall into the subnormal range of e4m3 under one scaling and not the other round differently)


@pytest.mark.parametrize("L", [1024, 200, 77])
def test_attention_single_wide_head_512(ops, L):
    """The VAE mid-block attention: one head over all 512 channels (attn_wide_kernel: the head dimension split over the four
    waves of a workgroup, partial scores summed through LDS).  q | k | v as strided views of one fused projection."""
    n, C = 3, 512
    qkv = rnd(n * L, 3 * C, seed=1, scale=0.5)
    o = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], bq=n, lq=L, lk=L, kv_rows=L, heads=1, q_per_kv=1, frames=1)
    f = qkv.float().reshape(n, L, 3, C)
    ref = F.scaled_dot_product_attention(f[:, None, :, 0], f[:, None, :, 1], f[:, None, :, 2]).reshape(n * L, C)
    err = rel_l2(o, ref)
    print(f"attention d=512 one head, L={L}: rel-L2 vs fp32 {err:.3e}")
    assert o.shape == (n * L, C) and err < TOL_BF16


def test_copy_and_replicate(ops):
    x = rnd(96, 40, seed=1)
    assert torch.equal(ops.copy(x, rep=3), torch.cat([x] * 3))
    f = rndf(24, 3, 8, seed=2)
    dst = torch.empty(48, 3, 8, device=dev())
    ops.copy(f, dst, rep=2)
    assert torch.equal(dst, torch.cat([f, f]))
    with pytest.raises(ValueError):
        ops.copy(x[:, :8])                       # not contiguous
    with pytest.raises(ValueError):
        ops.copy(rnd(3, 3, seed=3))              # 18 bytes: not a multiple of 16


@pytest.mark.parametrize("gather", [False, True])
def test_xattn_pack_kv_matches_indexing(ops, gather):
    """the K / V^T operand layout of the fused cross-attention block from cached K|V rows: plain (text, 77 keys) and through the
    per-frame gather list of the audio segment mask"""
    from asva_amd.conditioning import audio_segment_mask, mask_to_key_index

    n_kv, C, Fr = 2, 320, 12
    rows = 229 if gather else 77
    kv = rnd(n_kv * rows, 2 * C, seed=5)
    idx = mask_to_key_index(audio_segment_mask(Fr)).to(dev()) if gather else None
    lk = idx.shape[1] if gather else rows
    nb = n_kv * Fr if gather else n_kv
    lkp = (lk + 31) // 32 * 32
    k = torch.full((nb, lkp, C), float("nan"), dtype=kv.dtype, device=dev())      # poisoned: the launch owns the padding too
    vt = torch.full((nb, C, lkp), float("nan"), dtype=kv.dtype, device=dev())
    ops.xattn_pack_kv(kv, n_kv, rows, C, idx, k, vt)
    kv3 = kv.view(n_kv, rows, 2 * C)
    if gather:
        kv3 = kv3[:, idx.long()].reshape(nb, lk, 2 * C)
    assert torch.equal(k[:, :lk], kv3[..., :C]) and torch.equal(vt[:, :, :lk], kv3[..., C:].transpose(1, 2))
    assert not k[:, lk:].any() and not vt[:, :, lk:].any()          # padding zero-filled by the launch


# ---- hand-scheduled 4-wave tiles (csrc/gemm4.hip, ids 60-67) ---------------------------------------------------------------
@pytest.fixture
def krot_off(ops):
    """the unrotated K walk: same f32 order as the LDS-direct tiles"""
    ops.set_krot(False)
    yield
    ops.set_krot(True)


ASM_TILES = [60, 61, 62, 63, 64, 65, 66, 67]


@pytest.mark.parametrize("tile", ASM_TILES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 384, 320), (1000, 640, 1280), (77, 132, 192), (3000, 320, 640)])
def test_gemm_asm_tiles_plain(ops, tile, M, N, K, krot_off):
    """csrc/gemm4.hip: same products in the same K order as the LDS-direct tiles -> bit-identical to tile 9 (odd and even numbers of K
    tiles, M / N tails, K = one tile), against torch in f32, run to run; split-K slabs; every epilogue term"""
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res, res2 = rndf(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
    out = ops.gemm(a, w, bias=bias, res1=res, res2=res2, alpha=0.5, tile=tile)
    assert rel_l2(out, 0.5 * (a.float() @ w.float().T) + bias + res.float() + res2.float()) < TOL_BF16
    assert torch.equal(out, ops.gemm(a, w, bias=bias, res1=res, res2=res2, alpha=0.5, tile=9))
    o32 = ops.gemm(a, w, bias=bias, out_f32=True, tile=tile)
    assert rel_l2(o32, a.float() @ w.float().T + bias) < TOL_F32 and torch.equal(o32, ops.gemm(a, w, bias=bias, out_f32=True, tile=9))
    assert all(torch.equal(ops.gemm(a, w, bias=bias, out_f32=True, tile=tile), o32) for _ in range(3))
    if K >= 256:
        for sk in (2, 3):
            assert torch.equal(ops.gemm(a, w, bias=bias, out_f32=True, tile=tile, split_k=sk), ops.gemm(a, w, bias=bias, out_f32=True, tile=9, split_k=sk))
    if N % 32 == 0:         # LayerNorm producer / consumer flags and GEGLU through the shared epilogue
        st, st9 = torch.empty(M, N // 32, 2, device=dev()), torch.empty(M, N // 32, 2, device=dev())
        h = ops.gemm(a, w, bias=bias, res1=res, rowstats=st, tile=tile)
        assert torch.equal(h, ops.gemm(a, w, bias=bias, res1=res, rowstats=st9, tile=9)) and torch.equal(st, st9)
        if N % 64 == 0:
            w2 = rnd(64, N, seed=7, scale=N ** -0.5)
            cs = w2.float().sum(1)
            y = ops.gemm(h, w2, ln=(st, cs, 1e-5), tile=tile)
            assert torch.equal(y, ops.gemm(h, w2, ln=(st, cs, 1e-5), tile=9))
            assert torch.equal(ops.gemm(h, w2, geglu=True, ln=(ops.ln_fold(st), cs, 1e-5), tile=tile),
                               ops.gemm(h, w2, geglu=True, ln=(ops.ln_fold(st), cs, 1e-5), tile=9))


@pytest.mark.parametrize("M,N,K", [(960, 2560, 320), (1000, 2560, 320), (96, 1280, 320), (2000, 5120, 640), (6144, 2560, 640), (200, 256 * 5, 320),
                                   (500, 320, 320), (3000, 960, 320), (700, 1920, 640), (130, 64, 640)])
def test_gemm_nstream_tile(ops, M, N, K, krot_off):
    """csrc/nstream.hip (tile 70): the A band resident in LDS, W streamed in fragment order by 8 independent waves.  Same products in the
    same K order through the shared epilogue -> bit-identical to the LDS-direct tile 9: plain + residual, f32 output, GEGLU, GEGLU with
    the LayerNorm fold from raw (K / 32 pairs) and pre-folded statistics; M tails; refused without the fragment-ordered weights."""
    from asva_amd.weights import pack_frag, pack_geglu
    by_rule = ops.nstream_supported(M, N, K)          # (the other shapes: explicit tile 70 — fragment counts that are no multiple of 8, few fragments)
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res = rndf(N, seed=3), rnd(M, N, seed=4)
    wf = pack_frag(w)
    out = ops.gemm(a, w, bias=bias, res1=res, tile=70, w_frag=wf)
    assert rel_l2(out, a.float() @ w.float().T + bias + res.float()) < TOL_BF16
    assert torch.equal(out, ops.gemm(a, w, bias=bias, res1=res, tile=9))
    o32 = ops.gemm(a, w, bias=bias, out_f32=True, tile=70, w_frag=wf)
    assert rel_l2(o32, a.float() @ w.float().T + bias) < TOL_F32 and torch.equal(o32, ops.gemm(a, w, bias=bias, out_f32=True, tile=9))
    assert all(torch.equal(ops.gemm(a, w, bias=bias, out_f32=True, tile=70, w_frag=wf), o32) for _ in range(3))
    # the GEGLU projection as the transformer block runs it: producer statistics -> LayerNorm fold -> value * gelu(gate)
    wp, bp = pack_geglu(w.float(), bias)
    wpf = pack_frag(wp)
    st = torch.empty(M, K // 32, 2, device=dev())
    h = ops.gemm(rnd(M, K, seed=6), rnd(K, K, seed=7, scale=K ** -0.5), rowstats=st, tile=9)
    cs = wp.float().sum(1)
    want = ops.gemm(h, wp, bias=bp, geglu=True, ln=(st, cs, 1e-5), tile=9)
    assert torch.equal(ops.gemm(h, wp, bias=bp, geglu=True, ln=(st, cs, 1e-5), tile=70, w_frag=wpf), want)
    assert torch.equal(ops.gemm(h, wp, bias=bp, geglu=True, ln=(ops.ln_fold(st), cs, 1e-5), tile=70, w_frag=wpf), want)
    if by_rule:
        assert torch.equal(ops.gemm(h, wp, bias=bp, geglu=True, ln=(st, cs, 1e-5), w_frag=wpf), want)          # tile 0: chosen by rule
    hf = h.float()
    ln = (hf - hf.mean(1, keepdim=True)) * torch.rsqrt(hf.var(1, unbiased=False, keepdim=True) + 1e-5)
    y = ln @ w.float().T + bias
    assert rel_l2(want, y[:, :N // 2] * torch.nn.functional.gelu(y[:, N // 2:])) < TOL_BF16
    with pytest.raises(ValueError, match="w_frag"):
        ops.gemm(a, w, tile=70)


@pytest.mark.parametrize("tile", [61, 62, 63, 64, 65, 66, 67])
@pytest.mark.parametrize("B,hw,C,N", [(2, 64, 320, 320), (1, 32, 128, 132), (3, 96, 64, 64), (1, 16, 1280, 1280)])
def test_gemm_asm_tiles_tmix(ops, tile, B, hw, C, N, krot_off):
    """the temporal-mix A operand in the hand-scheduled loop: per-vector jumps at the two K-segment boundaries (frame 0 -> previous
    frame -> current frame), also for K slices that start inside a segment; full epilogue; against the f32 statement and tile 9"""
    Fr = 12
    M = B * Fr * hw
    y = rnd(M, C, seed=1)
    w = rnd(N, 3 * C, seed=2, scale=(3 * C) ** -0.5)
    b, temb, res2 = rndf(N, seed=3), rndf(B, N, seed=5), rnd(M, N, seed=4)
    y5 = y.float().reshape(B, Fr, hw, C)
    prev = torch.cat([y5[:, :1], y5[:, :-1]], 1)
    cat = torch.cat([y5[:, :1].expand_as(y5), prev, y5], -1).reshape(M, 3 * C)
    ref = cat @ w.float().T + b + temb.repeat_interleave(Fr * hw, 0) + res2.float()
    kw = dict(bias=b, rowvec=temb, rows_per_vec=Fr * hw, res2=res2, mode=ops.TMIX, tmix=(hw, Fr))
    assert rel_l2(ops.gemm(y, w, tile=tile, **kw), ref) < TOL_BF16
    o32 = ops.gemm(y, w, out_f32=True, tile=tile, **kw)
    assert rel_l2(o32, ref) < TOL_F32 and torch.equal(o32, ops.gemm(y, w, out_f32=True, tile=9, **kw))
    for sk in (2, 3):
        if (3 * C // 64) // sk >= 2:
            assert torch.equal(ops.gemm(y, w, out_f32=True, tile=tile, split_k=sk, **kw), ops.gemm(y, w, out_f32=True, tile=9, split_k=sk, **kw))


@pytest.mark.parametrize("tile", [61, 63, 64, 65, 66, 67])
def test_gemm_asm_tiles_rotated_k_walk(ops, tile):
    """AVSD_GEMM_KROT (the default of the asm tiles): every row band starts its K walk at another tile and wraps — also inside split-K
    slices and across the temporal-mix segment boundaries.  Same products, another f32 order: f32-output tolerance against torch,
    bit-identical run to run, and different bits from the unrotated walk somewhere (the flag really rotates)."""
    ops.set_krot(True)
    for M, N, K in [(1000, 640, 1280), (3000, 320, 640), (1536, 1280, 1920)]:
        a, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rndf(N, seed=3)
        ref = a.float() @ w.float().T + bias
        for sk in (1, 2, 3):
            o = ops.gemm(a, w, bias=bias, out_f32=True, tile=tile, split_k=sk)
            assert rel_l2(o, ref) < TOL_F32
            assert all(torch.equal(ops.gemm(a, w, bias=bias, out_f32=True, tile=tile, split_k=sk), o) for _ in range(3))
        ops.set_krot(False)
        plain = ops.gemm(a, w, bias=bias, out_f32=True, tile=tile)
        ops.set_krot(True)
        if tile != 61 or M > 256:
            assert not torch.equal(ops.gemm(a, w, bias=bias, out_f32=True, tile=tile), plain)
    if tile not in (61, 67):          # (their temporal-mix loops walk K unrotated)
        for B, hw, C, N in [(2, 64, 320, 320), (1, 32, 128, 132), (1, 16, 1280, 1280), (2, 64, 192, 64)]:
            Fr = 12
            M = B * Fr * hw
            y, w, b = rnd(M, C, seed=1), rnd(N, 3 * C, seed=2, scale=(3 * C) ** -0.5), rndf(N, seed=3)
            y5 = y.float().reshape(B, Fr, hw, C)
            prev = torch.cat([y5[:, :1], y5[:, :-1]], 1)
            ref = torch.cat([y5[:, :1].expand_as(y5), prev, y5], -1).reshape(M, 3 * C) @ w.float().T + b
            for sk in (1, 2, 3, 4):
                nk = 3 * C // 64
                if nk // sk < 2:
                    continue
                if (sk - 1) * -(-nk // sk) >= nk:        # a slice without K tiles: refused, not zero-filled
                    with pytest.raises(RuntimeError, match="empty slice"):
                        ops.gemm(y, w, bias=b, out_f32=True, mode=ops.TMIX, tmix=(hw, Fr), tile=tile, split_k=sk)
                    continue
                o = ops.gemm(y, w, bias=b, out_f32=True, mode=ops.TMIX, tmix=(hw, Fr), tile=tile, split_k=sk)
                assert rel_l2(o, ref) < TOL_F32, (B, hw, C, N, sk)


def test_gemm_asm_tiles_refuse(ops):
    a, w = rnd(512, 1280, seed=1), rnd(256, 1280, seed=2)
    with pytest.raises(RuntimeError, match="asm tiles"):
        ops.gemm(a[:, :640], w, a2=a[:, 640:], tile=63)                    # two-source operand
    with pytest.raises(RuntimeError, match="asm tiles"):
        ops.gemm(rnd(128, 200, seed=1), rnd(64, 200, seed=2), tile=63)     # K % 64 != 0
    with pytest.raises(RuntimeError, match="asm tiles"):
        ops.gemm(rnd(2 * 12 * 32, 64, seed=1), rnd(64, 192, seed=2), mode=ops.TMIX, tmix=(32, 12), tile=60)    # no TMIX form of the 256 x 256 tile


# nearest-2x upsample + 3x3 convolution in its sub-pixel form (AVSD_GEMM_CONV3 with ups = 2, include/avsd.h; FFSpatioTempResUpsample3D,
# ff_spatio_temp_resnet_3d.py:48-55): four per-parity 2x2 convolutions on the original image, results scattered to the upsampled pixels
@pytest.mark.parametrize("tile,split", [(0, 1), (4, 1), (6, 1), (9, 1), (11, 1), (13, 1), (17, 1), (20, 1), (24, 1), (25, 1), (30, 1), (38, 1), (6, 2), (9, 4)])
@pytest.mark.parametrize("n_img,hs,ws,cin,cout", [(3, 4, 4, 64, 64), (24, 8, 8, 128, 320), (2, 16, 16, 320, 640), (5, 3, 5, 64, 128)])
def test_gemm_conv3_subpixel_upsample(ops, tile, split, n_img, hs, ws, cin, cout):
    from asva_amd import precision as P
    from asva_amd.weights import subpixel_conv3x3

    if tile in ops.TILE_BN and cout % ops.TILE_BN[tile]:
        pytest.skip("column tile wider than a divisor of cout")
    assert tile == 0 or tile in ops.SUBPIX_TILES
    if split > (4 * cin) // 64 // 2:
        pytest.skip("too few K tiles for this split")
    x = rnd(n_img * hs * ws, cin, seed=1)
    w = rndf(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    b = rndf(cout, seed=3)
    wp = subpixel_conv3x3(w.permute(0, 2, 3, 1).contiguous()).to(P.ACT).contiguous()
    ref = F.conv2d(F.interpolate(x.float().reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"),
                   w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    # the upsampled 3x3 convolution on the un-folded weights, through the tap-major tile: same function, other rounding of the weights
    w9 = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).to(P.ACT).contiguous()
    old = ops.gemm(x, w9, bias=b, mode=ops.CONV3, conv=(n_img, hs, ws, 1, 1), out_f32=True)
    o32 = ops.gemm(x, wp, bias=b.repeat(4), mode=ops.CONV3, conv=(n_img, hs, ws, 1, 2), out_f32=True, tile=tile, split_k=split)
    assert o32.shape == (n_img * 4 * hs * ws, cout)
    assert rel_l2(o32, ref) < TOL_BF16 and rel_l2(old, ref) < TOL_BF16
    # exactly: f32 of the packed operands, parity by parity (a wrong tap, parity or output pixel is an O(1) error)
    wq = wp.float()
    xp = F.pad(x.float().reshape(n_img, hs, ws, cin), (0, 0, 1, 1, 1, 1))
    want = torch.zeros(n_img, 2 * hs, 2 * ws, cout, device=dev())
    for dy in range(2):
        for dx in range(2):
            taps = torch.cat([xp[:, dy + i:dy + i + hs, dx + j:dx + j + ws] for i in range(2) for j in range(2)], -1).reshape(-1, 4 * cin)
            par = 2 * dy + dx
            want[:, dy::2, dx::2] = (taps @ wq[par * cout:(par + 1) * cout].T + b).reshape(n_img, hs, ws, cout)
    assert rel_l2(o32, want.reshape(-1, cout)) < TOL_F32
    # 16-bit output + f32 master + rest plane through the same scatter
    out, rest = ops.alloc_planes((n_img * 4 * hs * ws, cout), dev())
    master = torch.empty((n_img * 4 * hs * ws, cout), dtype=torch.float32, device=dev())
    ops.gemm(x, wp, bias=b.repeat(4), mode=ops.CONV3, conv=(n_img, hs, ws, 1, 2), out=out, out_rest=rest, master=master, tile=tile, split_k=split)
    assert torch.equal(master, o32) and torch.equal(out, master.to(P.ACT)) and torch.equal(rest, (master - out.float()).to(P.ACT))
    # three MFMA passes on (main, rest) planes: the summed weights to 2^-17
    if tile in (0, 11, 13, 24, 25):
        wm = subpixel_conv3x3(w.permute(0, 2, 3, 1).contiguous())
        pw, pwr = ops.alloc_planes(tuple(wm.shape), dev())
        pw.copy_(wm.to(P.ACT)), pwr.copy_((wm - wm.to(P.ACT).float()).to(P.ACT))
        xf = rndf(n_img * hs * ws, cin, seed=1)
        px, pxr = ops.alloc_planes(tuple(xf.shape), dev())
        px.copy_(xf.to(P.ACT)), pxr.copy_((xf - xf.to(P.ACT).float()).to(P.ACT))
        ref3 = F.conv2d(F.interpolate((px.double() + pxr.double()).reshape(n_img, hs, ws, cin).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"),
                        w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
        o3 = ops.gemm(px, pw, bias=b.repeat(4), mode=ops.CONV3, conv=(n_img, hs, ws, 1, 2), out_f32=True, a_rest=pxr, w_rest=pwr, tile=tile, split_k=split)
        assert rel_l2(o3, ref3) < 3e-5
