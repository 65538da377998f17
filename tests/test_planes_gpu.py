"""Explicit (main, rest) planes of the per-layer precision plan (asva_amd/precision.py; -m gpu): the kernels behind
`ops.gemm(..., out_rest=)` (AVSD_GEMM_OUT_REST: a one-pass product also writes the rest plane of its 16-bit output) and
`ops.gemm(..., a_rest=, w_rest=, out=, out_rest=, master=)` (a three-pass product writing two planes and the f32 value), which replaced
the avsd_split_f32 launches of round 5 (68 per step).  What is checked is the CONTRACT the consumers rely on, bit for bit:
    rest == round16(master - main)     (exactly what avsd_split_f32 computes from the f32 master)
and that neither option changes the other outputs of the launch.  Reference call sites: the tensors concerned are the ResBlock /
transformer outputs feeding conv_shortcut (ff_spatio_temp_resnet_3d.py:159,189) and the samplers (:30-62,:88-96).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0, dtype=None):
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = (torch.randn(*shape, generator=g) * scale).to(dev())
    return t if dtype is None else t.to(dtype)


@pytest.fixture(scope="module", params=["bf16", "fp16"])
def ops(request):
    from asva_amd import ops as _ops, precision as P

    P.set_split(False)
    P.set_precision(request.param)
    yield _ops
    P.set_precision("bf16")


def one_pass_rest_built(ops):
    """AVSD_GEMM_OUT_REST is compiled into the IEEE-half build only (the per-layer precision plan is fp16 storage; csrc/gemm_common.h EPI_REST):
    the bfloat16 library refuses the flag loudly instead of ignoring it"""
    if ops.P.NAME == "fp16":
        return True
    from asva_amd._lib import AvsdError

    a = rnd(64, 64, seed=1, dtype=ops.P.ACT)
    out, rest = ops.alloc_planes((64, 64), dev())
    with pytest.raises(AvsdError, match="IEEE-half build"):
        ops.gemm(a, a, out=out, out_rest=rest)
    return False


def planes_of(ops, x):
    from asva_amd import precision as P

    main = x.to(P.ACT)
    return main, (x - main.float()).to(P.ACT)


def check_rest(ops, out, rest, master):
    from asva_amd import precision as P

    assert torch.equal(out, master.to(P.ACT))
    assert torch.equal(rest, (master - out.float()).to(P.ACT))


# tile ids: 4 / 6 / 9 small LDS-direct tiles (term-at-a-time epilogue), 17 / 19 / 38 big ones (fragment-at-a-time / epilogue_each),
# 3 register-staged, 60-67 hand-scheduled, 0 = table / heuristic
@pytest.mark.parametrize("tile,split_k", [(0, 1), (3, 1), (4, 1), (6, 1), (9, 1), (17, 1), (19, 1), (38, 1), (6, 2), (9, 4), (60, 1), (63, 1), (66, 1), (67, 1), (63, 2)])
@pytest.mark.parametrize("M,N,K", [(1536, 640, 640), (520, 320, 1280), (96, 132, 64)])
def test_one_pass_gemm_writes_the_rest_plane(ops, tile, split_k, M, N, K):
    if tile >= 60 and (K % 64 or (tile == 67 and N % 320)):
        pytest.skip("asm tiles: K % 64 == 0 (tile 67: N = 320 k)")
    if split_k > (K + 63) // 64 // 2:
        pytest.skip("too few K tiles for this split")
    if not one_pass_rest_built(ops):
        return
    a = rnd(M, K, seed=1, dtype=ops.P.ACT)
    w = rnd(N, K, seed=2, scale=K ** -0.5, dtype=ops.P.ACT)
    bias = rnd(N, seed=3)
    res = rnd(M, N, seed=4)
    plain_master = torch.empty((M, N), dtype=torch.float32, device=dev())
    plain = ops.gemm(a, w, bias=bias, res1=res, master=plain_master, tile=tile, split_k=split_k)
    out, rest = ops.alloc_planes((M, N), dev())
    rest.fill_(7.0)
    master = torch.empty((M, N), dtype=torch.float32, device=dev())
    r = ops.gemm(a, w, bias=bias, res1=res, master=master, out=out, out_rest=rest, tile=tile, split_k=split_k)
    assert r.data_ptr() == out.data_ptr()
    assert torch.equal(out, plain) and torch.equal(master, plain_master)      # the flag changes nothing else
    check_rest(ops, out, rest, master)
    ref = a.float() @ w.float().T + bias + res
    assert ((out.float() + rest.float() - ref).norm() / ref.norm()).item() < 3e-5
    assert ((out.float() - ref).norm() / ref.norm()).item() > 1e-4               # ... which the main plane alone does not reach


@pytest.mark.parametrize("hw,frames,C", [(64, 4, 320), (256, 12, 640), (16, 3, 64)])
def test_temporal_mix_and_convolution_write_the_rest_plane(ops, hw, frames, C):
    """the producers of the plan's stream tensors: the temporal mix behind every FFInflatedConv3d (utils.py:43-53) with rowvec and two f32
    residuals, and the 3x3 convolution in front of a three-pass temporal mix (strided / upsampling: the LDS-direct tiles)"""
    if not one_pass_rest_built(ops):
        return
    B = 2
    M = B * frames * hw
    y = rnd(M, C, seed=1, dtype=ops.P.ACT)
    wt = rnd(C, 3 * C, seed=2, scale=(3 * C) ** -0.5, dtype=ops.P.ACT)
    bt = rnd(C, seed=3)
    r1, r2 = rnd(M, C, seed=4), rnd(M, C, seed=5)
    temb = rnd(B, C, seed=6)
    kw = dict(bias=bt, res1=r1, res2=r2, rowvec=temb, rows_per_vec=frames * hw, mode=ops.TMIX, tmix=(hw, frames))
    pm = torch.empty((M, C), dtype=torch.float32, device=dev())
    plain = ops.gemm(y, wt, master=pm, **kw)
    out, rest = ops.alloc_planes((M, C), dev())
    master = torch.empty_like(pm)
    ops.gemm(y, wt, master=master, out=out, out_rest=rest, **kw)
    assert torch.equal(out, plain) and torch.equal(master, pm)
    check_rest(ops, out, rest, master)
    # 3x3 convolution, square image of hw pixels
    side = int(hw ** 0.5)
    n_img = B * frames
    x = rnd(n_img * side * side, C, seed=7, dtype=ops.P.ACT)
    wc = rnd(C, 9 * C, seed=8, scale=(9 * C) ** -0.5, dtype=ops.P.ACT)
    for stride, ups in ((1, 0), (2, 0), (1, 1)):
        if stride == 2 and side % 2:
            continue
        conv = (n_img, side, side, stride, ups)
        rows = n_img * (((side << ups) - 1) // stride + 1) ** 2
        tile = 13 if stride == 1 and ups == 0 else 0          # (stride 1: the table would pick an LDS-resident tile, which has no rest-plane store)
        pm = torch.empty((rows, C), dtype=torch.float32, device=dev())
        plain = ops.gemm(x, wc, bias=bt, mode=ops.CONV3, conv=conv, master=pm, tile=tile)
        out, rest = ops.alloc_planes((rows, C), dev())
        master = torch.empty_like(pm)
        ops.gemm(x, wc, bias=bt, mode=ops.CONV3, conv=conv, master=master, out=out, out_rest=rest, tile=tile)
        assert torch.equal(out, plain) and torch.equal(master, pm)
        check_rest(ops, out, rest, master)


@pytest.mark.parametrize("tile,split_k", [(0, 1), (7, 1), (11, 1), (25, 1), (7, 2), (63, 1), (64, 1)])
@pytest.mark.parametrize("M,N,K", [(1536, 640, 640), (264, 320, 1280)])
def test_three_pass_product_writes_planes_and_master(ops, tile, split_k, M, N, K):
    """gemm(a_rest=, w_rest=) with (out, out_rest, master) against the same product with an f32 output (the round-5 form + avsd_split_f32)"""
    af, wf = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    a, ar = planes_of(ops, af)
    w, wr = planes_of(ops, wf)
    a, ar = [t.contiguous() for t in (a, ar)]
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    pa, par = ops.alloc_planes((M, K), dev())
    pa.copy_(a), par.copy_(ar)
    pw, pwr = ops.alloc_planes((N, K), dev())
    pw.copy_(w), pwr.copy_(wr)
    f32 = ops.gemm(pa, pw, bias=bias, res1=res, out_f32=True, a_rest=par, w_rest=pwr, tile=tile, split_k=split_k)
    out, rest = ops.alloc_planes((M, N), dev())
    master = torch.empty((M, N), dtype=torch.float32, device=dev())
    ops.gemm(pa, pw, bias=bias, res1=res, out=out, out_rest=rest, master=master, a_rest=par, w_rest=pwr, tile=tile, split_k=split_k)
    assert torch.equal(master, f32)
    check_rest(ops, out, rest, master)
    m2, r2 = ops.split_planes(f32)                   # what round 5 launched behind the product
    assert torch.equal(out, m2) and torch.equal(rest, r2)
    ref = (a.double() + ar.double()) @ (w.double() + wr.double()).T + bias.double() + res.double()
    assert ((f32.double() - ref).norm() / ref.norm()).item() < 2e-5
    # without a master: the planes alone
    out2, rest2 = ops.alloc_planes((M, N), dev())
    ops.gemm(pa, pw, bias=bias, res1=res, out=out2, out_rest=rest2, a_rest=par, w_rest=pwr, tile=tile, split_k=split_k)
    assert torch.equal(out2, out) and torch.equal(rest2, rest)


def test_out_rest_argument_checks(ops):
    a = rnd(64, 64, seed=1, dtype=ops.P.ACT)
    w = rnd(64, 64, seed=2, dtype=ops.P.ACT)
    out, rest = ops.alloc_planes((64, 64), dev())
    with pytest.raises(ValueError):
        ops.gemm(a, w, out_rest=rest)                                     # needs the explicit main plane
    with pytest.raises(ValueError):
        ops.gemm(a, w, out=out, out_rest=out)                             # a distinct tensor
    with pytest.raises((ValueError, TypeError)):
        ops.gemm(a, w, out=torch.empty((64, 64), dtype=torch.float32, device=dev()), out_f32=True, out_rest=rest)
    w1 = rnd(64, 64, seed=3, dtype=ops.P.ACT)
    with pytest.raises(ValueError):
        ops.gemm(a, w1, geglu=True, out=out[:, :32], out_rest=rest[:, :32])
