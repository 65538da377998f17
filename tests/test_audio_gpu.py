"""Audio conditioning front-end on the MI355X through the C ABI, against the CPU oracle
(oracle/audio_ref.py; parity unpinned — ImageBind / torchaudio sources absent) and the kernel-contract emulation."""
import math

import numpy as np
import pytest
import torch

from oracle import audio_ref
from tests import emu_ops
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


def _wave(seed, n=40000):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    w = 0.3 * torch.sin(2 * math.pi * 440.0 * t) + 0.1 * torch.sin(2 * math.pi * 3000.0 * t) + 0.05 * torch.randn(n, generator=g)
    return torch.stack([w, -w]) + 0.01


@pytest.mark.parametrize("n", [40000, 32000, 20000, 399])
def test_fbank_matches_oracle(n):
    from asva_amd.audio_features import AudioMelspectrogramExtractor

    w = _wave(n, n)
    feats = AudioMelspectrogramExtractor()([w])
    assert feats.is_cuda and feats.shape == (1, 1, 128, 204)
    ref = audio_ref.waveform_to_melspectrogram_ref(w.numpy())
    assert np.abs(feats[0].cpu().numpy() - ref).max() < 1e-3          # f32 direct DFT vs float64 FFT, normalised log-mel units


def test_fbank_silence_and_batch():
    from asva_amd import ops
    from asva_amd.audio_features import hanning_window, kaldi_mel_banks

    dev = torch.device("cuda")
    win, fb = torch.from_numpy(hanning_window(400)).to(dev), torch.from_numpy(kaldi_mel_banks()).to(dev)
    wave = torch.stack([torch.zeros(32000), _wave(5, 32000)[0]]).to(dev)
    out = ops.kaldi_fbank(wave, win, fb, shift=160, nfft=512, t_out=204)
    assert torch.allclose(out[0, :, :198], torch.full((128, 198), math.log(audio_ref.EPS32), device=dev))
    assert torch.equal(out[:, :, 198:], torch.zeros(2, 128, 6, device=dev))
    emu = emu_ops.kaldi_fbank(wave.cpu(), win.cpu(), fb.cpu(), shift=160, nfft=512, t_out=204)
    assert (out.cpu() - emu).abs().max() < 2e-3
    with pytest.raises(Exception):
        ops.kaldi_fbank(wave, win, fb, shift=160, nfft=500, t_out=204)


def test_patchify_and_tokens_exact():
    from asva_amd import ops

    torch.manual_seed(0)
    x = torch.randn(2, 1, 128, 204)
    p = ops.patchify(x.cuda(), 16, 16, 10)
    assert p.shape == (2 * 228, 256) and torch.equal(p.cpu(), emu_ops.patchify(x, 16, 16, 10))
    x3 = torch.randn(1, 3, 20, 23)
    assert torch.equal(ops.patchify(x3.cuda(), 4, 5, 3).cpu(), emu_ops.patchify(x3, 4, 5, 3))
    emb = torch.randn(2 * 228, 768).bfloat16()
    cls, pos = torch.randn(768), torch.randn(229, 768)
    t = ops.vit_tokens(emb.cuda(), cls.cuda(), pos.cuda(), 2, tail_rows=1)
    assert t.shape == (2 * 230, 768) and torch.equal(t.cpu(), emu_ops.vit_tokens(emb, cls, pos, 2, tail_rows=1))


def test_gemm_gelu_epilogue():
    from asva_amd import ops

    torch.manual_seed(0)
    a, w, b = torch.randn(460, 768).bfloat16(), (0.05 * torch.randn(3072, 768)).bfloat16(), torch.randn(3072)
    ref = emu_ops.gemm(a, w, bias=b, gelu=True)
    for tile, sk in ((0, 1), (3, 1), (6, 2)):
        got = ops.gemm(a.cuda(), w.cuda(), bias=b.cuda(), gelu=True, tile=tile, split_k=sk)
        assert rel_l2(got, ref) < 4e-3


@pytest.mark.parametrize("split", [False, True])
def test_audio_encoder_matches_oracle(split):
    """split = True: the ImageBind audio trunk in split precision (asva_amd/precision.py) against the fp32 oracle on the UN-rounded
    weights, at 2e-4."""
    from asva_amd import precision as P

    P.set_split(split)
    try:
        _audio_encoder_vs_oracle(split)
    finally:
        P.set_split(False)


def _audio_encoder_vs_oracle(split):
    from asva_amd.audio_encoder import ImageBindSegmaskAudioEncoder

    torch.manual_seed(0)
    m = ImageBindSegmaskAudioEncoder(n_segment=12).eval()
    sd = m.state_dict()
    for k, v in sd.items():
        if v.dim() == 1 and v.numel() > 1:
            v.add_(0.1 * torch.randn_like(v))
    m.load_state_dict(sd)
    sd_cpu = {k: v.clone() for k, v in sd.items()}
    mel = torch.cat([torch.randn(2, 1, 128, 204), torch.zeros(1, 1, 128, 204)])      # incl. the null-audio input (:180)
    m = m.cuda()
    cls, enc, masks = m(mel.cuda(), normalize=False, return_dict=False)
    assert enc.is_cuda and enc.shape == (3, 229, 768) and masks.shape == (3, 12, 229)
    sdr = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 and "pos_embed" not in k and "cls_token" not in k and not split else v)
           for k, v in sd_cpu.items()}
    cls_ref, enc_ref = audio_ref.audio_encoder_ref(sdr, mel)
    e_enc, e_cls = rel_l2(enc, enc_ref), rel_l2(cls, cls_ref)
    print(f"audio encoder (split={split}): tokens rel-L2 {e_enc:.3e}, cls {e_cls:.3e}")
    if split:
        assert e_enc < 2e-4 and e_cls < 2e-4
    assert e_enc < 2e-2 and e_cls < 3e-2           # bf16 residual stream over 12 blocks vs fp32
    assert torch.isfinite(enc).all()


def test_pipeline_encode_audio_from_waveforms():
    from asva_amd.audio_encoder import ImageBindSegmaskAudioEncoder
    from asva_amd.audio_features import AudioMelspectrogramExtractor
    from asva_amd.pipeline import AudioCondAnimationPipeline

    torch.manual_seed(0)
    enc = ImageBindSegmaskAudioEncoder(n_segment=12).eval().cuda()
    pipe = AudioCondAnimationPipeline(audio_encoder=enc)
    assert isinstance(pipe.audio_processor, AudioMelspectrogramExtractor)
    a, m = pipe.encode_audio([_wave(1), _wave(2)], video_length=12, do_audio_classifier_free_guidance=True, device=torch.device("cuda"))
    # the product keeps one copy per clip (the kernels share K/V across frames) and one mask table for all clips
    assert a.shape == (4, 229, 768) and m.shape == (12, 229)
    assert torch.equal(a[0], a[1]) and not torch.equal(a[2], a[3])               # [null, null, clip0, clip1]
