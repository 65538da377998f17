"""Audio conditioning front-end on CPU: the oracle restatement pinned against torch built-ins and
closed-form answers, and the host logic of asva_amd/audio_features.py / audio_encoder.py on the emulated kernel
contracts (tests/emu_ops.py).  ImageBind / torchaudio sources are absent: parity unpinned (oracle/audio_ref.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import audio_ref
from tests import emu_ops
from tests.helpers import rel_l2


# ---- oracle pins ------------------------------------------------------------------------------------------------
def test_oracle_attention_with_bias_kv_matches_torch_mha():
    torch.manual_seed(0)
    mha = torch.nn.MultiheadAttention(64, 4, bias=True, add_bias_kv=True, batch_first=True).eval()
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.1)
        mha.out_proj.bias.normal_(0, 0.1)
    x = torch.randn(3, 17, 64)
    ref = mha(x, x, x, need_weights=False)[0]
    got = audio_ref.mha_bias_kv(x, mha.in_proj_weight, mha.in_proj_bias, mha.bias_k, mha.bias_v, mha.out_proj.weight,
                                mha.out_proj.bias, 4)
    assert torch.allclose(got, ref, atol=1e-5)


def test_oracle_fbank_known_answers():
    sr = 16000
    t = np.arange(2 * sr) / sr
    assert audio_ref.kaldi_fbank(np.zeros(2 * sr)).shape == (198, 128)                    # 1 + (32000 - 400) // 160
    assert np.allclose(audio_ref.kaldi_fbank(np.zeros(2 * sr)), math.log(audio_ref.EPS32))  # silence sits on the log floor
    tone = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    fb = audio_ref.kaldi_fbank(tone)
    lo, hi = audio_ref.mel_scale(20.0), audio_ref.mel_scale(8000.0)
    centre = lo + (np.arange(128) + 1) * (hi - lo) / 129
    expect = int(np.argmin(np.abs(centre - audio_ref.mel_scale(1000.0))))
    assert set(np.argmax(fb, 1)) <= {expect, expect + 1, expect - 1} and np.bincount(np.argmax(fb, 1)).argmax() == expect
    assert np.allclose(audio_ref.kaldi_fbank(tone + 0.25), fb, atol=1e-9)                  # per-frame DC removal
    assert audio_ref.kaldi_fbank(tone[:399]).shape == (0, 128)


def test_product_mel_banks_and_window_match_oracle():
    from asva_amd.audio_features import hanning_window, kaldi_mel_banks

    fb = kaldi_mel_banks(128, 512, 16000.0)
    assert fb.shape == (128, 257) and fb.dtype == np.float32 and not fb[:, 256].any()
    assert np.allclose(fb[:, :256], audio_ref.mel_banks(128, 512, 16000.0), atol=1e-6)
    assert np.allclose(hanning_window(400), torch.hann_window(400, periodic=False).numpy(), atol=1e-6)
    assert (fb.sum(1) == 0).sum() <= 2      # 128 bins over a 512-point FFT leave a low bin or two without an FFT bin (as Kaldi)


# ---- host logic on the emulated kernels -----------------------------------------------------------------------
@pytest.fixture
def emu(monkeypatch):
    import asva_amd.audio_encoder as ae
    import asva_amd.audio_features as af

    monkeypatch.setattr(ae, "ops", emu_ops)
    monkeypatch.setattr(af, "ops", emu_ops)


def _wave(seed, n=40000, ch=2):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    w = 0.3 * torch.sin(2 * math.pi * 440.0 * t) + 0.1 * torch.sin(2 * math.pi * 3000.0 * t) + 0.05 * torch.randn(n, generator=g)
    return torch.stack([w, -w][:ch]) + 0.01


def test_melspectrogram_extractor_contract(emu):
    from asva_amd.audio_features import AudioMelspectrogramExtractor

    ex = AudioMelspectrogramExtractor()
    assert ex.sampling_rate == 16000 and ex.max_length_s == 2
    waves = [_wave(1), _wave(2, n=20000).numpy()]                        # longer than 2 s (centre crop) / shorter (padding)
    feats = ex(waves, device="cpu")
    assert feats.shape == (2, 1, 128, 204) and feats.dtype == torch.float32
    for w, f in zip(waves, feats):
        ref = audio_ref.waveform_to_melspectrogram_ref(np.asarray(w))
        assert np.abs(f.numpy() - ref).max() < 1e-3
    pad = (0.0 + 4.268) / 9.138
    assert torch.allclose(feats[0, 0, :, 198:], torch.full((128, 6), pad))               # 198 frames, then normalised zeros
    n_short = 1 + (20000 - 400) // 160
    assert torch.allclose(feats[1, 0, :, n_short:], torch.full((128, 204 - n_short), pad))
    assert ex(_wave(3), device="cpu").shape == (1, 1, 128, 204)                          # single (c, n) waveform


def test_audio_encoder_state_dict_layout():
    from asva_amd.audio_encoder import ImageBindSegmaskAudioEncoder

    m = ImageBindSegmaskAudioEncoder(n_segment=12)
    sd = m.state_dict()
    assert len(sd) == 5 + 12 * 14 + 3 + 1 + 2
    assert sd["preprocessor.rgbt_stem.proj.weight"].shape == (768, 1, 16, 16)
    assert sd["preprocessor.pos_embedding_helper.pos_embed"].shape == (1, 229, 768)
    assert sd["trunk.blocks.11.attn.in_proj_weight"].shape == (2304, 768) and sd["trunk.blocks.0.attn.bias_k"].shape == (1, 1, 768)
    assert sd["trunk.blocks.3.mlp.fc1.weight"].shape == (3072, 768) and sd["head.2.weight"].shape == (1024, 768)
    assert sd["final_layer_norm.weight"].shape == (768,) and "postprocessor.1.log_logit_scale" in sd
    assert m.config.n_segment == 12 and m.config.pretrained_model_name == "imagebind-huge"


def test_audio_encoder_forward_vs_oracle_and_io(emu, tmp_path):
    from asva_amd.audio_encoder import ImageBindSegmaskAudioEncoder
    from asva_amd.conditioning import audio_segment_mask

    torch.manual_seed(0)
    m = ImageBindSegmaskAudioEncoder(n_segment=12).eval()
    sd = m.state_dict()
    for k, v in sd.items():                                             # non-trivial norms / biases
        if v.dim() == 1 and v.numel() > 1:
            v.add_(0.1 * torch.randn_like(v))
    m.load_state_dict(sd)
    mel = torch.randn(2, 1, 128, 204)
    cls, enc, masks = m(mel, normalize=False, return_dict=False)
    sdr = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 and "pos_embed" not in k and "cls_token" not in k else v) for k, v in sd.items()}
    cls_ref, enc_ref = audio_ref.audio_encoder_ref(sdr, mel)
    assert enc.shape == (2, 229, 768) and cls.shape == (2, 1024)
    assert rel_l2(enc, enc_ref) < 2e-2 and rel_l2(cls, cls_ref) < 3e-2
    assert masks.shape == (2, 12, 229) and masks.dtype == torch.bool and torch.equal(masks[1], audio_segment_mask(12))
    out = m(mel, return_dict=True)
    assert torch.equal(out.audio_encodings, enc) and out.to_tuple()[0].shape == (2, 1024)
    with pytest.raises(NotImplementedError):
        m(mel, normalize=True)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 128, 200))
    m.save_pretrained(str(tmp_path / "audio_encoder"))
    m2 = ImageBindSegmaskAudioEncoder.from_pretrained(str(tmp_path), subfolder="audio_encoder")
    assert m2.n_segment == 12 and all(torch.equal(v, m2.state_dict()[k]) for k, v in m.state_dict().items())


def test_imagebind_checkpoint_mapping(tmp_path):
    """ImageBind's own checkpoint (whole multi-modal state dict) -> the audio branch; final_layer_norm stays identity."""
    from asva_amd.audio_encoder import ImageBindSegmaskAudioEncoder

    m = ImageBindSegmaskAudioEncoder(n_segment=12)
    ren = {"preprocessor.": "modality_preprocessors.audio.", "trunk.": "modality_trunks.audio.", "head.": "modality_heads.audio.",
           "postprocessor.": "modality_postprocessors.audio."}
    full = {"modality_trunks.vision.blocks.0.attn.in_proj_weight": torch.zeros(3)}
    for k, v in m.state_dict().items():
        for new, old in ren.items():
            if k.startswith(new):
                full[old + k[len(new):]] = v + 1
    path = str(tmp_path / "imagebind_huge.pth")
    torch.save(full, path)
    m2 = ImageBindSegmaskAudioEncoder(n_segment=12, imagebind_checkpoint=path)
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert all(torch.equal(sd2[k], sd[k] + 1) for k in sd if not k.startswith("final_layer_norm"))
    assert torch.equal(sd2["final_layer_norm.weight"], torch.ones(768)) and not sd2["final_layer_norm.bias"].any()
    del full["modality_trunks.audio.blocks.5.attn.bias_k"]
    torch.save(full, path)
    with pytest.raises(KeyError):
        ImageBindSegmaskAudioEncoder(imagebind_checkpoint=path)


def test_reference_import_paths():
    from avgen.data.utils import AudioMelspectrogramExtractor as A, waveform_to_melspectrogram  # noqa: F401
    from avgen.models.audio_encoders import ImageBindSegmaskAudioEncoder as E
    import asva_amd.audio_encoder as ae
    import asva_amd.audio_features as af

    assert A is af.AudioMelspectrogramExtractor and E is ae.ImageBindSegmaskAudioEncoder
