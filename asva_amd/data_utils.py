"""Dataset / file loaders the generation drivers call — host-side mirror of the reference's
avgen/data/utils.py (:118-470): `load_image`, `load_av_clips_uniformly`, `load_video_clips_uniformly`,
`load_audio_clips_uniformly`, `load_and_transform_images_stable_diffusion`, `get_evaluation_data` with the same
names, arguments, return shapes and value ranges.

Backends.  The reference decodes with torchvision.io.VideoReader / torchaudio / PIL.  This image has PIL, scipy and
numpy only, so
  * images: PIL (as the reference);
  * audio files: torchaudio when importable, else `.wav` through scipy.io.wavfile; resampling through
    torchaudio.functional.resample when importable, else scipy.signal.resample_poly (both are polyphase windowed-sinc
    resamplers; their filters differ, so waveforms agree to ~1e-3, not bit for bit);
  * video files: torchvision's VideoReader when importable; additionally a pre-decoded clip container
    (`.npz` with `frames` uint8 (T, H, W, 3), `fps`, optional `audio` f32 (C, T) + `audio_sr`) that needs no codec —
    what the synthetic dataset driver test and offline-decoded datasets use.  A real `.mp4` without torchvision
    raises a RuntimeError naming the missing backend (no silent fallback).
The crop / resize arithmetic (centre crop to the target aspect ratio with the reference's integer rounding, antialiased
bilinear resize of the shorter side, centre crop) follows :118-196.
"""
from __future__ import annotations

import os
from typing import List, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from .audio_features import waveform_to_melspectrogram


# ---- images (:118-196, :371-386) --------------------------------------------------------------------------------------
def _resize_shorter_side(images: torch.Tensor, size) -> torch.Tensor:
    """torchvision.transforms.Resize(size, BILINEAR, antialias=True): an int resizes the SHORTER side to `size`
    keeping the aspect ratio (long side = int(size * long / short)); a pair resizes to exactly (h, w)."""
    h, w = images.shape[-2:]
    if isinstance(size, int):
        if h <= w:
            nh, nw = size, int(size * w / h)
        else:
            nh, nw = int(size * h / w), size
    else:
        nh, nw = size
    if (nh, nw) == (h, w):
        return images
    return F.interpolate(images, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)


def _center_crop(images: torch.Tensor, size) -> torch.Tensor:
    th, tw = (size, size) if isinstance(size, int) else size
    h, w = images.shape[-2:]
    top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
    return images[..., top:top + th, left:left + tw]


def load_and_transform_images_stable_diffusion(images, size=512, flip: bool = False, randcrop: bool = False,
                                               normalize: bool = True) -> torch.Tensor:
    """(List of) uint8 (h, w, 3) arrays, a uint8 (f, h, w, 3) array or a float (b, 3, h, w) tensor in [0, 1]
    -> (b, 3, H, W); normalize=True maps to [-1, 1] (:118-196)."""
    if isinstance(images, (list, tuple)):
        images = np.stack(images, axis=0)
    if isinstance(images, np.ndarray):
        assert images.dtype == np.uint8 and images.shape[-1] == 3, (images.dtype, images.shape)
        images = torch.from_numpy(images).permute(0, 3, 1, 2).float() / 255.0
    assert images.shape[1] == 3
    assert bool((images <= 1.0).all()) and bool((images >= 0.0).all())
    if randcrop:
        raise NotImplementedError("randcrop is a training-time augmentation; the generation drivers never set it")
    h, w = images.shape[-2:]
    th, tw = (size, size) if isinstance(size, int) else size
    target_ar, cur_ar = float(th) / tw, float(h) / w
    if target_ar >= cur_ar:                                    # trim w
        tw_ = int(h / target_ar)
        images = images[:, :, :, (w - tw_) // 2:(w - tw_) // 2 + tw_]
    else:                                                      # trim h
        th_ = int(w * target_ar)
        images = images[:, :, (h - th_) // 2:(h - th_) // 2 + th_]
    images = _center_crop(_resize_shorter_side(images, size), size)
    if flip:
        images = images.flip(-1)
    if normalize:
        images = (images - 0.5) / 0.5
    return images.clamp(-1.0 if normalize else 0.0, 1.0)


def load_image(image_path: str, image_size=(256, 256)) -> torch.Tensor:
    """-> (3, h, w) in [0, 1] (:371-386)."""
    from PIL import Image

    img = torch.from_numpy(np.array(Image.open(image_path).convert("RGB")))
    img = img.permute(2, 0, 1)[None].float() / 255.0
    return load_and_transform_images_stable_diffusion(img, size=image_size, normalize=False)[0].contiguous()


# ---- audio ------------------------------------------------------------------------------------------------------------
def _resample(audio: torch.Tensor, orig: int, new: int = 16000) -> torch.Tensor:
    if orig == new:
        return audio
    try:
        import torchaudio  # type: ignore

        return torchaudio.functional.resample(audio, orig_freq=orig, new_freq=new)
    except ImportError:
        from math import gcd

        from scipy.signal import resample_poly

        g = gcd(int(orig), int(new))
        return torch.from_numpy(resample_poly(audio.numpy().astype(np.float64), new // g, orig // g, axis=1).astype(np.float32))


def _load_audio_file(path: str) -> Tuple[torch.Tensor, int]:
    """-> (C, T) f32 in [-1, 1], sample rate."""
    try:
        import torchaudio  # type: ignore

        return torchaudio.load(path)
    except ImportError:
        pass
    if not path.lower().endswith(".wav"):
        raise RuntimeError(f"load_audio_clips_uniformly({path}): torchaudio is not installed in this image and only .wav "
                           "files can be read without it (scipy.io.wavfile)")
    from scipy.io import wavfile

    sr, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        data = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    return torch.from_numpy(np.ascontiguousarray(data.T.astype(np.float32))), int(sr)


def _clip_starts(duration: float, clip_duration: float, num_clips: int) -> np.ndarray:
    """:221-225 / :409-413 — one clip: centred; several: uniformly spaced, both ends included."""
    if num_clips == 1:
        return np.array([(duration - clip_duration) / 2.0])
    return np.linspace(0.0, duration - clip_duration, endpoint=True, num=num_clips)


def load_audio_clips_uniformly(audio_path: str, clip_duration: float = 2.0, num_clips: int = 1,
                               load_audio_as_melspectrogram: bool = True):
    """-> (b, 1, n, t) mel-spectrograms, or a list of b waveforms (c, t) at 16 kHz (:389-424)."""
    audio, sr = _load_audio_file(audio_path)
    duration = audio.shape[1] / float(sr)
    audio = _resample(audio, sr, 16000)
    clips = []
    for t0 in _clip_starts(duration, clip_duration, num_clips):
        clip = audio[:, int(t0 * 16000):int((t0 + clip_duration) * 16000)].contiguous()
        clips.append(waveform_to_melspectrogram(clip) if load_audio_as_melspectrogram else clip)
    return torch.stack(clips) if load_audio_as_melspectrogram else clips


# ---- video ------------------------------------------------------------------------------------------------------------
class _DecodedClip:
    """Pre-decoded container: frames uint8 (T, H, W, 3) at `fps`, optional audio f32 (C, Ta) at `audio_sr`."""

    def __init__(self, path: str):
        z = np.load(path)
        self.frames = z["frames"]
        self.fps = float(z["fps"])
        self.audio = torch.from_numpy(z["audio"].astype(np.float32)) if "audio" in z.files else None
        self.audio_sr = int(z["audio_sr"]) if "audio_sr" in z.files else 16000
        self.video_duration = self.frames.shape[0] / self.fps
        self.audio_duration = self.audio.shape[1] / float(self.audio_sr) if self.audio is not None else self.video_duration

    def video_clip(self, t0: float, clip_duration: float, video_fps: int, n_frame: int) -> torch.Tensor:
        """The frame-picking rule of load_video_clip_from_videoreader (:199-231): walk the frames with pts in
        [t0, t0 + duration + half a key-frame period], keep the first frame at or after each multiple of 1/video_fps,
        repeat the last kept frame when the stream ends early."""
        period = 1.0 / video_fps
        picked, want = [], t0
        first = max(int(np.floor(max(t0, 0.0) * self.fps)), 0)
        for i in range(first, self.frames.shape[0]):
            pts = i / self.fps
            if pts > t0 + clip_duration + period / 2.0:
                break
            if pts >= want:
                picked.append(self.frames[i])
                want += period
            if len(picked) == n_frame:
                break
        if not picked:
            raise ValueError("no frame inside the requested clip window")
        picked += [picked[-1]] * (n_frame - len(picked))
        return torch.from_numpy(np.stack(picked)).permute(0, 3, 1, 2).float() / 255.0

    def audio_clip(self, t0: float, clip_duration: float) -> torch.Tensor:
        a = self.audio[:, max(int(round(t0 * self.audio_sr)), 0):int(round((t0 + clip_duration) * self.audio_sr))]
        return _resample(a.contiguous(), self.audio_sr, 16000)


def _open_video(video_path: str):
    if video_path.lower().endswith(".npz"):
        return _DecodedClip(video_path)
    try:
        import torchvision  # type: ignore  # noqa: F401
        from torchvision.io import VideoReader  # type: ignore  # noqa: F401
    except ImportError as e:
        raise RuntimeError(f"load_av_clips_uniformly({video_path}): decoding a video file needs torchvision.io.VideoReader, "
                           "which is not installed in this image; pre-decode to the .npz clip container "
                           "(asva_amd.data_utils._DecodedClip) or pass decoded clips with clips=...") from e
    return _TorchvisionClip(video_path)


class _TorchvisionClip:
    """torchvision.io.VideoReader backend (the reference's decoder), same interface as _DecodedClip."""

    def __init__(self, path: str):
        import itertools

        import torchvision  # type: ignore
        from torchvision.io import VideoReader  # type: ignore

        torchvision.set_video_backend("video_reader")
        self._it = itertools
        self.reader = VideoReader(path, stream="video")
        md = self.reader.get_metadata()
        self.video_duration = float(md["video"]["duration"][0])
        self.fps = float(md["video"]["fps"][0])
        self.has_audio = "audio" in md and len(md["audio"].get("duration", [])) > 0
        self.audio_duration = float(md["audio"]["duration"][0]) if self.has_audio else self.video_duration
        self.audio_sr = int(md["audio"]["framerate"][0]) if self.has_audio else 16000

    def video_clip(self, t0, clip_duration, video_fps, n_frame):
        self.reader.set_current_stream("video")
        period = 1.0 / video_fps
        picked, want = [], t0
        for fr in self._it.takewhile(lambda x: x["pts"] <= t0 + clip_duration + period / 2.0, self.reader.seek(max(t0, 0.0))):
            if fr["pts"] >= want:
                picked.append(fr["data"])
                want += period
            if len(picked) == n_frame:
                break
        picked += [picked[-1]] * (n_frame - len(picked))
        return torch.stack(picked).float() / 255.0

    def audio_clip(self, t0, clip_duration):
        self.reader.set_current_stream("audio")
        chunks = []
        for fr in self._it.takewhile(lambda x: x["pts"] <= t0 + clip_duration, self.reader.seek(t0)):
            if t0 <= fr["pts"] <= t0 + clip_duration:
                d = fr["data"]
                t, c = d.shape
                chunks.append(d.contiguous().view(c, t).contiguous())     # the reference's (t, c) -> (c, t) VIEW (:252-254)
        return _resample(torch.cat(chunks, dim=1), self.audio_sr, 16000)


def load_av_clips_uniformly(video_path: str, video_fps: int = 6, video_num_frame: int = 12,
                            image_size: Union[int, Tuple[int, int]] = 512, num_clips: int = 1,
                            load_audio_as_melspectrogram: bool = True):
    """-> video (b, f, 3, h, w) in [0, 1]; audio (b, 1, n, t) mel-spectrograms or a list of b waveforms (c, t_i)
    (:268-334)."""
    clip_duration = video_num_frame / video_fps
    src = _open_video(video_path)
    av_duration = min(src.video_duration, src.audio_duration)
    videos, audios = [], []
    for t0 in _clip_starts(av_duration, clip_duration, num_clips):
        fr = src.video_clip(float(t0), clip_duration, video_fps, video_num_frame)
        videos.append(load_and_transform_images_stable_diffusion(fr, size=image_size, normalize=False).float())
        au = src.audio_clip(float(t0), clip_duration)
        audios.append(waveform_to_melspectrogram(au) if load_audio_as_melspectrogram else au)
    videos = torch.stack(videos)
    if load_audio_as_melspectrogram:
        audios = torch.stack(audios)
    return videos, audios


def load_video_clips_uniformly(video_path: str, video_fps: int = 6, video_num_frame: int = 12,
                               image_size: Union[int, Tuple[int, int]] = 512, num_clips: int = 1) -> torch.Tensor:
    """-> (b, f, 3, h, w) in [0, 1] (:337-368)."""
    clip_duration = video_num_frame / video_fps
    src = _open_video(video_path)
    out = []
    for t0 in _clip_starts(src.video_duration, clip_duration, num_clips):
        fr = src.video_clip(float(t0), clip_duration, video_fps, video_num_frame)
        out.append(load_and_transform_images_stable_diffusion(fr, size=image_size, normalize=False).float())
    return torch.stack(out)


# ---- evaluation lists (:427-470) -----------------------------------------------------------------------------------------
def _datasets_root() -> str:
    return os.environ.get("AVSD_DATASETS_ROOT", "./datasets")


def _read_list(path: str) -> List[str]:
    with open(path) as f:
        return [ln.strip() for ln in f.readlines()]


def get_evaluation_data(dataset: str):
    """-> (video_root, relative video paths, categories, "video").  Same directory layout as the reference
    (`./datasets/<name>/test.txt`, videos under `videos/` — `videos/test/` for Landscapes); AVSD_DATASETS_ROOT moves
    the `./datasets` prefix."""
    root = f"{_datasets_root()}/{dataset}"
    if dataset == "AVSync15":
        video_root, paths = f"{root}/videos", _read_list(f"{root}/test.txt")
        cats = [p.split("/")[0] for p in paths]
    elif dataset == "TheGreatestHits":
        video_root, paths = f"{root}/videos", _read_list(f"{root}/test.txt")
        cats = ["hitting with a stick"] * len(paths)
    elif dataset == "Landscapes":
        video_root, paths = f"{root}/videos/test", _read_list(f"{root}/test.txt")
        cats = [p.split("/")[0] for p in paths]
    else:
        raise Exception()
    return video_root, paths, cats, "video"
