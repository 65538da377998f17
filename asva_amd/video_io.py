"""Codec-free video file writer / reader for the generation drivers.

The reference saves every generated clip with torchvision.io.write_video (H.264 + AAC in .mp4,
pipeline_audio_cond_animation.py:451-458).  Neither torchvision nor any H.264/AAC encoder exists in this image, so
`asva_amd.pipeline.write_video` falls back to this module: Motion-JPEG frames (PIL's JPEG encoder) + 16-bit PCM audio in a
RIFF/AVI container — playable by ffmpeg / VLC / OpenCV, and readable back here for the evaluation tools.  Same
arguments as the reference call: video (T, H, W, 3) uint8, fps, audio (C, Ta) float waveform in [-1, 1], audio_fps.
"""
from __future__ import annotations

import io
import struct
from typing import Optional, Tuple

import numpy as np
import torch


def _chunk(tag: bytes, payload: bytes) -> bytes:
    return tag + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")


def _list(kind: bytes, payload: bytes) -> bytes:
    return b"LIST" + struct.pack("<I", len(payload) + 4) + kind + payload


def write_mjpeg_avi(filename: str, video_array, fps: float, audio_array=None, audio_fps: int = 16000, quality: int = 95) -> str:
    from PIL import Image

    video = video_array.detach().cpu().numpy() if torch.is_tensor(video_array) else np.asarray(video_array)
    if video.dtype != np.uint8 or video.ndim != 4 or video.shape[-1] != 3:
        raise ValueError(f"write_mjpeg_avi: video must be uint8 (T, H, W, 3), got {video.dtype} {video.shape}")
    T, H, W, _ = video.shape
    frames = []
    for t in range(T):
        buf = io.BytesIO()
        Image.fromarray(video[t]).save(buf, format="JPEG", quality=quality)
        frames.append(buf.getvalue())
    pcm, nch = b"", 0
    if audio_array is not None:
        a = audio_array.detach().cpu().numpy() if torch.is_tensor(audio_array) else np.asarray(audio_array)
        if a.ndim == 1:
            a = a[None]
        nch = a.shape[0]
        pcm = (np.clip(a.T.astype(np.float64), -1.0, 1.0) * 32767.0).round().astype("<i2").tobytes()      # interleaved
    rate, scale = int(round(fps * 1000)), 1000
    max_frame = max(len(f) for f in frames)
    streams = 1 + (1 if nch else 0)
    avih = struct.pack("<14I", int(round(1e6 / fps)), 0, 0, 0x10, T, 0, streams, max_frame, W, H, 0, 0, 0, 0)
    strh_v = struct.pack("<4s4sIHHIIIIIIII4h", b"vids", b"MJPG", 0, 0, 0, 0, scale, rate, 0, T, max_frame, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf_v = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"MJPG", W * H * 3, 0, 0, 0, 0)
    hdrl = _chunk(b"avih", avih) + _list(b"strl", _chunk(b"strh", strh_v) + _chunk(b"strf", strf_v))
    if nch:
        align = 2 * nch
        nsamp = len(pcm) // align
        strh_a = struct.pack("<4s4sIHHIIIIIIII4h", b"auds", b"\0\0\0\0", 0, 0, 0, 0, align, audio_fps * align, 0, nsamp, len(pcm), 0xFFFFFFFF,
                             align, 0, 0, 0, 0)
        strf_a = struct.pack("<HHIIHHH", 1, nch, audio_fps, audio_fps * align, align, 16, 0)
        hdrl += _list(b"strl", _chunk(b"strh", strh_a) + _chunk(b"strf", strf_a))
    movi, index, off = b"", b"", 4
    items = [(b"00dc", f) for f in frames] + ([(b"01wb", pcm)] if nch else [])
    for tag, data in items:
        index += struct.pack("<4sIII", tag, 0x10, off, len(data))
        c = _chunk(tag, data)
        movi += c
        off += len(c)
    body = b"AVI " + _list(b"hdrl", hdrl) + _list(b"movi", movi) + _chunk(b"idx1", index)
    with open(filename, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return filename


def read_mjpeg_avi(filename: str) -> Tuple[torch.Tensor, float, Optional[torch.Tensor], int]:
    """-> (video (T, H, W, 3) uint8, fps, audio (C, Ta) float or None, audio_fps) of a file written above."""
    from PIL import Image

    with open(filename, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError(f"{filename}: not a RIFF/AVI file")
    frames, pcm, fps, nch, afps = [], b"", 0.0, 0, 16000

    def walk(lo: int, hi: int):
        nonlocal pcm, fps, nch, afps
        p = lo
        while p + 8 <= hi:
            tag, n = data[p:p + 4], struct.unpack("<I", data[p + 4:p + 8])[0]
            body = p + 8
            if tag == b"LIST":
                walk(body + 4, body + n)
            elif tag == b"strh" and data[body:body + 4] == b"vids":
                scale, rate = struct.unpack("<II", data[body + 20:body + 28])
                fps = rate / scale
            elif tag == b"strf" and n == 18:
                _, nch, afps = struct.unpack("<HHI", data[body:body + 8])
            elif tag == b"00dc":
                frames.append(np.array(Image.open(io.BytesIO(data[body:body + n])).convert("RGB")))
            elif tag == b"01wb":
                pcm += data[body:body + n]
            p = body + n + (n & 1)

    walk(12, len(data))
    video = torch.from_numpy(np.stack(frames))
    audio = None
    if nch:
        audio = torch.from_numpy(np.frombuffer(pcm, dtype="<i2").reshape(-1, nch).T.astype(np.float32) / 32767.0)
    return video, fps, audio, afps
