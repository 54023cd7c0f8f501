"""PCM ingest: RIFF/WAVE parsing (host) and sample-format conversion (device).

Mirrors what the reference gets from symphonia (audio_player.rs:169-267): one interleaved f32 buffer,
the sample rate and the channel count.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .analyzer import AnalyzerError, _check


def wav_parse(data: bytes) -> L.WavInfo:
    info = L.WavInfo()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    rc = L.lib().ss_wav_parse(buf, len(data), C.byref(info))
    if rc:
        raise AnalyzerError(rc)
    return info


def pcm_decode(raw: bytes, fmt: int) -> np.ndarray:
    sb = L.lib().ss_pcm_sample_bytes(fmt)
    n = len(raw) // sb
    out = np.empty(n, np.float32)
    buf = (C.c_ubyte * len(raw)).from_buffer_copy(raw)
    _check(L.lib().ss_pcm_decode(buf, n, fmt, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def decode_wav(data: bytes):
    """-> (samples f32 interleaved, sample_rate, channels), like AudioFile::decode_file."""
    info = wav_parse(data)
    fb = L.lib().ss_pcm_sample_bytes(info.format) * info.channels
    raw = data[info.data_offset:info.data_offset + info.frames * fb]
    return pcm_decode(raw, info.format), info.sample_rate, info.channels
