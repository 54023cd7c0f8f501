"""Pipelined corpus analysis: host-resident PCM in, per-stream loudness results out (the callers' side of the path).

Two `Batch` objects are a double buffer — each owns its HIP stream, so chunk k+1's upload (copy engine) runs while
chunk k is analysed.  The host corpus is page-locked in place (`ss_host_register`) and uploaded with
`ss_batch_upload_pcm_async` as raw PCM (s16 input halves the PCIe bytes; the conversion to f32 is a kernel, N2).
Results that stay small (loudness, peaks, histograms) come back per chunk; spectra stay on the device unless asked for.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .analyzer import _check
from .batch import Batch

_NP_FORMAT = {np.dtype(np.uint8): L.SS_PCM_U8, np.dtype(np.int16): L.SS_PCM_S16, np.dtype(np.int32): L.SS_PCM_S32,
              np.dtype(np.float32): L.SS_PCM_F32, np.dtype(np.float64): L.SS_PCM_F64}


def analyze_corpus(pcm, sample_rate, channels, frames_per_stream, chunk_streams=256,
                   flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK, fft_n=4096, hop_frames=1024, on_chunk=None):
    """pcm: one contiguous numpy array holding n_streams equal-length interleaved streams (u8/s16/s32/f32/f64).
    Returns (results, corpus_hist): `results` is a list of StreamResult-like tuples
    (integrated, lra, true_peak[2], sample_peak[2]) per stream, `corpus_hist` the summed 2 x 1000 histograms.
    `on_chunk(batch, first_stream, count)` is called after each chunk's pass while its outputs are still resident."""
    a = np.ascontiguousarray(pcm)
    fmt = _NP_FORMAT[a.dtype]
    per = frames_per_stream * channels
    n_streams = a.size // per
    assert n_streams * per == a.size
    chunk_streams = max(1, min(chunk_streams, n_streams))
    lib = L.lib()
    pinned = lib.ss_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes) == L.SS_OK
    bufs = [Batch(sample_rate, channels, chunk_streams, frames_per_stream, fft_n, hop_frames, flags=flags) for _ in range(2)]
    pending = [None, None]                                  # (first, count) in flight on each buffer
    results = [None] * n_streams
    hist = np.zeros(2000, np.uint64)

    def collect(slot):
        if pending[slot] is None:
            return
        first, count = pending[slot]
        b = bufs[slot]
        b.sync()
        if on_chunk is not None:
            on_chunk(b, first, count)
        r = b.results()
        for i in range(count):
            results[first + i] = (r[i].integrated_lufs, r[i].loudness_range, tuple(r[i].true_peak), tuple(r[i].sample_peak))
        hb, hs = b.histograms()
        hist[:1000] += hb
        hist[1000:] += hs
        pending[slot] = None

    try:
        k = 0
        full = n_streams - n_streams % chunk_streams
        for first in range(0, full, chunk_streams):
            slot = k & 1
            collect(slot)                                    # the buffer's previous chunk must be done before its input is overwritten
            src = a.reshape(-1)[first * per:(first + chunk_streams) * per]
            _check(lib.ss_batch_upload_pcm_async(bufs[slot]._h, 0, chunk_streams, src.ctypes.data_as(C.c_void_p), fmt))
            bufs[slot].run()
            pending[slot] = (first, chunk_streams)
            k += 1
        collect(k & 1)
        collect((k + 1) & 1)
        if full < n_streams:                                 # the short last chunk gets a batch of its own size
            count = n_streams - full
            tail = Batch(sample_rate, channels, count, frames_per_stream, fft_n, hop_frames, flags=flags)
            bufs.append(tail)
            pending.append((full, count))
            src = a.reshape(-1)[full * per:]
            _check(lib.ss_batch_upload_pcm_async(tail._h, 0, count, src.ctypes.data_as(C.c_void_p), fmt))
            tail.run()
            collect(2)
    finally:
        for b in bufs:
            b.sync()
            b.close()
        if pinned:
            lib.ss_host_unregister(a.ctypes.data_as(C.c_void_p))
    return results, hist


def analyze_streams(streams, sample_rate, channels, chunk_streams=256,
                    flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK, fft_n=4096, hop_frames=1024, on_chunk=None):
    """Streams of DIFFERENT lengths (a list of interleaved numpy arrays of one dtype, e.g. decoded files): sorted
    by length, cut into chunks, each chunk one ragged batch (`ss_batch_set_lengths`) whose slot is its longest
    stream, every stream uploaded straight from its own array.  Returns (results, corpus_hist) in input order;
    `on_chunk(batch, indices)` sees each chunk's batch while its spectra / waveforms are still resident."""
    n = len(streams)
    arrs = [np.ascontiguousarray(x) for x in streams]
    if n == 0:
        return [], np.zeros(2000, np.uint64)
    fmt = _NP_FORMAT[arrs[0].dtype]
    order = sorted(range(n), key=lambda i: arrs[i].size)
    results = [None] * n
    hist = np.zeros(2000, np.uint64)
    lib = L.lib()
    for c0 in range(0, n, chunk_streams):
        idx = order[c0:c0 + chunk_streams]
        frames = [arrs[i].size // channels for i in idx]
        slot = max(max(frames), 1)
        b = Batch(sample_rate, channels, len(idx), slot, fft_n, hop_frames, flags=flags)
        try:
            b.set_lengths(frames)
            for k, i in enumerate(idx):
                if frames[k]:
                    # one slot at a time: a slot-sized "count" would read past the end of a shorter array
                    _check(_upload_stream(lib, b, k, arrs[i], frames[k] * channels, fmt))
            b.run()
            b.sync()
            if on_chunk is not None:
                on_chunk(b, idx)
            r = b.results()
            for k, i in enumerate(idx):
                results[i] = (r[k].integrated_lufs, r[k].loudness_range, tuple(r[k].true_peak), tuple(r[k].sample_peak))
            hb, hs = b.histograms()
            hist[:1000] += hb
            hist[1000:] += hs
        finally:
            b.close()
    return results, hist


def _upload_stream(lib, batch, slot_index, arr, n_samples, fmt):
    """Upload n_samples of `arr` into the head of slot `slot_index` (the rest of the slot is never read)."""
    return lib.ss_batch_upload_samples(batch._h, slot_index, arr.ctypes.data_as(C.c_void_p), n_samples, fmt)


def analyze_wav_files(paths, chunk_streams=256, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK, fft_n=4096,
                      hop_frames=1024, on_chunk=None):
    """RIFF/WAVE files -> per-file loudness results, nothing decoded on the host: the header is walked by
    `ss_wav_parse`, the data chunk goes up as raw PCM (`ss_batch_upload_samples`: u8 / s16 / s24 / s32 / f32 / f64)
    and is converted by the ingest kernel.  Files are grouped by (rate, channels, format) and each group runs as
    ragged batches.  Returns {path: (integrated, lra, true_peak[2], sample_peak[2])}; unreadable or unsupported
    files map to the AnalyzerError they raised."""
    from .analyzer import AnalyzerError
    groups, out = {}, {}
    for p in paths:
        try:
            data = np.fromfile(p, dtype=np.uint8)
            info = L.WavInfo()                                   # the header walk reads the mapped bytes in place
            rc = L.lib().ss_wav_parse(data.ctypes.data_as(C.c_void_p), data.size, C.byref(info))
            if rc:
                raise AnalyzerError(rc)
            sb = L.lib().ss_pcm_sample_bytes(info.format)
            n = int(info.frames) * int(info.channels)
            raw = data[int(info.data_offset):int(info.data_offset) + n * sb]
            if raw.size != n * sb:
                raise AnalyzerError(L.SS_ERR_INVALID_ARG)
            groups.setdefault((int(info.sample_rate), int(info.channels), int(info.format)), []).append((p, raw, n))
        except (AnalyzerError, OSError) as e:
            out[p] = e
    lib = L.lib()
    for (rate, ch, fmt), items in groups.items():
        order = sorted(range(len(items)), key=lambda i: items[i][2])
        for c0 in range(0, len(items), chunk_streams):
            idx = order[c0:c0 + chunk_streams]
            frames = [items[i][2] // ch for i in idx]
            b = Batch(rate, ch, len(idx), max(max(frames), 1), fft_n, hop_frames, flags=flags)
            try:
                b.set_lengths(frames)
                for k, i in enumerate(idx):
                    if frames[k]:
                        raw = np.ascontiguousarray(items[i][1])
                        _check(lib.ss_batch_upload_samples(b._h, k, raw.ctypes.data_as(C.c_void_p), frames[k] * ch, fmt))
                b.run()
                b.sync()
                if on_chunk is not None:
                    on_chunk(b, [items[i][0] for i in idx])
                r = b.results()
                for k, i in enumerate(idx):
                    out[items[i][0]] = (r[k].integrated_lufs, r[k].loudness_range, tuple(r[k].true_peak), tuple(r[k].sample_peak))
            finally:
                b.close()
    return out
