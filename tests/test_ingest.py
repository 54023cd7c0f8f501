"""PCM ingest: RIFF/WAVE parsing is host logic (no GPU); conversion runs on the device
and is bit-exact against the oracle's restatement of symphonia's sample conversions."""
import io
import struct
import wave

import numpy as np
import pytest

import soundscope_amd as ssa
from soundscope_amd import _lib as L
from soundscope_amd import ingest


def make_wav(samples_i, channels, rate, width):
    """PCM WAV via the stdlib writer (width bytes per sample: 1 = u8, 2, 3, 4)."""
    bio = io.BytesIO()
    w = wave.open(bio, "wb")
    w.setnchannels(channels); w.setsampwidth(width); w.setframerate(rate)
    if width == 1:
        raw = (np.asarray(samples_i) + 128).astype(np.uint8).tobytes()
    elif width == 3:
        a = np.asarray(samples_i, np.int32)
        raw = b"".join(struct.pack("<i", int(v))[:3] for v in a)
    else:
        raw = np.asarray(samples_i).astype({2: "<i2", 4: "<i4"}[width]).tobytes()
    w.writeframes(raw); w.close()
    return bio.getvalue(), raw


def make_float_wav(x, channels, rate, extensible=False, f64=False, junk=True):
    data = np.asarray(x, np.float64 if f64 else np.float32).tobytes()
    bits = 64 if f64 else 32
    block = channels * bits // 8
    if extensible:
        guid = struct.pack("<H", 3) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, channels, rate, rate * block, block, bits, 22, bits, 3) + guid
    else:
        fmt = struct.pack("<HHIIHH", 3, channels, rate, rate * block, block, bits)
    chunks = b""
    if junk:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"      # odd-sized chunk is padded to even
    chunks += b"fmt " + struct.pack("<I", len(fmt)) + fmt
    chunks += b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks, data


def test_wav_parse_pcm_widths():
    rng = np.random.default_rng(0)
    for width, fmt, bits in [(1, L.SS_PCM_U8, 8), (2, L.SS_PCM_S16, 16), (3, L.SS_PCM_S24, 24), (4, L.SS_PCM_S32, 32)]:
        lim = 2 ** (8 * width - 1)
        s = rng.integers(-lim, lim, 2 * 1000)
        data, raw = make_wav(s, 2, 44100, width)
        info = ingest.wav_parse(data)
        assert (info.format, info.channels, info.sample_rate, info.bits_per_sample) == (fmt, 2, 44100, bits)
        assert info.frames == 1000 and info.data_bytes == len(raw)
        assert data[info.data_offset:info.data_offset + info.data_bytes] == raw


def test_wav_parse_float_extensible_and_odd_chunks():
    x = np.linspace(-1, 1, 3 * 777).astype(np.float32)
    for ext in (False, True):
        data, raw = make_float_wav(x, 3, 96000, extensible=ext)
        info = ingest.wav_parse(data)
        assert (info.format, info.channels, info.sample_rate, info.frames) == (L.SS_PCM_F32, 3, 96000, 777)
        assert data[info.data_offset:info.data_offset + info.data_bytes] == raw
    data, _ = make_float_wav(x.astype(np.float64), 1, 48000, f64=True)
    assert ingest.wav_parse(data).format == L.SS_PCM_F64


def test_wav_parse_rejects_garbage_and_truncation():
    with pytest.raises(ssa.AnalyzerError) as e:
        ingest.wav_parse(b"not a wave file at all")
    assert e.value.code == L.SS_ERR_INVALID_ARG
    data, _ = make_wav(np.zeros(100, np.int16), 1, 8000, 2)
    with pytest.raises(ssa.AnalyzerError):
        ingest.wav_parse(data[:30])                           # header cut inside fmt
    info = ingest.wav_parse(data[:-50])                        # data shorter than its declared size
    assert info.frames == 75
    bad = bytearray(data); bad[20:22] = struct.pack("<H", 0x55)   # MP3-in-WAV
    with pytest.raises(ssa.AnalyzerError) as e:
        ingest.wav_parse(bytes(bad))
    assert e.value.code == L.SS_ERR_UNSUPPORTED


def test_wav_parse_mutated_headers_never_point_outside_the_file():
    """Randomised: valid headers of every supported and a few unsupported formats with a handful of bytes flipped and / or the file cut
    short — the parser either refuses or returns a data range that lies inside the buffer it was given (the copy is exact-sized: an
    over-read would leave the allocation)."""
    import ctypes as C
    rng = np.random.default_rng(20250928)
    lib = L.lib()
    accepted = 0
    for _ in range(4000):
        ch, bits = int(rng.integers(0, 9)), int(rng.choice([8, 16, 24, 32, 64, 12]))
        tag, rate = int(rng.choice([1, 3, 7])), int(rng.choice([0, 8000, 44100, 48000, 2 ** 31]))
        fb = bits // 8 * ch
        ext = rng.random() < 0.3
        fmt = struct.pack("<HHIIHH", 0xFFFE if ext else tag, ch, rate, (rate * fb) & 0xFFFFFFFF, fb & 0xFFFF, bits)
        if ext:
            fmt += struct.pack("<HHI", 22, bits, 3) + struct.pack("<H", tag) + b"\0" * 14
        data = bytes(rng.integers(0, 256, int(rng.integers(0, 40)) * fb, dtype=np.uint8))
        junk = b"LIST" + struct.pack("<I", 3) + b"abc\0" if rng.random() < 0.3 else b""
        body = b"WAVE" + junk + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data
        b = bytearray(b"RIFF" + struct.pack("<I", len(body)) + body)
        for _ in range(int(rng.integers(0, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        if rng.random() < 0.3:
            b = b[:int(rng.integers(0, len(b) + 1))]
        buf = (C.c_ubyte * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")
        info = L.WavInfo()
        if lib.ss_wav_parse(buf, len(b), C.byref(info)) == L.SS_OK:
            accepted += 1
            assert info.data_offset + info.data_bytes <= len(b)
            assert info.channels > 0 and info.format > 0
            assert info.frames * lib.ss_pcm_sample_bytes(info.format) * info.channels <= info.data_bytes
    assert accepted > 200                                     # (the mutations leave plenty of valid files)


def test_oracle_pcm_conversion_values(oracle):
    assert oracle.pcm_to_f32(bytes([0, 128, 255]), 1).tolist() == [-1.0, 0.0, 127 / 128]
    assert oracle.pcm_to_f32(struct.pack("<hhh", -32768, 0, 32767), 2).tolist() == [-1.0, 0.0, 32767 / 32768]
    assert oracle.pcm_to_f32(bytes([0, 0, 0x80, 0xFF, 0xFF, 0x7F]), 3).tolist() == [-1.0, 8388607 / 8388608]
    v = oracle.pcm_to_f32(struct.pack("<ii", -2 ** 31, 2 ** 31 - 1), 4)
    assert v[0] == -1.0 and v[1] == np.float32(1.0)            # (2^31-1)/2^31 rounds to 1.0 in f32


@pytest.mark.gpu
def test_pcm_decode_bit_exact(oracle):
    rng = np.random.default_rng(5)
    n = 100003
    cases = {
        L.SS_PCM_U8: rng.integers(0, 256, n).astype(np.uint8).tobytes(),
        L.SS_PCM_S16: rng.integers(-32768, 32768, n).astype("<i2").tobytes(),
        L.SS_PCM_S24: rng.integers(0, 256, 3 * n).astype(np.uint8).tobytes(),
        L.SS_PCM_S32: rng.integers(-2 ** 31, 2 ** 31, n).astype("<i4").tobytes(),
        L.SS_PCM_F32: rng.standard_normal(n).astype("<f4").tobytes(),
        L.SS_PCM_F64: rng.standard_normal(n).astype("<f8").tobytes(),
    }
    for fmt, raw in cases.items():
        got = ingest.pcm_decode(raw, fmt)
        assert np.array_equal(got, oracle.pcm_to_f32(raw, fmt)), fmt


@pytest.mark.gpu
def test_wav_file_end_to_end(oracle):
    """BASELINE config 1 shape: a 16-bit stereo WAV through ingest + the whole hot path."""
    import ctypes as C
    from conftest import make_stereo
    rate, frames = 48000, 48000 * 3
    x = make_stereo(3, frames, rate)
    s16 = np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)
    data, raw = make_wav(s16, 2, rate, 2)
    samples, sr, ch = ingest.decode_wav(data)
    ref = oracle.pcm_to_f32(raw, L.SS_PCM_S16)
    assert sr == rate and ch == 2 and np.array_equal(samples, ref)
    b = ssa.Batch(rate, 2, 1, frames, 4096, 1024)
    buf = (C.c_ubyte * len(raw)).from_buffer_copy(raw)
    assert L.lib().ss_batch_upload_pcm(b._h, 0, 1, buf, L.SS_PCM_S16) == 0
    assert np.array_equal(b.download_input(0), ref)
    b.run(); b.sync()
    r = oracle.analyze_stream(rate, ref, 4096, 1024)
    assert abs(b.results()[0].integrated_lufs - r["integrated"]) <= 0.01
    assert np.array_equal(b.waveform(0).reshape(-1), r["wave"][:, 1].astype(np.float32))


@pytest.mark.gpu
def test_analyze_wav_files_end_to_end(oracle, tmp_path):
    """Files of four sample formats, two rates and different lengths, a mono one and a broken one: raw PCM goes up
    as is, results equal the oracle meter on the samples symphonia would have decoded."""
    from conftest import make_stereo
    specs = [("a16.wav", 48000, 2, 2, 3.0), ("b24.wav", 48000, 2, 3, 1.5), ("c16.wav", 48000, 2, 2, 0.7),
             ("d32f.wav", 44100, 2, "f32", 2.0), ("e8.wav", 44100, 2, 1, 1.0), ("mono16.wav", 48000, 1, 2, 2.0)]
    expect = {}
    paths = []
    for name, rate, ch, width, secs in specs:
        frames = int(rate * secs)
        x = make_stereo(len(name) + frames, frames, rate, level=0.4)[:frames * ch] if ch == 2 else \
            make_stereo(5, frames, rate, level=0.4)[0::2].copy()
        if width == "f32":
            data, _ = make_float_wav(x, ch, rate)
            ref = x.astype(np.float32)
        else:
            full = {1: 127, 2: 32767, 3: 8388607}[width]
            xi = np.round(x * full).astype(np.int32)
            data, _ = make_wav(xi, ch, rate, width)
            ref = (xi.astype(np.float64) / {1: 128.0, 2: 32768.0, 3: 8388608.0}[width]).astype(np.float32)
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(str(p))
        m = oracle.Meter(ch, rate)
        m.add_frames(ref)
        expect[str(p)] = m
    bad = tmp_path / "broken.wav"
    bad.write_bytes(b"RIFF\x00\x00\x00\x00WAVEjunk")
    out = ssa.analyze_wav_files(paths + [str(bad)], chunk_streams=2)
    assert isinstance(out[str(bad)], ssa.AnalyzerError)
    for p, m in expect.items():
        integ, lra, tp, sp = out[p]
        assert (integ == m.integrated()) or abs(integ - m.integrated()) <= 0.01, p
        assert abs(lra - m.loudness_range()) <= 0.01
        assert abs(tp[0] - m.true_peak(0)) <= 1e-4 * m.true_peak(0)
        assert sp[0] == m.sample_peak(0)
