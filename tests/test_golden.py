"""Golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py).

CPU: the oracle still reproduces them (guards the checker against silent drift).
GPU: the HIP path reproduces them through the C ABI, to the north_star tolerances."""
import os

import numpy as np
import pytest

from conftest import db_close
from golden.make_golden import BATCH, CASES_FFT, CASES_METER, CASES_WAVE, golden_input

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz"))


def test_golden_input_is_stable():
    x = golden_input(1, 8)
    assert x.dtype == np.float32
    assert np.abs(golden_input(1, 100000)).max() < 0.5
    assert np.array_equal(golden_input(3, 1000)[:10], golden_input(3, 10))


def test_oracle_reproduces_golden(oracle):
    for rate, n, seed in CASES_FFT:
        assert np.array_equal(oracle.get_fft(rate, golden_input(seed, n)), G[f"fft_{rate}_{n}_{seed}"])
    for n, win, seed in CASES_WAVE:
        assert np.array_equal(oracle.get_waveform(golden_input(seed, n), win), G[f"wave_{n}_{win}_{seed}"])
    for ch, rate, secs, seed in CASES_METER:
        x = golden_input(seed, int(rate * secs) * ch, 0.6)
        m = oracle.Meter(ch, rate)
        m.add_frames(x)
        g = G[f"meter_{ch}_{rate}_{seed}"]
        assert m.integrated() == g[0] and m.loudness_range() == g[1]
        assert [m.true_peak(c) for c in range(ch)] == g[3:3 + ch].tolist()


@pytest.mark.gpu
def test_hip_reproduces_golden_single_calls():
    import soundscope_amd as ssa
    an = ssa.Analyzer()
    for rate, n, seed in CASES_FFT:
        an.create_loudness_meter(2, rate)
        got = an.get_fft(golden_input(seed, n))
        ref = G[f"fft_{rate}_{n}_{seed}"]
        assert np.array_equal(got[:, 0], ref[:, 0])
        assert db_close(got[:, 1], ref[:, 1], 0.01)
    for n, win, seed in CASES_WAVE:
        assert np.array_equal(ssa.Analyzer.get_waveform(golden_input(seed, n), win), G[f"wave_{n}_{win}_{seed}"])
    for ch, rate, secs, seed in CASES_METER:
        x = golden_input(seed, int(rate * secs) * ch, 0.6)
        an.create_loudness_meter(ch, rate)
        st = []
        step = 16384 - (16384 % ch)
        for off in range(0, x.size, step):
            an.add_samples(x[off:off + step])
            st.append(an.get_shortterm_lufs())
        g = G[f"meter_{ch}_{rate}_{seed}"]
        gst = G[f"meter_st_{ch}_{rate}_{seed}"]
        fin = np.isfinite(gst)
        assert np.array_equal(np.isfinite(st), fin)
        assert np.abs(np.array(st)[fin] - gst[fin]).max() <= 0.01
        assert abs(an.get_integrated_lufs() - g[0]) <= 0.01
        assert abs(an.get_loudness_range() - g[1]) <= 0.01
        assert abs(an.get_momentary_lufs() - g[2]) <= 0.01
        for c in range(ch):
            assert abs(an.get_true_peak_channel(c) - g[3 + c]) <= 1e-4 * g[3 + c]
            assert an.get_sample_peak_channel(c) == g[3 + ch + c]


@pytest.mark.gpu
def test_hip_reproduces_golden_batch():
    import soundscope_amd as ssa
    seeds = BATCH["seeds"]
    xs = [golden_input(s, BATCH["frames"] * 2, 0.7) for s in seeds]
    b = ssa.Batch(BATCH["rate"], 2, len(seeds), BATCH["frames"], BATCH["fft_n"], BATCH["hop"])
    b.upload(0, np.concatenate(xs))
    b.run(); b.sync()
    res = b.results()
    for i, s in enumerate(seeds):
        sc = G[f"batch_scalars_{s}"]
        w = int(sc[6])
        assert b.layout.n_windows == w and b.layout.n_bins == int(sc[7])
        fft = b.fft(i)[[0, w // 2, w - 1]]
        ref = G[f"batch_fft_{s}"]
        for k in range(3):
            for c in range(2):
                assert db_close(fft[k, c], ref[k, c], 0.01)
        assert abs(res[i].integrated_lufs - sc[0]) <= 0.01 and abs(res[i].loudness_range - sc[1]) <= 0.01
        for c in range(2):
            assert abs(res[i].true_peak[c] - sc[2 + c]) <= 1e-4 * sc[2 + c]
            assert res[i].sample_peak[c] == sc[4 + c]
        assert np.array_equal(b.waveform(i).reshape(-1), G[f"batch_wave_{s}"])
