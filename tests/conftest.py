import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_stereo(seed, frames, rate=48000, level=0.5, gap=False):
    """Seeded synthetic interleaved stereo f32: two sines + uniform noise (SURVEY §8d shape)."""
    rng = np.random.default_rng(seed)
    t = np.arange(frames, dtype=np.float64) / rate
    f1, f2 = np.exp(rng.uniform(np.log(50), np.log(12000), 2))
    ph = rng.uniform(0, 2 * np.pi)
    l = 0.25 * np.sin(2 * np.pi * f1 * t) + 0.05 * rng.uniform(-1, 1, frames)
    r = 0.25 * np.sin(2 * np.pi * f2 * t + ph) + 0.05 * rng.uniform(-1, 1, frames)
    x = np.empty(2 * frames, np.float32)
    x[0::2] = (level * l).astype(np.float32)
    x[1::2] = (level * r).astype(np.float32)
    if gap:
        g0 = frames // 3
        x[2 * g0:2 * (g0 + 3 * rate)] *= np.float32(1e-4)
    return x


def make_multich(seed, frames, channels, rate=48000, level=0.4):
    rng = np.random.default_rng(seed)
    t = np.arange(frames, dtype=np.float64) / rate
    x = np.empty((frames, channels), np.float32)
    for c in range(channels):
        f = np.exp(rng.uniform(np.log(60), np.log(9000)))
        x[:, c] = (level * (0.3 * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28)) + 0.05 * rng.uniform(-1, 1, frames))).astype(np.float32)
    return x.reshape(-1)


def db_close(got, ref, tol_db=0.01, floor_db=-90.0, ref_floor=None):
    """Spectrum parity metric (SURVEY §7 hard part 3): |d| <= tol_db where the reference bin is
    within `floor` of the window's loudest bin region (>= floor_db absolute); below that the
    comparison is absolute-linear: the error must stay under 1e-4 of the loudest bin's amplitude."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    strong = ref >= floor_db
    ok_strong = np.abs(got[strong] - ref[strong]) <= tol_db
    peak = ref.max()
    lin_err = np.abs(10 ** (got[~strong] / 20) - 10 ** (ref[~strong] / 20))
    ok_weak = lin_err <= 1e-4 * 10 ** (peak / 20) + 1e-12
    return bool(ok_strong.all() and ok_weak.all())


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle
