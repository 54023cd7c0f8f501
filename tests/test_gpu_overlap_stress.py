"""Launch modes must not change a single bit: the two-stream passes of ss_batch_set_overlap (modes 1 and 2) against the
sequential pass, over MANY repetitions, with EVERY stream compared — device-side checksums of each stream's whole
spectrum block, decimation bins and sub-block energies (ss_batch_checksums), and every field of every stream's results.

Round 2 saw isolated wrong spectrum windows with an experimental K = 32 MFMA form of the true peak running beside the
spectrum kernel; that variant is gone, and this is the guard that a recurrence — with any kernel —
cannot pass unnoticed: a one-pass check over eight streams would miss a corruption that hits a few windows per pass.
"""
import numpy as np
import pytest

import soundscope_amd as ssa
from soundscope_amd import _lib as L

pytestmark = pytest.mark.gpu


def _results_table(b):
    res = b.results()
    return np.array([(r.integrated_lufs, r.loudness_range, r.true_peak[0], r.true_peak[1], r.sample_peak[0],
                      r.sample_peak[1], float(r.n_gating_blocks), float(r.n_st_blocks)) for r in res], np.float64)


def _same(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))        # bit patterns: -inf and NaN compare like any value


def _stress(b, passes):
    b.set_overlap(0)
    b.run(); b.sync()
    ref_c, ref_r, ref_h = b.checksums(), _results_table(b), np.concatenate(b.histograms())
    assert ref_c[:, 0].all(), "spectrum checksums must not be zero"
    b.run(); b.sync()                                                   # the sequential pass itself repeats bit for bit
    assert np.array_equal(b.checksums(), ref_c) and _same(_results_table(b), ref_r)
    bad = []
    for rep in range(passes):
        for mode in (1, 2):
            b.set_overlap(mode)
            b.run(); b.sync()
            c = b.checksums()
            if not np.array_equal(c, ref_c):
                rows, cols = np.nonzero(c != ref_c)
                bad.append((rep, mode, "checksum", sorted(set(zip(rows.tolist(), cols.tolist())))[:8]))
            if not _same(_results_table(b), ref_r):
                bad.append((rep, mode, "results", None))
            if not np.array_equal(np.concatenate(b.histograms()), ref_h):
                bad.append((rep, mode, "corpus histograms", None))
    assert not bad, f"{len(bad)} of {2 * passes} overlapped passes differ from the sequential pass: {bad[:6]}"


def test_overlap_modes_bit_identical_config3_x50():
    """BASELINE config 3 at the bench shape (1024 x 10 s x 48 kHz stereo, N = 4096): 50 passes each of modes 1 and 2."""
    b = ssa.Batch(48000, 2, 1024, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(0x5EED0000, 0)
    _stress(b, 50)
    b.close()


def test_overlap_modes_bit_identical_config5_x20():
    """BASELINE config 5 (64 x 10 s x 96 kHz x 8 ch, N = 16384, forced 4x true peak): 20 passes each of modes 1 and 2."""
    b = ssa.Batch(96000, 8, 64, 960000, 16384, 1024, flags=L.SS_BATCH_ALL, true_peak_factor=4)
    b.synthesize(7, 0)
    _stress(b, 20)
    b.close()


def test_checksums_see_a_single_changed_sample():
    """The checksum is a detector, so it is checked as one: flipping one input sample of one stream changes that stream's
    spectrum and sub-block checksums and nobody else's."""
    b = ssa.Batch(48000, 2, 8, 96000, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(3, 0)
    b.run(); b.sync()
    c0 = b.checksums()
    x = b.download_input(5)
    x[40001] = np.float32(x[40001] + 1e-3)
    b.upload(5, x)
    b.run(); b.sync()
    c1 = b.checksums()
    changed = np.nonzero((c0 != c1).any(axis=1))[0].tolist()
    assert changed == [5], changed
    assert c0[5, 0] != c1[5, 0] and c0[5, 2] != c1[5, 2]
    b.close()
