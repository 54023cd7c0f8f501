"""The LDS layout of the batch FFT kernels is free of bank conflicts by construction: tools/lds_bank_model.py applies the
per-instruction banking rules of MI355X_MICROARCH.md (section LDS) to every access of k_fft4096_ms1's window loop, with the
row stride and the publish swizzle read from ss_fft.hip itself.  (The counters that confirmed the model on the GPU —
SQ_LDS_BANK_CONFLICT = 0 — are in profiles/r03_ab_fft_lds_layout.txt.)"""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(ROOT, "tools", "lds_bank_model.py"))
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)


def test_window_loop_of_the_spectrum_kernel_has_no_modelled_bank_conflict():
    lay = model.layout_from_source()
    # every retained-bin geometry the batch kernels see at N = 4096: 48 kHz, 96 kHz, 44.1 kHz, 32 kHz
    for first_bin, n_bins in ((2, 1705), (1, 853), (2, 1856), (3, 2046)):
        for name, cyc, ideal in model.ms1_window(lay, first_bin, n_bins):
            assert cyc == ideal, (first_bin, n_bins, name, cyc, ideal)


def test_the_model_sees_the_conflicts_of_the_previous_layout():
    # rows of 18 complex: the first exchange's ds_write_b64 groups land two deep (4 tb mod 32)
    old = {"row": 18, "plane": 288, "spec": (lambda k: k ^ (((k >> 6) & 1) << 1))}
    rows = dict((name, (cyc, ideal)) for name, cyc, ideal in model.ms1_window(old))
    cyc, ideal = rows["exchange 1 write (ka; tb, hi)"]
    assert cyc == 2 * ideal
    cyc, ideal = rows["epilogue bins and mirrors"]
    assert cyc == 2 * ideal
