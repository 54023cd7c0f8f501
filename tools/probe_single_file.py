import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from oracle import pyoracle as po
for secs in (10, 60):
    b = ssa.Batch(48000, 2, 1, 48000 * secs, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(0x5EED0000, 0)
    g = b.geometry
    for ov in (0, 1):
        b.set_overlap(ov)
        for _ in range(3): b.run(); b.sync()
        t0 = time.perf_counter()
        for _ in range(50): b.run(); b.sync()
        wall = (time.perf_counter() - t0) / 50 * 1e3
        print(f"{secs} s file, overlap {ov}: wall {wall:.4f} ms; geometry split {g.td_split} segments {g.td_segments} x {g.td_segment_subblocks} fixup {g.td_fixup_subblocks} fft wpb {g.fft_windows_per_block}")
    b.set_overlap(0)
    b.timing_enable(True)
    for _ in range(10): b.run(); b.sync()
    print("   kernels:", {L.lib().ss_batch_kernel_name(b._h, k).decode(): round(b.timing_read(k)[0] / max(b.timing_read(k)[1], 1), 4) for k in range(L.SS_KERNEL_COUNT)})
    x = b.download_input(0)
    ref = po.analyze_stream(48000, x, 4096, 1024)
    r = b.results()[0]
    print("   vs oracle: I", r.integrated_lufs - ref["integrated"], "LRA", r.loudness_range - ref["lra"], "TP", r.true_peak[0] - ref["true_peak"][0], "wave", np.array_equal(b.waveform(0).reshape(-1), ref["wave"][:, 1].astype(np.float32)))
