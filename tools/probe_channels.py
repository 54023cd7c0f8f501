import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
for ch, ns in ((1, 2048), (2, 1024), (3, 683), (4, 512), (6, 340), (8, 256), (16, 128)):
    b = ssa.Batch(48000, ch, ns, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(3, 0)
    b.run(); b.sync()
    b.timing_enable(True)
    for _ in range(3):
        b.run(); b.sync()
    ms=[b.timing_read(k)[0]/3 for k in range(4)]
    print(f"channels {ch} x {ns} streams: {L.lib().ss_batch_kernel_name(b._h,0).decode()} {ms[0]:.3f} ms, time domain {ms[1]:.3f} ms, waveform {ms[3]:.3f} -> {ns*480000*ch/sum(ms)/1e6:.1f} Gsamples/s")
    b.close()
