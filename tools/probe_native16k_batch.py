import os, sys
sys.path.insert(0, os.getcwd())
import soundscope_amd as ssa
from soundscope_amd import _lib as L
b = ssa.Batch(48000, 2, 1024, 480000, 16384, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(3, 0)
b.run(); b.sync()
b.timing_enable(True)
for _ in range(3): b.run()
ms=[b.timing_read(k) for k in range(4)]
lay=b.layout
print("windows", lay.n_windows, "bins", lay.n_bins, [ (L.lib().ss_batch_kernel_name(b._h,k).decode(), round(m/n,3)) for k,(m,n) in enumerate(ms) if n])
tot=sum(m/n for m,n in ms if n)
print(f"{1024*480000*2/tot/1e6:.1f} G samples/s; spectrum out {lay.n_windows*2*lay.n_bins*4*1024/1e9:.1f} GB -> {lay.n_windows*2*lay.n_bins*4*1024/(ms[0][0]/ms[0][1])/1e6:.0f} GB/s of stores")
