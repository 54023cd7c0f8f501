import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import soundscope_amd as ssa
from conftest import make_stereo, make_multich
from oracle import pyoracle as po
for rate, channels in [(48000,2),(44100,2),(96000,2),(192000,2),(48000,6)]:
    frames = rate*4+123
    x = make_multich(7 + channels, frames, channels, rate) if channels != 2 else make_stereo(7, frames, rate, level=0.7, gap=True)
    a,b = ssa.Analyzer(), ssa.Analyzer(); a.create_loudness_meter(channels, rate); b.create_loudness_meter(channels, rate)
    m = po.Meter(channels, rate)
    wd=0; wa=0; wb=0; wr=0
    for off in range(0, frames, 8192):
        sl = x[off*channels:(off+8192)*channels]
        a.add_samples(sl); m.add_frames(sl)
        for o2 in range(off, min(off+8192, frames), 256):
            b.add_samples(x[o2*channels:min(o2+256, off+8192, frames)*channels])
        for c in range(channels):
            ga, gb, go = a.filter_state(c), b.filter_state(c), m.filter_state(c)
            sc = max(np.abs(go).max(), 1e-300)
            wd = max(wd, np.abs(ga-gb).max()/sc); wa = max(wa, np.abs(ga-go).max()/sc); wb = max(wb, np.abs(gb-go).max()/sc)
        for name in ("get_shortterm_lufs", "get_momentary_lufs"):
            va, vb = getattr(a,name)(), getattr(b,name)()
            if np.isfinite(va) and np.isfinite(vb): wr = max(wr, abs(va-vb))
    print(rate, channels, "state: split-vs-one %.2e  split-vs-oracle %.2e  one-vs-oracle %.2e   readings split-vs-one %.2e LU" % (wd, wa, wb, wr))
