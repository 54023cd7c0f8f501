import sys,os
sys.path.insert(0,os.getcwd())
import soundscope_amd as ssa
from soundscope_amd import _lib as L
for args in ((48000,2,1024,480000,4096,1024),(96000,8,64,960000,16384,1024),(48000,2,1,480000,4096,1024),(48000,2,300,480000,4096,1024)):
    b=ssa.Batch(*args); g=b.geometry
    print(args, "segments", g.td_segments, "x", g.td_segment_subblocks, "split", g.td_split, "fixup", g.td_fixup_subblocks); b.close()
