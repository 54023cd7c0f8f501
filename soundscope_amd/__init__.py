"""soundscope_amd — MI355X-native implementation of soundscope's analyzer hot path.

The product is the C-ABI library `soundscope_amd/lib/libsoundscope_hip.so`
(hand-written HIP kernels for gfx950, see include/soundscope_hip.h); this package
is the thin host-side mirror of the reference's `Analyzer` interface used by the
tests and the benchmark.  It has no CPU compute path.
"""
from . import _lib
from ._lib import build
from .analyzer import Analyzer, AnalyzerError, DeviceError, get_mid_and_side_samples
from .batch import Batch, corpus_integrated_lufs, corpus_loudness_range
from .session import CaptureSession, FileSession
from .pipeline import analyze_corpus, analyze_streams, analyze_wav_files

__all__ = ["Analyzer", "AnalyzerError", "DeviceError", "Batch", "build", "get_mid_and_side_samples",
           "corpus_integrated_lufs", "corpus_loudness_range", "FileSession", "CaptureSession", "analyze_corpus", "analyze_streams", "analyze_wav_files", "_lib"]
