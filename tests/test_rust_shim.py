"""rust/src/analyzer.rs (the shim a maintainer drops over the reference's src/analyzer.rs) against the C header.

There is no Rust toolchain in the image, so the shim cannot be compiled; this parses its `extern "C"` block and the
header and checks, for every declared function: it exists in include/soundscope_hip.h, has the same number of
arguments, every argument and the return type map to the same C type, and the built library exports it.  It also
checks that the shim keeps every public method signature of the reference's `Analyzer` (analyzer.rs:29-183; the
signatures are restated here because /root/reference is not available where the tests run)."""
import ctypes
import os
import re

from soundscope_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "rust", "src", "analyzer.rs")
HEADER = os.path.join(ROOT, "include", "soundscope_hip.h")

RUST_TO_C = {
    "u32": "uint32_t", "c_int": "int", "i32": "int", "usize": "size_t", "c_double": "double", "c_float": "float",
    "*mut SsAnalyzer": "ss_analyzer *", "*const SsAnalyzer": "const ss_analyzer *",
    "*mut *mut SsAnalyzer": "ss_analyzer **", "*const c_float": "const float *", "*mut c_double": "double *",
    "*mut usize": "size_t *", "*mut f64": "double *", "*const c_char": "const char *", "*mut c_float": "float *",
}


def rust_externs():
    src = open(SHIM).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', src, re.S).group(1)
    block = re.sub(r"//.*", "", block)
    out = {}
    for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([\w\s\*]+))?;", block, re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "").strip()
        types = [a.split(":", 1)[1].strip() for a in re.split(r",\s*(?![^()]*\))", args.strip()) if a.strip()]
        out[name] = (types, ret)
    return out


def header_decls():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\n\s*((?:const\s+)?[\w]+\s*\*?)\s*(ss_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                a = re.sub(r"\s*\b\w+$", "", a) if not a.endswith("*") else a      # drop the parameter name
                types.append(a.replace(" *", " *").strip())
        out[name] = (types, ret)
    return out


def norm(c):
    return re.sub(r"\s+", " ", c.replace("*", " * ")).strip()


def test_extern_block_matches_header_and_library():
    rs, hd = rust_externs(), header_decls()
    assert len(rs) >= 13
    lib = ctypes.CDLL(L.LIB_PATH)
    for name, (types, ret) in rs.items():
        assert name in hd, f"{name}: declared in the shim but not in soundscope_hip.h"
        ctypes_, cret = hd[name]
        assert len(types) == len(ctypes_), f"{name}: {len(types)} arguments in the shim, {len(ctypes_)} in the header"
        for i, (rt, ct) in enumerate(zip(types, ctypes_)):
            assert rt in RUST_TO_C, f"{name}: unmapped Rust type {rt!r}"
            assert norm(RUST_TO_C[rt]) == norm(ct), f"{name} argument {i}: {rt} vs {ct}"
        if ret:
            assert norm(RUST_TO_C[ret]) == norm(cret), f"{name}: return {ret} vs {cret}"
        else:
            assert cret == "void", f"{name}: the shim returns nothing, the header {cret}"
        assert hasattr(lib, name), f"{name}: not exported by {L.LIB_PATH}"


def test_shim_keeps_the_reference_api():
    """The twelve items of `impl Analyzer` / `impl Default` in the reference (analyzer.rs:29-183), verbatim signatures."""
    src = re.sub(r"\s+", " ", open(SHIM).read())
    for sig in [
        "pub struct Analyzer",
        "impl Default for Analyzer",
        "pub fn create_loudness_meter(&mut self, channels: u32, rate: u32) -> Result<()>",
        "pub fn get_fft(&self, samples: &[f32]) -> Result<Vec<(f64, f64)>>",
        "pub fn get_waveform(samples: &[f32], waveform_window: f64) -> Vec<(f64, f64)>",
        "pub fn add_samples(&mut self, samples: &[f32]) -> Result<(), ebur128::Error>",
        "pub fn reset(&mut self)",
        "pub fn get_shortterm_lufs(&mut self) -> Result<f64, ebur128::Error>",
        "pub fn get_integrated_lufs(&mut self) -> Result<f64, ebur128::Error>",
        "pub fn get_loudness_range(&mut self) -> Result<f64, ebur128::Error>",
        "pub fn get_true_peak(&mut self) -> Result<(f64, f64), ebur128::Error>",
        "pub fn sample_rate(&self) -> u32",
        "pub fn calculate_integrated_lufs(&mut self, channels: u32, samples: &[f32]) -> Option<f64>",
        "impl Drop for Analyzer",
    ]:
        assert sig in src, f"missing in rust/src/analyzer.rs: {sig}"
    assert "ss_analyzer_create(2, 44100" in src                    # Analyzer::default(): 2 channels, 44.1 kHz (analyzer.rs:34-45)


def test_shim_checks_the_abi_version_of_the_header():
    src = open(SHIM).read()
    m = re.search(r"const SS_ABI_VERSION: c_int = (\d+);", src)
    hv = re.search(r"#define SS_ABI_VERSION (\d+)", open(HEADER).read())
    assert m and hv and m.group(1) == hv.group(1) == str(L.SS_ABI_VERSION)
    assert "ss_abi_version()" in src and "abi != SS_ABI_VERSION" in src


def test_get_fft_errors_map_to_the_crate_variants():
    """status 10..15 of ss_get_fft become the SpectrumAnalyzerError variants the reference's `?`
    would have produced (analyzer.rs:60-65), so the text the TUI prints (tui.rs:1439-1442) is the crate's own.  The header's
    status values and the shim's match arms must agree one to one."""
    src = open(SHIM).read()
    hdr = open(HEADER).read()
    want = {"SS_ERR_TOO_FEW_SAMPLES": "TooFewSamples", "SS_ERR_NAN": "NaNValuesNotSupported",
            "SS_ERR_INFINITY": "InfinityValuesNotSupported", "SS_ERR_NOT_POW2": "SamplesLengthNotAPowerOfTwo",
            "SS_ERR_FREQ_LIMIT": "InvalidFrequencyLimit", "SS_ERR_SCALING": "ScalingError"}
    body = re.search(r"fn fft_err\(h: \*const SsAnalyzer, rc: c_int\) -> eyre::Report \{(.*?)\n\}", src, re.S).group(1)
    for name, variant in want.items():
        value = int(re.search(name + r"\s*=\s*(\d+)", hdr).group(1))
        arm = re.search(r"\b%d\s*=>\s*SpectrumAnalyzerError::(\w+)" % value, body)
        assert arm and arm.group(1) == variant, f"status {value} ({name}) must map to SpectrumAnalyzerError::{variant}"
    assert "return Err(fft_err(self.h, rc))" in src and 'eyre!("spectrum analyzer error' not in src
    # the two variants with a payload take it from the ABI (no placeholder values)
    assert "ss_get_fft_error_values(h, &mut a, &mut b)" in body and "ValueAboveNyquist(a)" in body and "ScalingError(a, b)" in body
    assert "NAN, f32::NAN).into()" not in body and "int ss_get_fft_error_values(const ss_analyzer *h, float *a, float *b);" in hdr
    ebu = re.search(r"fn ebu_err\(rc: c_int\) -> ebur128::Error \{(.*?)\n\}", src, re.S).group(1)
    for name, variant in {"SS_ERR_INVALID_MODE": "InvalidMode", "SS_ERR_INVALID_CHANNEL": "InvalidChannelIndex"}.items():
        value = int(re.search(name + r"\s*=\s*(\d+)", hdr).group(1))
        assert re.search(r"\b%d\s*=>\s*ebur128::Error::%s" % (value, variant), ebu)


def test_build_rs_links_the_library():
    b = open(os.path.join(ROOT, "rust", "build.rs")).read()
    assert "rustc-link-lib=dylib=soundscope_hip" in b and "rustc-link-search=native=" in b
