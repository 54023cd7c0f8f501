//! soundscope Analyzer on libsoundscope_hip.so (MI355X).  Same public API as upstream analyzer.rs.
use eyre::{eyre, Result};
use spectrum_analyzer::error::SpectrumAnalyzerError;
use spectrum_analyzer::FrequencyLimitError;
use std::ffi::CStr;
use std::os::raw::{c_char, c_double, c_float, c_int};

#[repr(C)]
pub struct SsAnalyzer { _private: [u8; 0] }

#[link(name = "soundscope_hip")]
extern "C" {
    fn ss_analyzer_create(channels: u32, rate: u32, out: *mut *mut SsAnalyzer) -> c_int;
    fn ss_analyzer_destroy(h: *mut SsAnalyzer);
    fn ss_analyzer_configure(h: *mut SsAnalyzer, channels: u32, rate: u32) -> c_int;
    fn ss_get_fft(h: *const SsAnalyzer, x: *const c_float, n: usize,
                  out_xy: *mut c_double, cap_pairs: usize, out_n: *mut usize) -> c_int;
    fn ss_get_fft_error_values(h: *const SsAnalyzer, a: *mut c_float, b: *mut c_float) -> c_int;
    fn ss_get_waveform(x: *const c_float, n: usize, window: c_double,
                       out_xy: *mut c_double, cap_pairs: usize, out_n: *mut usize) -> c_int;
    fn ss_add_samples(h: *mut SsAnalyzer, x: *const c_float, n: usize) -> c_int;
    fn ss_reset(h: *mut SsAnalyzer);
    fn ss_get_shortterm_lufs(h: *mut SsAnalyzer, out: *mut c_double) -> c_int;
    fn ss_get_integrated_lufs(h: *mut SsAnalyzer, out: *mut c_double) -> c_int;
    fn ss_get_loudness_range(h: *mut SsAnalyzer, out: *mut c_double) -> c_int;
    fn ss_get_true_peak(h: *mut SsAnalyzer, l: *mut c_double, r: *mut c_double) -> c_int;
    fn ss_sample_rate(h: *const SsAnalyzer) -> u32;
    fn ss_calculate_integrated_lufs(h: *mut SsAnalyzer, channels: u32, x: *const c_float,
                                    n: usize, out: *mut c_double) -> c_int;
    // extensions the reference's ebur128 meter has under Mode::all() but the app does not surface (tui.rs:1217-1221)
    fn ss_get_momentary_lufs(h: *mut SsAnalyzer, out: *mut c_double) -> c_int;
    fn ss_get_true_peak_channel(h: *mut SsAnalyzer, channel: u32, out: *mut c_double) -> c_int;
    fn ss_get_sample_peak_channel(h: *mut SsAnalyzer, channel: u32, out: *mut c_double) -> c_int;
    fn ss_analyzer_set_true_peak_factor(h: *mut SsAnalyzer, factor: c_int) -> c_int;
    fn ss_status_string(status: c_int) -> *const c_char;
    fn ss_abi_version() -> c_int;
}

/// `SS_ABI_VERSION` of include/soundscope_hip.h this shim was written against (checked once, in `Default`).
const SS_ABI_VERSION: c_int = 2;

fn ebu_err(rc: c_int) -> ebur128::Error {
    match rc { 2 => ebur128::Error::InvalidMode, 3 => ebur128::Error::InvalidChannelIndex, _ => ebur128::Error::NoMem }
}

/// get_fft's `?` in upstream analyzer.rs:60-65 turns a `SpectrumAnalyzerError` into the eyre report whose text the TUI
/// prints (tui.rs:1439-1442): the same variants come back here WITH their payloads (`ss_get_fft_error_values`: the limit
/// that exceeds Nyquist; the original and the scaled value of the first bin the scaling function spoiled), so the message
/// a user sees does not change.
fn fft_err(h: *const SsAnalyzer, rc: c_int) -> eyre::Report {
    let (mut a, mut b) = (f32::NAN, f32::NAN);
    if rc == 14 || rc == 15 { unsafe { ss_get_fft_error_values(h, &mut a, &mut b) }; }
    match rc {
        10 => SpectrumAnalyzerError::TooFewSamples.into(),
        11 => SpectrumAnalyzerError::NaNValuesNotSupported.into(),
        12 => SpectrumAnalyzerError::InfinityValuesNotSupported.into(),
        13 => SpectrumAnalyzerError::SamplesLengthNotAPowerOfTwo.into(),
        14 => SpectrumAnalyzerError::InvalidFrequencyLimit(FrequencyLimitError::ValueAboveNyquist(a)).into(),
        15 => SpectrumAnalyzerError::ScalingError(a, b).into(),
        _ => eyre!("soundscope_hip: {}", unsafe { CStr::from_ptr(ss_status_string(rc)) }.to_string_lossy()),
    }
}

pub struct Analyzer { h: *mut SsAnalyzer }
unsafe impl Send for Analyzer {}

impl Default for Analyzer {
    fn default() -> Self {
        let abi = unsafe { ss_abi_version() };
        if abi != SS_ABI_VERSION { panic!("libsoundscope_hip.so has ABI version {abi}, this shim needs {SS_ABI_VERSION}"); }
        let mut h = std::ptr::null_mut();
        let rc = unsafe { ss_analyzer_create(2, 44100, &mut h) };
        if rc != 0 { panic!("Failed to create loudness meter: status {rc}"); }
        Self { h }
    }
}
impl Drop for Analyzer { fn drop(&mut self) { unsafe { ss_analyzer_destroy(self.h) } } }

impl Analyzer {
    pub fn create_loudness_meter(&mut self, channels: u32, rate: u32) -> Result<()> {
        match unsafe { ss_analyzer_configure(self.h, channels, rate) } { 0 => Ok(()), rc => Err(ebu_err(rc).into()) }
    }
    pub fn get_fft(&self, samples: &[f32]) -> Result<Vec<(f64, f64)>> {
        let cap = samples.len() / 2 + 1;
        let mut out = vec![(0f64, 0f64); cap];          // (f64,f64) is two consecutive doubles
        let mut n = 0usize;
        let rc = unsafe { ss_get_fft(self.h, samples.as_ptr(), samples.len(),
                                     out.as_mut_ptr() as *mut f64, cap, &mut n) };
        if rc != 0 { return Err(fft_err(self.h, rc)); }
        out.truncate(n);
        Ok(out)
    }
    pub fn get_waveform(samples: &[f32], waveform_window: f64) -> Vec<(f64, f64)> {
        let cap = 2 * (waveform_window * 1000.) as usize + 2;
        let mut out = vec![(0f64, 0f64); cap];
        let mut n = 0usize;
        let rc = unsafe { ss_get_waveform(samples.as_ptr(), samples.len(), waveform_window,
                                          out.as_mut_ptr() as *mut f64, cap, &mut n) };
        out.truncate(if rc == 0 { n } else { 0 });
        out
    }
    pub fn add_samples(&mut self, samples: &[f32]) -> Result<(), ebur128::Error> {
        match unsafe { ss_add_samples(self.h, samples.as_ptr(), samples.len()) } { 0 => Ok(()), rc => Err(ebu_err(rc)) }
    }
    pub fn reset(&mut self) { unsafe { ss_reset(self.h) } }
    pub fn get_shortterm_lufs(&mut self) -> Result<f64, ebur128::Error> { self.scalar(ss_get_shortterm_lufs) }
    pub fn get_integrated_lufs(&mut self) -> Result<f64, ebur128::Error> { self.scalar(ss_get_integrated_lufs) }
    pub fn get_loudness_range(&mut self) -> Result<f64, ebur128::Error> { self.scalar(ss_get_loudness_range) }
    fn scalar(&mut self, f: unsafe extern "C" fn(*mut SsAnalyzer, *mut f64) -> c_int) -> Result<f64, ebur128::Error> {
        let mut v = 0f64;
        match unsafe { f(self.h, &mut v) } { 0 => Ok(v), rc => Err(ebu_err(rc)) }
    }
    pub fn get_true_peak(&mut self) -> Result<(f64, f64), ebur128::Error> {
        let (mut l, mut r) = (0f64, 0f64);
        match unsafe { ss_get_true_peak(self.h, &mut l, &mut r) } { 0 => Ok((l, r)), rc => Err(ebu_err(rc)) }
    }
    pub fn sample_rate(&self) -> u32 { unsafe { ss_sample_rate(self.h) } }
    /// Extensions (not in upstream analyzer.rs): momentary loudness, any channel's peaks, forced oversampling.
    pub fn get_momentary_lufs(&mut self) -> Result<f64, ebur128::Error> { self.scalar(ss_get_momentary_lufs) }
    pub fn get_true_peak_channel(&mut self, channel: u32) -> Result<f64, ebur128::Error> {
        let mut v = 0f64;
        match unsafe { ss_get_true_peak_channel(self.h, channel, &mut v) } { 0 => Ok(v), rc => Err(ebu_err(rc)) }
    }
    pub fn get_sample_peak_channel(&mut self, channel: u32) -> Result<f64, ebur128::Error> {
        let mut v = 0f64;
        match unsafe { ss_get_sample_peak_channel(self.h, channel, &mut v) } { 0 => Ok(v), rc => Err(ebu_err(rc)) }
    }
    pub fn set_true_peak_factor(&mut self, factor: i32) -> bool { unsafe { ss_analyzer_set_true_peak_factor(self.h, factor) == 0 } }
    pub fn calculate_integrated_lufs(&mut self, channels: u32, samples: &[f32]) -> Option<f64> {
        let mut v = 0f64;
        (unsafe { ss_calculate_integrated_lufs(self.h, channels, samples.as_ptr(), samples.len(), &mut v) } == 0).then_some(v)
    }
}
