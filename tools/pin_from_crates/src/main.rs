//! Golden vectors from the crates that hold the reference's arithmetic (ebur128 0.1.10, spectrum-analyzer 1.7.0 + microfft 0.6.0),
//! on the seeded inputs of tests/golden/make_golden.py.  Output: one little-endian .npy file per array in the directory given as
//! the first argument; tools/pin_from_crates/pack.py turns them into tests/golden/crates_v1.npz.
//!
//! With the default feature the reference's own `Analyzer` is compiled from /root/reference/src/analyzer.rs (through #[path]:
//! the file is used where it lies and is not part of this repository); with `--no-default-features --features crates-only`
//! the three formulas of analyzer.rs:11-27, :75-102, :107-137 are this harness's own restatement over the same crate calls.
use std::fs::File;
use std::io::Write;
use std::path::{Path, PathBuf};

#[cfg(feature = "reference-source")]
#[allow(dead_code)]
#[path = "../../../../reference/src/analyzer.rs"]
mod analyzer;

#[cfg(not(feature = "reference-source"))]
mod analyzer {
    //! Restatement used only when the reference tree is absent (same public surface as the reference's `Analyzer`).
    use ebur128::{EbuR128, Mode};
    use spectrum_analyzer::{samples_fft_to_spectrum, scaling::SpectrumDataStats, windows::hann_window, FrequencyLimit};

    fn to_dbfs(val: f32, stats: &SpectrumDataStats) -> f32 {
        if val == 0.0 { -150.0 } else { 20.0 * (val * 4.0 / stats.n).log10() }
    }
    pub struct Analyzer { meter: EbuR128, rate: u32 }
    impl Default for Analyzer {
        fn default() -> Self { Self { meter: EbuR128::new(2, 44100, Mode::all()).unwrap(), rate: 44100 } }
    }
    impl Analyzer {
        pub fn create_loudness_meter(&mut self, channels: u32, rate: u32) -> eyre::Result<()> {
            self.rate = rate;
            self.meter = EbuR128::new(channels, rate, Mode::all())?;
            Ok(())
        }
        pub fn get_fft(&self, samples: &[f32]) -> eyre::Result<Vec<(f64, f64)>> {
            let windowed = hann_window(samples);
            let spectrum = samples_fft_to_spectrum(&windowed, self.rate, FrequencyLimit::Range(20., 20_000.), Some(&to_dbfs))?;
            // analyzer.rs:75-102: every bin gains 10 log10(f / 1 kHz) (pink noise reads flat), x = position of log10 f between
            // log10 20 and log10 20000, scaled to 0..100 — all in f64 on the crate's f32 (frequency, value) pairs
            let (lo, hi) = (20f64.log10(), 20_000f64.log10());
            Ok(spectrum.data().iter().map(|(f, v)| {
                let f = f.val() as f64;
                ((f.log10() - lo) / (hi - lo) * 100.0, v.val() as f64 + 10.0 * (f / 1000.0).log10())
            }).collect())
        }
        pub fn get_waveform(samples: &[f32], window: f64) -> Vec<(f64, f64)> {
            let w = (window * 1000.) as usize;
            let spp = samples.len() as f64 / w as f64;
            let mut out = Vec::with_capacity(2 * w);
            for i in 0..w {
                let (a, b) = ((i as f64 * spp) as usize, (((i + 1) as f64 * spp).ceil() as usize).min(samples.len()));
                if a >= samples.len() { break; }
                let s = &samples[a..b];
                let (mut mn, mut mx) = (if s.is_empty() { 0.0 } else { f32::NAN }, if s.is_empty() { 0.0 } else { f32::NAN });
                for &v in s { mn = f32::min(mn, v); mx = f32::max(mx, v); }      // f32::min / max ignore a NaN operand
                out.push((i as f64, mn as f64));
                out.push((i as f64, mx as f64));
            }
            out
        }
        pub fn add_samples(&mut self, s: &[f32]) -> Result<(), ebur128::Error> { self.meter.add_frames_f32(s) }
        pub fn get_shortterm_lufs(&mut self) -> Result<f64, ebur128::Error> { self.meter.loudness_shortterm() }
        pub fn get_integrated_lufs(&mut self) -> Result<f64, ebur128::Error> { self.meter.loudness_global() }
        pub fn get_loudness_range(&mut self) -> Result<f64, ebur128::Error> { self.meter.loudness_range() }
        pub fn get_true_peak(&mut self) -> Result<(f64, f64), ebur128::Error> { Ok((self.meter.true_peak(0)?, self.meter.true_peak(1)?)) }
    }
}
use analyzer::Analyzer;

/// tests/golden/make_golden.py::golden_input, integer arithmetic restated (xorshift-style noise + two triangle waves)
fn golden_input(seed: u64, n: usize, scale: f64) -> Vec<f32> {
    let s = ((seed.wrapping_mul(2654435761).wrapping_add(12345)) & 0xFFFF_FFFF) | 1;
    let (p1, p2) = (37 + seed % 23, 211 + (seed * 7) % 101);
    let tri = |i: u64, p: u64| 2.0 * (2.0 * ((i % p) as f64 / p as f64) - 1.0).abs() - 1.0;
    (0..n as u64).map(|i| {
        let mut x = (i.wrapping_add(s)).wrapping_mul(0x9E37_79B9_7F4A_7C15);
        x ^= x >> 33; x = x.wrapping_mul(0xff51_afd7_ed55_8ccd); x ^= x >> 33;
        let noise = ((x >> 40) as f64 / (1u64 << 24) as f64) * 2.0 - 1.0;
        (scale * (0.5 * tri(i, p1) + 0.3 * tri(i, p2) + 0.1 * noise)) as f32
    }).collect()
}

/// .npy v1.0, C order, little-endian f64
fn save_npy(dir: &Path, name: &str, shape: &[usize], data: &[f64]) {
    let dims = shape.iter().map(|d| d.to_string()).collect::<Vec<_>>().join(", ");
    let dims = if shape.len() == 1 { format!("{dims},") } else { dims };
    let mut header = format!("{{'descr': '<f8', 'fortran_order': False, 'shape': ({dims}), }}");
    while (10 + header.len() + 1) % 64 != 0 { header.push(' '); }
    header.push('\n');
    let mut f = File::create(dir.join(format!("{name}.npy"))).expect("create");
    f.write_all(b"\x93NUMPY\x01\x00").unwrap();
    f.write_all(&(header.len() as u16).to_le_bytes()).unwrap();
    f.write_all(header.as_bytes()).unwrap();
    for v in data { f.write_all(&v.to_le_bytes()).unwrap(); }
}
fn pairs(v: &[(f64, f64)]) -> Vec<f64> { v.iter().flat_map(|p| [p.0, p.1]).collect() }

fn main() {
    let dir = PathBuf::from(std::env::args().nth(1).expect("usage: pin_from_crates <out-dir>"));
    std::fs::create_dir_all(&dir).unwrap();
    // the cases of tests/golden/make_golden.py (CASES_FFT, CASES_WAVE, CASES_METER, BATCH)
    for (rate, n, seed) in [(44100u32, 16384usize, 1u64), (48000, 4096, 2), (96000, 16384, 3), (48000, 256, 4)] {
        let mut a = Analyzer::default();
        a.create_loudness_meter(2, rate).unwrap();
        let r = a.get_fft(&golden_input(seed, n, 0.5)).unwrap();
        save_npy(&dir, &format!("fft_{rate}_{n}_{seed}"), &[r.len(), 2], &pairs(&r));
    }
    for (n, win, seed) in [(44100usize, 15.0f64, 5u64), (9600, 0.1, 6), (1000, 0.3, 7)] {
        let r = Analyzer::get_waveform(&golden_input(seed, n, 0.5), win);
        save_npy(&dir, &format!("wave_{n}_{win:?}_{seed}"), &[r.len(), 2], &pairs(&r));
    }
    for (rate, secs, seed) in [(48000u32, 4.0f64, 8u64), (44100, 3.5, 9)] {
        // stereo cases only: the reference's Analyzer::get_true_peak is hard-wired to channels 0 and 1 (analyzer.rs:159-164)
        let x = golden_input(seed, (rate as f64 * secs) as usize * 2, 0.6);
        let mut a = Analyzer::default();
        a.create_loudness_meter(2, rate).unwrap();
        let mut st = Vec::new();
        for c in x.chunks(16384) {
            a.add_samples(c).unwrap();
            st.push(a.get_shortterm_lufs().unwrap());
        }
        let (l, r) = a.get_true_peak().unwrap();
        save_npy(&dir, &format!("meter_2_{rate}_{seed}"), &[4], &[a.get_integrated_lufs().unwrap(), a.get_loudness_range().unwrap(), l, r]);
        save_npy(&dir, &format!("meter_st_2_{rate}_{seed}"), &[st.len()], &st);
    }
    // the sub-normal question (oracle/ss_oracle.c filter_process, SO_FTZ_*): half a second of programme, then digital silence —
    // short-term readings per 16384-sample call; tests compare them with both models of the oracle
    {
        let rate = 48000u32;
        let mut x = golden_input(12, rate as usize, 0.8);
        x.resize(2 * rate as usize * 6, 0.0);
        let mut a = Analyzer::default();
        a.create_loudness_meter(2, rate).unwrap();
        let mut st = Vec::new();
        for c in x.chunks(16384) { a.add_samples(c).unwrap(); st.push(a.get_shortterm_lufs().unwrap()); }
        save_npy(&dir, "decay_st_2_48000_12", &[st.len()], &st);
        save_npy(&dir, "decay_scalars_2_48000_12", &[2], &[a.get_integrated_lufs().unwrap(), a.get_loudness_range().unwrap()]);
    }
    println!("wrote {}", dir.display());
}
