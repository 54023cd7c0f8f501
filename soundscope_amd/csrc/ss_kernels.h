// ss_kernels.h — parameter blocks and launchers of the gfx950 kernels.
// Host code (ss_host.cpp, ss_analyzer.cpp, ss_batch.cpp, ss_session.cpp, ss_ingest.cpp) sees only this header; device code lives in ss_fft.hip, ss_time_domain.hip, ss_loudness.hip, ss_util.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <cstddef>
#include <cstdint>

namespace ssk {

// hipFuncSetAttribute applies to the current device only: a launcher prepares its kernel once per device.  The bit of a
// device is set only AFTER the attribute call has succeeded there, under the launcher's mutex — a second thread on the
// same device cannot launch before the attribute exists, two threads cannot both act as "first", and a failure on one
// device leaves the other devices' bits alone.
struct DevicePrep {
    std::mutex mu;
    uint64_t done = 0;
};
template <class F>
inline hipError_t prepare_on_device(DevicePrep &p, F &&prepare)
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    const uint64_t bit = 1ull << (d & 63);
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.done & bit) return hipSuccess;
    const hipError_t e = prepare();
    if (e == hipSuccess) p.done |= bit;
    return e;
}

constexpr int kHistBins = 1000;
constexpr int kMaxChannels = 64;
constexpr int kTpHistMax = 24;    // longest polyphase branch (factor 2)

// ---- spectrum ---------------------------------------------------------------
struct FftBatchParams {
    const float *pcm;            // [stream][frame][channels] f32
    float *out;                  // [stream][window][fft_channels][n_bins] f32 dB (+pink)
    const float *window;         // N (generic kernel) — full Hann
    const float *half_window;    // N (4096 kernel)   — 0.5 * Hann
    const float2 *tw_n;          // W_N^k, k < N  (4096 kernel uses k < 3841; generic k < N/2)
    const float2 *tw_256;        // W_256^k, k < 256 (4096 / 16384 kernels)
    const float2 *tw_core;       // W_4096^k, k < 4096 (16384 kernel)
    const float *pink;           // bin_stride f32 (zero padded), or nullptr for raw dBFS (generic / 16k kernels)
    const float *offpink;        // bin_stride f32: db_offset + pink[bin] (4096 kernels)
    uint64_t frames_per_stream;
    uint64_t first_start;        // frame index where window 0 starts
    uint32_t n_streams;
    uint32_t channels;
    uint32_t n_windows;
    uint32_t hop;
    uint32_t n;                  // FFT length
    uint32_t first_bin, n_bins;
    uint32_t bin_stride;         // floats between output rows (n_bins rounded up to 4: 16-B aligned rows)
    uint32_t windows_per_block;  // 4096 kernel
    float db_offset;             // 10*log10(4/N^2) (4096) or 20*log10(4/N) (generic)
    uint32_t publish_mask;       // 4096 kernels: bit kc set when bins [256kc, 256kc+255] hold a retained bin or a mirror
    const uint32_t *windows_of;   // ragged batches: windows of each stream (nullable = n_windows for all)
    // columns-only spectrum (N3 fused into the epilogue; N = 4096 stereo at hop 1024): the rows are never stored, each
    // is reduced to `cols` chart columns of max(clamp(dB + gain, -100, 0)) — out_cols[stream][window][mid, side][cols]
    float *out_cols;             // nullptr: ordinary rows into `out`
    const uint16_t *bin_col;     // bin_stride entries: chart column of every retained bin, 0xFFFF for the row padding
    const uint2 *col_groups;     // bin_stride / 4 entries, one per group of four bins: x = o0 | o3 << 16 (byte offsets, 4 x column, of the
                                 // first / last bin's column in a row's accumulators), y = n: n bins lie in the first column, the rest in the
                                 // last; n = 0: a general group (more columns inside it, or row padding)
    const uint2 *col_bins;       // the same per bin: four u16 byte offsets per group, 2048 (a spare slot) for the row padding
    const float *col_init;       // cols entries: -inf where the column owns a bin, NaN where it owns none
    const double *integrated;    // per stream: gain = -13 - (float)integrated (tui.rs:1234); nullptr: gain_db for all
    uint32_t cols;               // 1 .. 512
    float gain_db;
    // one-window N = 16384 kernel (a tick): when set, the workgroup of channel ch stores done_value into done_flag[ch] behind its
    // row — in host-visible memory, so that the host can take the row while the rest of the launch is still running
    uint32_t *done_flag;
    uint32_t done_value;
};

// N = 4096 kernels: bit kc set when bins [256 kc, 256 kc + 255] hold a retained bin or the mirror 4096 - k of one (the other
// blocks of the transform are not published to LDS)
inline uint32_t fft4096_publish_mask(uint32_t first_bin, uint32_t n_bins)
{
    const uint32_t lo = first_bin, hi = first_bin + n_bins - 1;
    uint32_t mask = 0;
    for (uint32_t kc = 0; kc < 16; kc++) {
        const uint32_t a0 = 256 * kc, a1 = a0 + 255;
        const bool direct = a0 <= hi + 3 && a1 >= lo;                       // +3: the last group of four may run past
        const bool mirror = a0 <= 4096 - lo && a1 + 3 >= 4096 - hi - 3;
        if (direct || mirror) mask |= 1u << kc;
    }
    return mask;
}
// mid/side packed N=4096 kernel (stereo only).  hop must be a multiple of 256.
hipError_t launch_fft4096_ms(const FftBatchParams &p, hipStream_t s);
// any power-of-two N in [2, 32768]; mode 0: mono buffers (channels == 1),
// 1: stereo -> mid/side, 2: per channel
hipError_t launch_fft_generic(const FftBatchParams &p, int mode, hipStream_t s);
// N = 16384 real FFT per channel (two radix-16 4096-point halves); same modes as the generic kernel
hipError_t launch_fft16k(const FftBatchParams &p, int mode, hipStream_t s);
// N = 4096, hop 1024, one real channel per workgroup run, two windows per transform (mode 0 mono, 2 per channel)
hipError_t launch_fft4096_pairw(const FftBatchParams &p, int mode, hipStream_t s);
hipError_t launch_fft16k_run(FftBatchParams p, int mode, hipStream_t s);   // hop 1024, runs of windows
void fft16k_run_geometry(uint32_t n_streams, uint32_t fft_ch, uint32_t n_windows, uint32_t *windows_per_block, uint32_t *groups);

// ---- time domain ------------------------------------------------------------
struct TdConst {                 // one per (rate, true-peak factor), device resident
    double b[5], a[5];
    double m_pow[8][16];         // (A^L)^(2^k), A = zero-input transition, L = td_chunk_frames(C)
    double m_pow_split[8][16];   // the same for L = td_split_chunk_frames(C): streaming calls shared by a workgroup's waves (SPLIT)
    double m_step_split[68][16]; // A^n, n = 0 .. 67 (same coordinates): the partial last chunk of a SPLIT tile (n < L <= 65)
    double m_chunk_split[64][16];// (A^L)^(c + 1) for chunk c of a SPLIT tile: what the state in front of the tile adds to the state
                                 // behind chunk c — applied when that state arrives, behind a scan that ran without it
    double m_chunk[64][16];      // the same for L = td_chunk_frames(C): batches whose streams are walked by a whole workgroup (split_batch)
    float tp[3][kTpHistMax];     // polyphase branches 1..factor-1, coefficient of x[n - t]
    int32_t tp_factor;           // 0, 2, 4
    int32_t tp_len;              // taps per branch (12 or 24)
    uint32_t s100;               // frames per 100 ms sub-block = (rate + 5) / 10
    uint32_t st_off;             // 1: no short-term blocks at this rate — thirty sub-blocks are more frames than ebur128's 3 s ring holds
                                 // (rates under 145 Hz that round UP to their sub-block: 16 Hz -> 60 > 48), its energy_shortterm fails
                                 // and add_frames skips the block; loudness range then reads 0
};

struct TdState {                 // per stream / per handle, device resident
    double v[kMaxChannels][4];           // DF-II state v1..v4
    double acc[kMaxChannels];            // energy of the current incomplete sub-block
    float tp_hist[kMaxChannels][kTpHistMax]; // last samples (newest first is index 0)
    float sample_peak[kMaxChannels];
    float true_peak[kMaxChannels];
    uint64_t frames_fed;                 // since reset
    // Non-finite input (NaN, +-Inf).  In the crate such a sample poisons the channel's DF-II state for good: every later filtered
    // sample, hence every later gating block, is NaN and `sum >= boundary` never holds again.  A wave notes the FIRST sub-block
    // in which it met one: bad_key[c] = ~index (0 = none yet, so the zero fill of a reset is "none" and atomicMax keeps the
    // earliest); the gating kernels read every sub-block of a weighted channel BEHIND that one as NaN — which a launch cut into
    // time segments (each starting from a zero state) would otherwise not know.
    uint32_t bad_key[kMaxChannels];
};

struct TdParams {
    const float *pcm;            // [stream][frame][channels]
    uint64_t stream_stride;      // floats between consecutive streams
    uint64_t n_frames;           // frames to consume from each stream this call
    uint32_t n_streams;
    uint32_t channels;
    const TdConst *k;
    TdState *state;              // [stream]
    double *subblocks;           // [stream][slot][channels] f64, slot = sub-block index % sub_cap
    uint64_t sub_stride;         // doubles between streams
    uint32_t sub_cap;            // ring capacity in sub-blocks
    double *ring;                // optional filtered-sample ring [ring_frames][channels] (handle API)
    uint64_t ring_frames;
    int32_t tp_factor;           // must equal k->tp_factor (selects the kernel instantiation)
    uint32_t s100;               // must equal k->s100
    uint32_t nseg;               // time segments per stream (1 for streaming calls)
    uint32_t seg_sub;            // sub-blocks per segment (nseg > 1)
    uint32_t warm_sub;           // run-in sub-blocks of segments > 0
    // fused min-max decimation (batch only; nullptr = off): out[stream][bin] = (min, max)
    float *wave_out;
    uint64_t wave_stride;        // floats between streams
    uint32_t wave_window;        // number of decimation bins W
    uint32_t halo_frames;        // frames kept in front of each tile: >= longest bin, multiple of 4
    const uint64_t *frames_of;    // ragged batches: frames of each stream (nullable = n_frames for all)
    uint32_t tp_f32;              // 1: the factor-4 true peak as the f32 MFMA product everywhere (no f16 split)
    // Exact segment hand-over (batches, nseg > 1, warm_sub == 0): every segment's wave leaves the filter state behind its last frame in
    // seg_state[stream][segment][channel][4]; a second, light launch (fixup = 1: no true peak, no decimation) then re-runs the first
    // fix_sub sub-blocks of every segment > 0 from the state the segment in front of it left, and overwrites their energies.
    double *seg_state;
    uint32_t fixup, fix_sub;
    uint32_t split_batch;         // 2: the same with eight waves per (stream, SEGMENT) — a handful of streams cut into short segments,
                                  // where the length of the chain of tiles is what a pass takes.
                                  // 1 (batches, nseg == 1): a stream is ONE segment walked by the four waves of a workgroup, the filter
                                  // state handed from tile to tile through LDS (SPLIT with the batch's chunk length) — no run-in, nothing
                                  // of the recurrence truncated
    // a tick's short-term reading inside the same launch (k_tick only; st_out == nullptr: off).  The window of st_frames frames
    // ends with this call; the st_old_total ring elements from st_begin_elem on are the part in front of the call.
    double *st_out;               // (energy, loudness): device or mapped host memory
    double *st_scratch;           // kRingTickBlocks partial sums + the count of finished workgroups (zero between launches)
    const double *st_weights;     // [channels]
    double st_frames;
    uint32_t st_begin_elem, st_old_total, st_blocks;
};
// tick_fft (streaming calls on the ring only): the one-window mid/side spectrum of a tick (N = 16384, out rows [mid, side]) to
// run in the SAME launch, beside the loudness call (k_tick), together with the short-term reading if p.st_out is set;
// *tick_fused tells whether that happened — if not, only the time-domain kernel was launched (p.st_out ignored) and the spectrum
// and the reading are the caller's to launch
hipError_t launch_time_domain(const TdParams &p, hipStream_t s, const FftBatchParams *tick_fft = nullptr, bool *tick_fused = nullptr);
// the second launch of the exact segment hand-over (see TdParams::seg_state): p as given to launch_time_domain
hipError_t launch_time_domain_fixup(const TdParams &p, hipStream_t s);
// frames per sequential chunk for a channel count (the constant block's m_pow must match)
uint32_t td_chunk_frames(uint32_t channels, uint32_t s100);
// the same for a streaming call whose tiles are shared by the waves of one workgroup (SPLIT): see ss_time_domain.hip
uint32_t td_split_chunk_frames(uint32_t channels, uint32_t s100);
uint32_t td_resident_waves_per_cu(uint32_t channels, uint32_t s100, uint32_t halo_frames);

struct FinalizeParams {
    const TdConst *k;
    const double *subblocks; uint64_t sub_stride; uint32_t sub_cap;
    const double *hist_energies;     // 1000
    const double *hist_bounds;       // 1001
    const double *weights;           // [channels]
    uint64_t *hist;                  // [stream][2][1000] (block, short-term)
    uint64_t *corpus_hist;           // [2][1000] or nullptr
    uint32_t n_streams, channels;
    uint64_t sub_begin, sub_end;     // absolute sub-block index range completed by this call
    double *out_integrated;          // [stream] (nullable)
    double *out_lra;                 // [stream] (nullable)
    uint32_t *out_counts;            // [stream][2] gating / short-term blocks evaluated (nullable)
    const uint32_t *sub_end_of;   // ragged batches: sub-blocks of each stream (nullable = sub_end for all)
    const TdState *state;         // [stream]: bad_key (first sub-block with a non-finite sample per channel); nullable
    // streaming form only (one handle): behind the histogram updates the same wave takes the handle's readings — (integrated,
    // range) into readings_out, the peaks and the flag as in ReadingsExtra below (readings_out == nullptr: off)
    double *readings_out;
    const float *readings_peaks_src; float *readings_peaks_dst; uint32_t *readings_flag; uint32_t readings_seq;
};
hipError_t launch_finalize(const FinalizeParams &p, hipStream_t s);
// gate + LRA on explicit histograms (corpus gate after the all-reduce; handle getters)
// `peaks` (optional): a handle's readings in one launch — 2 * kMaxChannels floats copied from peaks_src to peaks_dst beside the
// evaluation, then `seq` stored into *flag (host-visible memory: whoever sees the flag sees out2 and the peaks)
struct ReadingsExtra { const float *peaks_src; float *peaks_dst; uint32_t *flag; uint32_t seq; };
hipError_t launch_hist_eval(const uint64_t *hist2000, const double *energies, const double *bounds,
                            double *out2, hipStream_t s, const ReadingsExtra *peaks = nullptr);
// mean-square of the filtered ring over the last `frames` frames (handle getters)
constexpr int kRingScratchDoubles = 320;     // k_ring_energy: 256 partial sums + the completion counter; k_tick: kRingTickBlocks + 1 from kRingTickScratch on
constexpr int kRingTickScratch = 264;        // where a tick launch keeps ITS partial sums and count (the two kernels never share a slot)
constexpr int kRingTickBlocks = 32;          // workgroups of a tick launch that sum the ring (at most 64: one lane of the last wave each)
hipError_t launch_ring_energy(const double *ring, uint64_t ring_frames, uint32_t channels,
                              uint64_t end_frame, uint64_t frames, const double *weights,
                              double *out2 /* energy, loudness: device or mapped host memory */,
                              double *scratch /* kRingScratchDoubles, zero before the first launch */,
                              hipStream_t s);

// ---- waveform ---------------------------------------------------------------
struct WaveParams {
    const float *pcm; uint64_t stream_stride; uint64_t n_samples; // interleaved samples per stream
    uint32_t n_streams; uint32_t window;   // number of decimation bins W
    float *out; uint64_t out_stride;       // [stream][W][2] f32 (min, max)
    const uint64_t *samples_of;   // ragged batches: interleaved samples / bins of each stream (nullable)
    const uint32_t *window_of;
    uint32_t mid_of_pairs;        // 1: `pcm` holds n_samples stereo PAIRS and the signal decimated is their mid, (l + r) / 2
                                  // (audio_player.rs:400-419 on the fly: the capture tick's chart, tui.rs:1453-1455)
};
hipError_t launch_waveform(const WaveParams &p, hipStream_t s);

// ---- utilities --------------------------------------------------------------
// raw little-endian PCM -> f32 (format: 1 u8, 2 s16, 3 s24, 4 s32, 5 f32, 6 f64)
hipError_t launch_pcm_to_f32(const void *src, size_t n_samples, int format, float *dst, hipStream_t s);
// render-side reductions (N3)
hipError_t launch_render_spectrum(const float *rows, uint32_t bin_stride, uint32_t n_bins, uint64_t n_rows,
                                  uint32_t rows_per_stream, const uint32_t *col_start, uint32_t cols,
                                  const double *integrated, float gain_db, float *out, hipStream_t s);
hipError_t launch_render_waveform(const float *wave, uint64_t wave_stride, uint32_t n_points, uint32_t n_streams,
                                  uint32_t x_min, uint32_t x_max, uint32_t cols, float *out, hipStream_t s);
hipError_t launch_mid_side(const float *interleaved, size_t frames, float *mid, float *side, hipStream_t s);
// pairs of an interleaved buffer whose mid or side value ((l + r) / 2, (l - r) / 2 in f32) is NaN or infinite — normally none:
// the tick drivers' index for the crate's NaN / infinity rejection.  cls bits: 1 mid NaN, 2 mid inf, 4 side NaN, 8 side inf.
// *count (zeroed by the launcher) counts ALL such pairs; the first `cap` to arrive are listed (in no particular order).
struct NonFinitePair { unsigned long long index; uint32_t cls; uint32_t pad; };
hipError_t launch_nonfinite_pairs(const float *interleaved, size_t pairs, uint32_t *count, NonFinitePair *list, uint32_t cap, hipStream_t s);
// verification utility: out[item * out_stride] += order-independent checksum of words [item * stride_words, + words) (32-bit words)
hipError_t launch_checksum(const void *base, uint64_t words, uint64_t stride_words, uint32_t n_items, uint64_t *out,
                           uint32_t out_stride, hipStream_t s);
// clears up to four device buffers (byte counts: multiples of four; null / zero entries are skipped) in one launch
hipError_t launch_zero4(void *const ptrs[4], const size_t bytes[4], hipStream_t s);
// measurement utility: k_fft4096_ms1's loads and stores with no arithmetic (same grid, occupancy and addresses)
hipError_t launch_fft4096_traffic(const FftBatchParams &p, hipStream_t s);
hipError_t launch_synth(float *pcm, uint32_t n_streams, uint64_t frames, uint32_t channels,
                        uint32_t rate, uint64_t seed, uint32_t first_id, hipStream_t s);

}  // namespace ssk
