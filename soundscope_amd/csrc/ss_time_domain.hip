// ss_time_domain.hip: K-weighting, sub-block energies, peaks, fused decimation — hand-written gfx950 (CDNA4, wave64) kernels of the soundscope analyzer hot path.
// Reference semantics: /root/reference/src/analyzer.rs (get_fft :55-105, get_waveform :107-137,
// add_samples/getters :139-164, calculate_integrated_lufs :170-182) and src/audio_player.rs:400-419, plus the
// arithmetic of ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 as restated in DESIGN.md.
// Nothing here is translated from the reference: the reference has no GPU code.
// This file: the host entry points and the chunk-length model; the kernels live in ss_td_impl.h and are instantiated per true-peak
// factor in ss_td_f4.hip / ss_td_f2.hip / ss_td_f0.hip.
#include "ss_td_impl.h"

namespace ssk {
// Chunk length L (frames one lane filters per tile; a tile is 64 / C chunks).  Candidates are rated by a model of the
// sequential work per 100 ms sub-block, in units of one batched K-weighting step of a lane, calibrated on chunk-length
// sweeps of four shapes (tools/sweep_td_chunk.py: 48 / 96 / 44.1 kHz stereo, 96 kHz 8 channels):
//   pieces x (F + steps(L) + 1.2 x frames of the incomplete last chunk)  x  occupancy(workgroups per CU)  x  exposure
//   F        ~ 80   what a tile costs whatever its length (staging, scan, conversion set-up, decimation, tile tail),
//   steps(L) = L, a step outside whole batches of kTdBatch (a look-ahead single) counting 1.5, plus 3 % per extra way of
//              LDS bank conflict of the per-lane walk (lane (chunk, channel) reads tile[chunk L C + channel]),
//   an incomplete last chunk runs the plain recurrence on one lane per channel, both passes: ~1.2 steps per frame,
//   occupancy: three workgroups per CU cost 1.25x, two 1.6x their step count (longer tiles need more LDS),
//   exposure : a tile beyond the 2048 floats the prefetch registers hold loads the rest at the point of use; tiles that
//              do not start on 16-byte boundaries are staged with scalar loads (x 1.6).
// A length that cuts the sub-block into whole tiles of whole chunks keeps every lane busy and has no incomplete chunk:
// 48 kHz stereo 4800 = 5 x 32 x 30 — k_time_domain 2.14 -> 1.98 ms against L = 33 (5 tiles of 29 chunks + 3 frames)
// although the walk of an even L is 2-way bank conflicted; 96 kHz stereo 2.91 -> 2.17 ms (was L = 65); BASELINE
// config 5 (96 kHz, 8 channels) 3.02 -> 2.08 ms (was L = 65: 19 tiles of 7 chunks + 51 frames; now 40 tiles of 8 x 30).
uint32_t td_lds_blocks(uint32_t C, uint32_t tile_len)
{
    uint32_t wave_floats = ((uint32_t)kTdHaloFrames + tile_len) * C + td_slack_floats(C) + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    const size_t lds = (size_t)wave_floats * 4 * kTdWavesPerBlock;
    uint32_t blocks = (uint32_t)((160u * 1024u) / lds);
    const uint32_t max_blocks = (4u * SS_TD_WAVES) / kTdWavesPerBlock;
    return blocks > max_blocks ? max_blocks : blocks;
}

uint32_t td_chunk_frames(uint32_t C, uint32_t s100)
{
#ifdef SS_TUNING        // development builds only: force a chunk length (tools/sweep_td_chunk.py)
    if (const char *e = std::getenv("SS_TD_L")) { const int v = std::atoi(e); if (v >= 8 && v <= 128) return (uint32_t)v; }
#endif
    const uint32_t nch = 64u / C;
    const bool rowscan = C == 1 || C == 2 || C == 8;          // the compile-time channel counts whose chunks are dealt to the DPP rows
    static const double occupancy[5] = {8.0, 3.0, 1.6, 1.25, 1.0};
    uint32_t best = 33; double best_cost = 1e300;
    for (uint32_t L : {20u, 25u, 30u, 33u, 35u, 40u, 45u, 49u, 50u, 55u, 60u, 65u}) {
        const uint32_t cap = nch * L;
        const uint32_t pieces = (s100 + cap - 1) / cap;
        uint32_t tile_len = (s100 + pieces - 1) / pieces;
        if (tile_len > cap) tile_len = cap;
        const uint32_t blocks = td_lds_blocks(C, tile_len);
        if (blocks == 0) continue;
        const uint32_t rem = tile_len % L;
        uint32_t ways = 1;                                    // bank conflicts of one pass read, per half wave
        for (uint32_t half = 0; half < 2; half++) {
            uint32_t hits[64] = {0};
            for (uint32_t l = 32 * half; l < 32 * half + 32; l++) {
                uint32_t chunk, ch;
                if (rowscan) { const uint32_t q = (l & 15u) / C; chunk = (l >> 4) + 4u * q; ch = (l & 15u) - q * C; }
                else { chunk = l / C; ch = l - chunk * C; }
                if (chunk >= nch) continue;
                const uint32_t n = ++hits[(chunk * L * C + ch) & 63u];
                if (n > ways) ways = n;
            }
        }
        const double steps = ((double)(kTdBatch * (L / kTdBatch)) + 1.5 * (double)(L % kTdBatch)) * (1.0 + 0.03 * (double)(ways - 1));
        const uint32_t tile_floats = tile_len * C;
        const double exposure = 1.0 + (tile_floats > 2048u ? 0.35 * (double)(tile_floats - 2048u) / 2048.0 : 0.0);
        double cost = (double)pieces * (80.0 + steps + 1.2 * (double)rem) * occupancy[blocks > 4 ? 4 : blocks] * exposure;
        if ((tile_floats & 3u) != 0u) cost *= 1.6;           // tiles that do not start on 16 bytes are staged with scalar loads
        if (cost < best_cost) { best_cost = cost; best = L; }
    }
    return best;
}

// Chunk length of a streaming call shared by the eight waves of a workgroup (SPLIT).  A different trade from the batch's: what
// counts is the length of the chain, and a wave that has to take a second tile waits for its first one's true-peak product
// before it can even stage it.
uint32_t td_split_chunk_frames(uint32_t C, uint32_t s100)
{
#ifdef SS_TUNING        // development builds only: force the chunk length of SPLIT calls
    if (const char *e = std::getenv("SS_TD_LSPLIT")) { const int v = std::atoi(e); if (v >= 8 && v <= 128) return (uint32_t)v; }
#endif
    // The smallest chunk length that cuts the sub-block into whole tiles of whole chunks, lets a tick-sized call (16384 / C
    // frames, tui.rs:1539) fit ONE round of the eight waves whatever its alignment, and fits the LDS; otherwise the batch's.
    // (48 kHz stereo: 40 -> tiles of 1200 frames, at most eight per tick, 16.4 us for the call where L = 30 — nine or ten tiles
    // of 960, wave 0 taking a second one behind its first one's true-peak product — takes 22.7; partial chunks are poison: a
    // tile that ends inside a chunk runs the unbatched loops, L = 35 / 38 / 45: 84-98 us, tools/sweep_tick_lsplit.sh.)
    const uint32_t nch = 64u / C;
    const uint32_t n_tick = 16384u / C;
    for (uint32_t L : {25u, 30u, 33u, 35u, 40u, 45u, 49u, 50u, 55u, 60u, 65u}) {
        const uint32_t cap = nch * L;
        const uint32_t pieces = (s100 + cap - 1) / cap;
        const uint32_t tile_len = (s100 + pieces - 1) / pieces;
        if (tile_len > cap || tile_len * pieces != s100 || tile_len % L) continue;
        if ((n_tick - 1u) / tile_len + 2u > (uint32_t)kTdSplitWaves) continue;
        uint32_t wave_floats = ((uint32_t)kTdHaloFrames + tile_len) * C + td_slack_floats(C) + kMaxChannels;
        wave_floats = (wave_floats + 3u) & ~3u;
        if ((size_t)wave_floats * 4 * kTdSplitWaves + sizeof(TdShare) > 160 * 1024) continue;
        return L;
    }
    return td_chunk_frames(C, s100);
}

// waves of k_time_domain one CU holds at once (LDS per wave grows with the channel count and the decimation halo)
uint32_t td_resident_waves_per_cu(uint32_t C, uint32_t s100, uint32_t halo_frames)
{
    const uint32_t L = td_chunk_frames(C, s100);
    const uint32_t cap = (64u / C) * L;
    const uint32_t pieces = (s100 + cap - 1) / cap;
    uint32_t tile_len = (s100 + pieces - 1) / pieces;
    if (tile_len > cap) tile_len = cap;
    const uint32_t halo = halo_frames ? halo_frames : (uint32_t)kTdHaloFrames;
    uint32_t wave_floats = (halo + tile_len) * C + td_slack_floats(C) + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    const size_t lds = (size_t)wave_floats * 4 * kTdWavesPerBlock;
    uint32_t blocks = lds ? (uint32_t)((160u * 1024u) / lds) : 4u;
    const uint32_t max_blocks = (4u * SS_TD_WAVES) / kTdWavesPerBlock;      // launch bound: SS_TD_WAVES waves per SIMD
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    return blocks * kTdWavesPerBlock;
}

hipError_t launch_time_domain(const TdParams &p, hipStream_t s, const FftBatchParams *tick_fft, bool *tick_fused)
{
    bool fused_local = false;
    bool *fused = tick_fused ? tick_fused : &fused_local;
    *fused = false;
    if (p.n_streams == 0 || p.n_frames == 0) return hipSuccess;
    const bool ring = p.ring != nullptr;
    switch (p.tp_factor) {
        case 4: return td_launch_f4(p, s, tick_fft, fused, ring);
        case 2: return td_launch_f2(p, s, tick_fft, fused, ring);
        default: return td_launch_f0(p, s, tick_fft, fused, ring);
    }
}

// Second launch of the exact segment hand-over: filter and energies only (the FACTOR = 0, WAVE = 0 instantiations), one wave per
// segment > 0 over its first fix_sub sub-blocks, from the state the segment in front of it left in p.seg_state.
hipError_t launch_time_domain_fixup(const TdParams &p, hipStream_t s)
{
    if (p.n_streams == 0 || p.n_frames == 0 || p.nseg < 2 || !p.seg_state || !p.fix_sub) return hipSuccess;
    TdParams q = p;
    q.fixup = 1u; q.wave_out = nullptr; q.wave_window = 0; q.halo_frames = 0; q.tp_factor = 0;      // (split_batch as in the main launch)
    bool fused = false;
    return td_launch_f0(q, s, nullptr, &fused, false);
}

}  // namespace ssk
