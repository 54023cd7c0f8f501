// ss_ingest.cpp — PCM ingest: RIFF/WAVE header walk on the host, sample-format conversion on the device
// (what symphonia's SampleBuffer::<f32>::copy_interleaved_ref does for PCM, /root/reference/src/audio_player.rs:169-267).
#include "ss_host.h"

using namespace ssh;

extern "C" {

size_t ss_pcm_sample_bytes(int format)
{
    switch (format) {
        case SS_PCM_U8: return 1; case SS_PCM_S16: return 2; case SS_PCM_S24: return 3;
        case SS_PCM_S32: return 4; case SS_PCM_F32: return 4; case SS_PCM_F64: return 8;
        default: return 0;
    }
}

// RIFF/WAVE header walk: "RIFF" size "WAVE", then chunks (id, size, payload padded to even)
int ss_wav_parse(const void *file_bytes, size_t len, ss_wav_info *out)
{
    if (!file_bytes || !out) return SS_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    const unsigned char *p = static_cast<const unsigned char *>(file_bytes);
    auto u16 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    if (len < 12 || std::memcmp(p, "RIFF", 4) != 0 || std::memcmp(p + 8, "WAVE", 4) != 0) return SS_ERR_INVALID_ARG;
    size_t pos = 12;
    bool have_fmt = false;
    uint32_t tag = 0, block_align = 0;
    while (pos + 8 <= len) {
        const uint32_t size = u32(pos + 4);
        const size_t body = pos + 8;
        if (std::memcmp(p + pos, "fmt ", 4) == 0) {
            if (size < 16 || body + 16 > len) return SS_ERR_INVALID_ARG;
            tag = u16(body);
            out->channels = u16(body + 2);
            out->sample_rate = u32(body + 4);
            block_align = u16(body + 12);
            out->bits_per_sample = u16(body + 14);
            if (tag == 0xFFFE) {                        // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                if (size < 40 || body + 40 > len) return SS_ERR_INVALID_ARG;
                tag = u16(body + 24);
            }
            have_fmt = true;
        } else if (std::memcmp(p + pos, "data", 4) == 0) {
            if (!have_fmt) return SS_ERR_INVALID_ARG;
            out->data_offset = body;
            uint64_t avail = len - body;
            out->data_bytes = size < avail ? size : avail;   // tolerate a truncated / streaming length
            break;
        }
        pos = body + (size_t)size + (size & 1u);
    }
    if (!have_fmt || !out->data_offset) return SS_ERR_INVALID_ARG;
    if (out->channels == 0) return SS_ERR_INVALID_ARG;
    const uint32_t bits = out->bits_per_sample;
    if (tag == 1) {
        out->format = bits == 8 ? SS_PCM_U8 : bits == 16 ? SS_PCM_S16 : bits == 24 ? SS_PCM_S24 : bits == 32 ? SS_PCM_S32 : 0;
    } else if (tag == 3) {
        out->format = bits == 32 ? SS_PCM_F32 : bits == 64 ? SS_PCM_F64 : 0;
    } else {
        return SS_ERR_UNSUPPORTED;
    }
    if (!out->format) return SS_ERR_UNSUPPORTED;
    const size_t fb = ss_pcm_sample_bytes((int)out->format) * out->channels;
    if (block_align && block_align != fb) return SS_ERR_UNSUPPORTED;
    out->frames = out->data_bytes / fb;
    return SS_OK;
}

int ss_pcm_decode(const void *pcm, size_t n_samples, int format, float *out)
{
    const size_t sb = ss_pcm_sample_bytes(format);
    if (!sb) return SS_ERR_INVALID_ARG;
    if (!n_samples) return SS_OK;
    if (!pcm || !out) return SS_ERR_INVALID_ARG;
    if (require_device()) return SS_ERR_DEVICE;
    Scratch &c = scratch();
    if (!c.stream) HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIPCHK(c.raw.ensure(n_samples * sb + 8));
    HIPCHK(c.out.ensure(n_samples));
    HIPCHK(hipMemcpyAsync(c.raw.p, pcm, n_samples * sb, hipMemcpyHostToDevice, c.stream));
    HIPCHK(ssk::launch_pcm_to_f32(c.raw.p, n_samples, format, c.out.p, c.stream));
    HIPCHK(hipMemcpyAsync(out, c.out.p, n_samples * sizeof(float), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return SS_OK;
}
}  // extern "C"
