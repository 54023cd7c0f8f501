/* A plain-C99 client of the batch and corpus-gate part of include/soundscope_hip.h — what a host program (C, or Rust /
 * Go through their C FFI) runs per rank of a multi-GPU job, with no Python and no PyTorch in the process:
 * batch of synthetic streams -> one pass -> the corpus gate queued on the device -> results.
 * With RANK / WORLD_SIZE in the environment it joins the communicator the launcher describes (SS_COMM_TRANSPORT=host-tcp
 * for ranks that share a GPU); alone it runs as the only rank.  Built and run by tests/test_abi.py (CPU: must fail loudly
 * with SS_ERR_DEVICE) and tests/test_gpu_parity.py (GPU).  Prints one line of "key=value" pairs. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "soundscope_hip.h"

int main(void)
{
    ss_batch_config cfg;
    ss_batch *b = NULL;
    ss_comm *comm = NULL;
    ss_stream_result res[16];
    ss_batch_geometry geo;
    uint64_t hist[2000], blocks = 0;
    double tp[2], sp[2], gate_i = 0.0, gate_lra = 0.0;
    const char *world = getenv("WORLD_SIZE"), *transport = getenv("SS_COMM_TRANSPORT");
    int rc, i;
    memset(&cfg, 0, sizeof cfg);
    cfg.sample_rate = 48000; cfg.channels = 2; cfg.n_streams = 16; cfg.fft_n = 4096; cfg.hop_frames = 1024;
    cfg.flags = SS_BATCH_ALL; cfg.frames_per_stream = 48000 * 3;
    printf("abi=%d devices=%d sizeof_cfg=%u sizeof_result=%u ", ss_abi_version(), ss_device_count(),
           (unsigned)sizeof cfg, (unsigned)sizeof res[0]);
    if (world && atoi(world) > 1) {
        rc = ss_comm_init_from_env(transport && !strcmp(transport, "host-tcp") ? SS_COMM_HOST_TCP : SS_COMM_RCCL, &comm);
        printf("comm=%d ranks=%d ", rc, rc == SS_OK ? ss_comm_size(comm) : 0);
        if (rc != SS_OK) { printf("error=\"%s: %s\"\n", ss_status_string(rc), ss_last_device_error()); return 0; }
    }
    rc = ss_batch_create(&cfg, &b);
    printf("create=%d ", rc);
    if (rc != SS_OK) { printf("error=\"%s: %s\"\n", ss_status_string(rc), ss_last_device_error()); return 0; }
    rc = ss_batch_synthesize(b, 0x5EED0000u, comm ? 16u * (uint32_t)ss_comm_rank(comm) : 0u);
    if (rc == SS_OK) rc = ss_batch_set_overlap(b, 2);
    if (rc == SS_OK) rc = ss_batch_run(b);
    if (rc == SS_OK) rc = ss_batch_corpus_gate_enqueue(b, comm);       /* [all-reduce] + gate + LRA, no host wait */
    if (rc == SS_OK) rc = ss_batch_corpus_gate_read(b, &gate_i, &gate_lra);
    if (rc == SS_OK) rc = ss_batch_results(b, res, 16);
    if (rc == SS_OK) rc = ss_batch_peaks(b, 5, tp, sp, 2);
    if (rc == SS_OK) rc = ss_batch_histograms(b, hist);
    if (rc == SS_OK) rc = ss_batch_geometry_get(b, &geo);
    printf("run=%d ", rc);
    if (rc == SS_OK) {
        for (i = 0; i < 1000; i++) blocks += hist[i];
        printf("gate_lufs=%.12f gate_lra=%.12f blocks=%llu overlap=%u ", gate_i, gate_lra, (unsigned long long)blocks, geo.overlap);
        printf("i0=%.12f i15=%.12f lra7=%.12f tp5l=%.9f tp5r=%.9f sp5l=%.9f host_gate=%.12f ", res[0].integrated_lufs,
               res[15].integrated_lufs, res[7].loudness_range, tp[0], tp[1], sp[0], ss_corpus_integrated_lufs(hist));
    } else {
        printf("error=\"%s: %s\" ", ss_status_string(rc), ss_last_device_error());
    }
    printf("\n");
    ss_batch_destroy(b);
    if (comm) ss_comm_destroy(comm);
    return 0;
}
