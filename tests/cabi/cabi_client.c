/* A plain-C99 client of include/soundscope_hip.h: what a maintainer's FFI layer links against.
 * Built and run by tests/test_abi.py (CPU: must fail loudly with SS_ERR_DEVICE) and tests/test_gpu_parity.py
 * (GPU: one tick of the reference driver through the C ABI, no Python in the loop).
 * Prints one line of "key=value" pairs. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "soundscope_hip.h"

int main(void)
{
    const uint32_t rate = 48000, frames = 48000 * 2;
    float *x = (float *)malloc(sizeof(float) * 2 * frames);
    double *mid = (double *)malloc(sizeof(double) * 2 * 8193), *side = (double *)malloc(sizeof(double) * 2 * 8193);
    ss_session *s = NULL;
    ss_tick_result r;
    size_t i;
    int rc;
    if (!x || !mid || !side) return 2;
    for (i = 0; i < frames; i++) {          /* 997 Hz, -6 dBFS, left only; right silent */
        x[2 * i] = 0.5f * (float)sin(2.0 * 3.14159265358979323846 * 997.0 * (double)i / rate);
        x[2 * i + 1] = 0.0f;
    }
    printf("abi=%d devices=%d sizeof_tick=%u ", ss_abi_version(), ss_device_count(), (unsigned)sizeof(ss_tick_result));
    rc = ss_session_open_file(x, (size_t)2 * frames, 2, rate, &s);
    printf("open=%d ", rc);
    if (rc == SS_OK) {
        float gain = 0.0f;
        double peak_x = 0.0, peak_db = -1e9, l = 0.0, rr = 0.0;
        rc = ss_session_tick_file(s, (size_t)2 * 60000, mid, side, 8193, &r);
        printf("tick=%d fft_ran=%d n_mid=%u lufs_ran=%d fed=%d shortterm=%.4f ", rc, r.fft_ran, r.n_mid, r.lufs_ran, r.fed, r.shortterm);
        for (i = 0; i < r.n_mid; i++)
            if (mid[2 * i + 1] > peak_db) { peak_db = mid[2 * i + 1]; peak_x = mid[2 * i]; }
        printf("mid_peak_db=%.3f mid_peak_x=%.4f ", peak_db, peak_x);
        ss_session_gain_db(s, &gain);
        ss_get_true_peak(ss_session_analyzer(s), &l, &rr);
        printf("gain_db=%.3f true_peak_l=%.5f true_peak_r=%.5f ", gain, l, rr);
        ss_session_close(s);
    } else {
        printf("error=\"%s: %s\" ", ss_status_string(rc), ss_last_device_error());
    }
    printf("\n");
    free(x); free(mid); free(side);
    return 0;
}
