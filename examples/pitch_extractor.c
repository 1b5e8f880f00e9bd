/*
 * examples/pitch_extractor.c -- the user loop of the reference's examples/pitch_detection.rs:23-30
 * (Windower::hanning(2048, 1024) over a sine, then PitchExtractor) through the C ABI, in plain C.
 *
 *   gcc -std=c11 -Iinclude examples/pitch_extractor.c -Lvox_box.rs_amd/lib -lvoxbox_hip \
 *       -Wl,-rpath,$PWD/vox_box.rs_amd/lib -lm -o pitch_extractor && ./pitch_extractor      (needs an MI355X)
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "voxbox_hip.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != VBX_SUCCESS) { fprintf(stderr, "%s: %s\n", #call, vbx_last_error(ctx)); return 1; } \
    } while (0)

int main(void) {
    const double sample_rate = 44100.0, hz = 150.0;
    const size_t n_samples = 44100, bin = 2048, hop = 1024, kmax = 1;
    double *h_audio = (double *)malloc(n_samples * sizeof(double));
    for (size_t i = 0; i < n_samples; i++) h_audio[i] = sin(2.0 * M_PI * hz * (double)i / sample_rate);

    vbx_ctx *ctx = NULL;
    if (vbx_ctx_create(&ctx, 0, NULL) != VBX_SUCCESS) { fprintf(stderr, "no gfx950 device: %s\n", vbx_last_error(NULL)); return 2; }

    /* Windower::hanning(&samples, bin, hop): a strided view plus a window table */
    const size_t n_frames = vbx_frame_count(n_samples, bin, hop);
    double *h_win = (double *)malloc(bin * sizeof(double));
    CHECK(vbx_window_table_f64(VBX_WINDOW_HANNING, bin, h_win));

    void *d_audio = NULL, *d_win = NULL, *d_cand = NULL, *d_count = NULL, *d_status = NULL;
    CHECK(vbx_malloc(ctx, &d_audio, n_samples * sizeof(double)));
    CHECK(vbx_malloc(ctx, &d_win, bin * sizeof(double)));
    CHECK(vbx_malloc(ctx, &d_cand, n_frames * kmax * sizeof(vbx_pitch)));
    CHECK(vbx_malloc(ctx, &d_count, n_frames * sizeof(int32_t)));
    CHECK(vbx_malloc(ctx, &d_status, n_frames * sizeof(int32_t)));
    CHECK(vbx_memcpy_h2d(ctx, d_audio, h_audio, n_samples * sizeof(double)));
    CHECK(vbx_memcpy_h2d(ctx, d_win, h_win, bin * sizeof(double)));

    /* chunk.pitch::<window::Hanning>(44100., 0.2, 0.05, 0.05, 100., 500.)[0] for every frame, one launch */
    CHECK(vbx_pitch_f64(ctx, (const double *)d_audio, n_frames, bin, hop, (const double *)d_win, sample_rate, 0.2, 100.0, 500.0,
                        kmax, (vbx_pitch *)d_cand, (int32_t *)d_count, (int32_t *)d_status));
    vbx_pitch *h_cand = (vbx_pitch *)malloc(n_frames * kmax * sizeof(vbx_pitch));
    CHECK(vbx_memcpy_d2h(ctx, h_cand, d_cand, n_frames * kmax * sizeof(vbx_pitch)));
    CHECK(vbx_sync(ctx));
    for (size_t t = 0; t < n_frames && t < 4; t++)
        printf("frame %zu: %.6f Hz  strength %.6f\n", t, h_cand[t * kmax].frequency, h_cand[t * kmax].strength);

    vbx_free(ctx, d_audio); vbx_free(ctx, d_win); vbx_free(ctx, d_cand); vbx_free(ctx, d_count); vbx_free(ctx, d_status);
    vbx_ctx_destroy(ctx);
    free(h_audio); free(h_win); free(h_cand);
    return 0;
}
