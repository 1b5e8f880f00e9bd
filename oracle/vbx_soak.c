/*
 * vbx_soak.c -- walks CONSECUTIVE frames of a recording with the oracle on n threads and stores every per-frame
 * result, so that a GPU parity test can hold tens of thousands of frames to the oracle in seconds
 * (tests/test_gpu_soak.py; the pytest suite's per-frame Python loops manage hundreds).
 *
 * TEST INFRASTRUCTURE ONLY (see vbx_oracle.h).  Nothing here is arithmetic of its own: every number comes from the
 * restated reference routines of vbx_oracle.c, called once per frame exactly as the reference's user loop calls the
 * traits (examples/pitch_detection.rs:23-30, tests/lib.rs:71-83):
 *
 *   VBXO_SOAK_PITCH     Windower::hanning frame -> Pitched::pitch                 (src/periodic.rs:396-455)
 *   VBXO_SOAK_LPC       the same frame -> autocorrelate(p + 1) -> lpc(p)          (src/periodic.rs:276-289, src/spectrum.rs:63-92)
 *   VBXO_SOAK_MFCC      the same frame -> mfcc(k, (lo, hi), sr)                   (src/spectrum.rs:410-440)
 *   VBXO_SOAK_FORMANTS  the rectangle frame -> find_formants up to the sorted resonance row: Burg coefficients,
 *                       resonances[32], status (src/lib.rs:40-110); the tracker (:114) is sequential over frames and
 *                       runs afterwards in vbxo_soak_track.
 */
#define _GNU_SOURCE
#include "vbx_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

enum { VBXO_SOAK_PITCH = 1, VBXO_SOAK_LPC = 2, VBXO_SOAK_MFCC = 4, VBXO_SOAK_FORMANTS = 8 };

typedef struct {
    const double *audio;
    size_t first, count, frame_len, hop, order, n_mfcc;
    double sample_rate, threshold, fmin, fmax, mfcc_lo, mfcc_hi;
    int what;
    const double *w_hann;
    atomic_ulong *next;
    int32_t *pitch_status, *pitch_count, *mfcc_status, *ff_status, *res_count;
    double *pitch_top, *r, *a, *mfcc, *burg, *res;
} soak_job_t;

static void *soak_worker(void *arg) {
    const soak_job_t *j = (const soak_job_t *)arg;
    const size_t n = j->frame_len, p = j->order;
    double *xw = (double *)malloc(n * sizeof(double));
    static const double male[4] = {320., 1440., 2760., 3200.};      /* MALE_FORMANT_ESTIMATES, src/lib.rs:27 */
    for (;;) {
        const unsigned long i = atomic_fetch_add(j->next, 1ul);
        if (i >= j->count) break;
        const double *fr = j->audio + (j->first + i) * j->hop;
        if (j->what & (VBXO_SOAK_PITCH | VBXO_SOAK_LPC | VBXO_SOAK_MFCC))
            for (size_t k = 0; k < n; k++) xw[k] = fr[k] * j->w_hann[k];       /* Windower::hanning */
        if (j->what & VBXO_SOAK_PITCH) {
            vbxo_pitch_t c[3] = {{0., 0.}, {0., 0.}, {0., 0.}};
            size_t cnt = 0;
            const int st = vbxo_pitch(xw, n, j->sample_rate, j->threshold, j->fmin, j->fmax, c, 3, &cnt);
            j->pitch_status[i] = st;
            j->pitch_count[i] = (st == VBXO_OK) ? (int32_t)cnt : 0;
            double *o = j->pitch_top + 6 * i;                                  /* the head of the sorted Vec */
            for (int k = 0; k < 3; k++) { o[2 * k] = c[k].frequency; o[2 * k + 1] = c[k].strength; }
        }
        if (j->what & VBXO_SOAK_LPC) {
            vbxo_autocorrelate(xw, n, j->r + i * (p + 1), p + 1);
            vbxo_lpc(j->r + i * (p + 1), p, j->a + i * (p + 1));
        }
        if (j->what & VBXO_SOAK_MFCC)
            j->mfcc_status[i] = vbxo_mfcc(xw, n, j->n_mfcc, j->mfcc_lo, j->mfcc_hi, j->sample_rate, j->mfcc + i * j->n_mfcc, 0);
        if (j->what & VBXO_SOAK_FORMANTS) {
            vbxo_resonance_t est[4], res[VBXO_MAX_RESONANCES];
            for (int e = 0; e < 4; e++) { est[e].frequency = male[e]; est[e].bandwidth = 1.0; }
            memset(res, 0, sizeof res);
            double *co = j->burg + i * p;
            memset(co, 0, p * sizeof(double));
            const int st = vbxo_find_formants(fr, n, j->sample_rate, p, est, 4, res, co);
            j->ff_status[i] = st;
            int32_t cnt = 0;
            if (st == VBXO_OK) for (int k = 0; k < VBXO_MAX_RESONANCES; k++) cnt += res[k].frequency != 0.0;
            else memset(res, 0, sizeof res);
            j->res_count[i] = cnt;
            memcpy(j->res + i * 2 * VBXO_MAX_RESONANCES, res, sizeof res);
        }
    }
    free(xw);
    return NULL;
}

/* Frames [first, first + count) of the hop-strided view of `audio` (n_samples >= (first + count - 1) * hop + frame_len).
 * Output arrays of parts not selected by `what` may be NULL.  Returns 0, or -1 on a bad argument. */
int vbxo_soak(const double *audio, size_t n_samples, size_t frame_len, size_t hop, size_t first, size_t count,
              size_t order, double sample_rate, int what, int n_threads,
              double threshold, double fmin, double fmax, size_t n_mfcc, double mfcc_lo, double mfcc_hi,
              int32_t *pitch_status, int32_t *pitch_count, double *pitch_top /* [count][3]{f, s} */,
              double *r /* [count][order + 1] */, double *a /* [count][order + 1] */,
              double *mfcc /* [count][n_mfcc] */, int32_t *mfcc_status,
              double *burg /* [count][order] */, int32_t *ff_status, double *res /* [count][32]{f, bw} */, int32_t *res_count) {
    if (!audio || frame_len < 4 || hop < 1 || count < 1 || order < 1 || order > 30 || n_threads < 1) return -1;
    if ((first + count - 1) * hop + frame_len > n_samples) return -1;
    if ((what & VBXO_SOAK_PITCH) && !(pitch_status && pitch_count && pitch_top)) return -1;
    if ((what & VBXO_SOAK_LPC) && !(r && a)) return -1;
    if ((what & VBXO_SOAK_MFCC) && !(mfcc && mfcc_status && n_mfcc >= 1)) return -1;
    if ((what & VBXO_SOAK_FORMANTS) && !(burg && ff_status && res && res_count)) return -1;
    double *w = (double *)malloc(frame_len * sizeof(double));
    vbxo_window_hanning(w, frame_len);
    atomic_ulong next;
    atomic_init(&next, 0ul);
    soak_job_t job = {audio, first, count, frame_len, hop, order, n_mfcc, sample_rate, threshold, fmin, fmax, mfcc_lo, mfcc_hi,
                      what, w, &next, pitch_status, pitch_count, mfcc_status, ff_status, res_count,
                      pitch_top, r, a, mfcc, burg, res};
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int k = 1; k < n_threads; k++) pthread_create(&th[k], NULL, soak_worker, &job);
    soak_worker(&job);
    for (int k = 1; k < n_threads; k++) pthread_join(th[k], NULL);
    free(th); free(w);
    return 0;
}

/* The sequential part of find_formants (src/lib.rs:114, FormantExtractor src/spectrum.rs:357-369): the estimates restart
 * from est_init at every seg_start entry; a frame whose status is not OK leaves them untouched (src/lib.rs:75 `?`). */
int vbxo_soak_track(const double *res /* [count][32]{f, bw} */, const int32_t *ff_status, size_t count,
                    const int64_t *seg_start, size_t n_seg, const double *est_init /* [n_est]{f, bw} */, size_t n_est,
                    double *out /* [count][n_est]{f, bw} */) {
    if (!res || !ff_status || !est_init || !out || n_est < 1 || n_est > VBXO_FORMANT_SLOTS) return -1;
    vbxo_resonance_t est[VBXO_FORMANT_SLOTS];
    size_t s = 0;
    for (size_t e = 0; e < n_est; e++) { est[e].frequency = est_init[2 * e]; est[e].bandwidth = est_init[2 * e + 1]; }
    for (size_t t = 0; t < count; t++) {
        while (seg_start && s < n_seg && (size_t)seg_start[s] <= t) {
            if ((size_t)seg_start[s] == t)
                for (size_t e = 0; e < n_est; e++) { est[e].frequency = est_init[2 * e]; est[e].bandwidth = est_init[2 * e + 1]; }
            s++;
        }
        if (ff_status[t] == VBXO_OK)
            vbxo_estimate_formants(est, n_est, (const vbxo_resonance_t *)(res + t * 2 * VBXO_MAX_RESONANCES), VBXO_MAX_RESONANCES);
        memcpy(out + t * 2 * n_est, est, n_est * sizeof(vbxo_resonance_t));
    }
    return 0;
}
