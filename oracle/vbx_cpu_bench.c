/*
 * vbx_cpu_bench.c -- native timing harness of the CPU oracle: bench.py's `cpu_baseline` leg.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see vbx_oracle.h).  The reference crate is single-threaded Rust whose
 * user loop calls the traits once per frame (examples/pitch_detection.rs:23-30, tests/lib.rs:71-83); it cannot be
 * built here, so the timed baseline is the oracle ("port").  This harness runs that frame loop natively: pthreads
 * over frames, no Python, no lock on the hot path (one atomic fetch-add per frame), window tables built once,
 * per-thread output buffers, and reports frames per second on 1 thread or on n threads.
 *
 *   workload 0  pipeline   hanning frame -> pitch (every candidate refined, as the reference does) + autocorrelate(p+1)
 *                          + lpc(p) + mfcc(13, (100, 8000)); rectangle frame -> find_formants(p) (state carried per thread)
 *   workload 1  config 2   hanning frame -> autocorrelate(p+1) -> lpc(p)
 *   workload 2  config 3   hanning frame -> pitch
 *   workload 3  config 4   rectangle frame -> find_formants(p)
 * Frame t is audio[t*hop .. t*hop+frame_len) (the Windower view); frames are visited in a scrambled order
 * (t = i*step mod n_frames) so that a bounded sample covers the voiced and the unvoiced stretches of the signal.
 */
#define _GNU_SOURCE
#include "vbx_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int workload;
    const double *audio;
    size_t n_frames, frame_len, hop, order, step;
    double sample_rate, seconds;
    const double *w_hann;
    atomic_ulong *next;
    unsigned long max_frames;
    unsigned long done;       /* out */
    double checksum;          /* out: keeps the optimiser honest */
} job_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    const size_t n = j->frame_len, p = j->order;
    double *xw = (double *)malloc(n * sizeof(double));
    double r[64], a[64], mf[64];
    vbxo_pitch_t cand[4];
    vbxo_resonance_t est[4];
    static const double male[4] = {320., 1440., 2760., 3200.};      /* MALE_FORMANT_ESTIMATES, src/lib.rs:27 */
    for (int e = 0; e < 4; e++) { est[e].frequency = male[e]; est[e].bandwidth = 1.0; }
    const double t_end = now_s() + j->seconds;
    double acc = 0.0;
    unsigned long done = 0;
    for (;;) {
        const unsigned long i = atomic_fetch_add(j->next, 1ul);
        if (i >= j->max_frames) break;
        const size_t t = (size_t)((i * (unsigned long)j->step) % (unsigned long)j->n_frames);
        const double *fr = j->audio + t * j->hop;
        if (j->workload != 3) for (size_t k = 0; k < n; k++) xw[k] = fr[k] * j->w_hann[k];   /* Windower::hanning */
        if (j->workload == 0 || j->workload == 2) {
            size_t count = 0;
            vbxo_pitch(xw, n, j->sample_rate, 0.2, 75.0, 600.0, cand, 4, &count);
            acc += cand[0].frequency + (double)count;
        }
        if (j->workload == 0 || j->workload == 1) {
            vbxo_autocorrelate(xw, n, r, p + 1);
            vbxo_lpc(r, p, a);
            acc += a[p];
        }
        if (j->workload == 0 || j->workload == 3) {
            vbxo_find_formants(fr, n, j->sample_rate, p, est, 4, NULL, NULL);
            acc += est[0].frequency;
        }
        if (j->workload == 0) {
            vbxo_mfcc(xw, n, 13, 100.0, 8000.0, j->sample_rate, mf, 1);
            acc += mf[0];
        }
        done++;
        if ((done & 7ul) == 0 || j->workload == 0 || j->workload == 2) { if (now_s() >= t_end) break; }
    }
    free(xw);
    j->done = done;
    j->checksum = acc;
    return NULL;
}

/* Runs `workload` on n_threads threads for about `seconds` (or until max_frames frames are done).
 * Returns 0; *frames_done = frames completed, *elapsed_s = wall time of the parallel region. */
int vbxo_cpu_bench(int workload, const double *audio, size_t n_samples, size_t frame_len, size_t hop, size_t order,
                   double sample_rate, int n_threads, double seconds, unsigned long max_frames,
                   unsigned long *frames_done, double *elapsed_s, double *checksum) {
    if (!audio || frame_len < 4 || hop < 1 || n_samples < frame_len || order < 1 || order > 30 || n_threads < 1 ||
        workload < 0 || workload > 3)
        return -1;
    const size_t n_frames = (n_samples - frame_len) / hop + 1;
    double *w = (double *)malloc(frame_len * sizeof(double));
    vbxo_window_hanning(w, frame_len);
    size_t step = 37;                                     /* scrambled visiting order, coprime with n_frames */
    while (n_frames > 1 && (n_frames % step == 0 || step % 2 == 0)) step++;
    for (;;) {                                            /* gcd(step, n_frames) == 1 */
        size_t a = step, b = n_frames;
        while (b) { size_t t = a % b; a = b; b = t; }
        if (a == 1) break;
        step++;
    }
    atomic_ulong next;
    atomic_init(&next, 0ul);
    job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int k = 0; k < n_threads; k++) {
        jobs[k].workload = workload; jobs[k].audio = audio; jobs[k].n_frames = n_frames; jobs[k].frame_len = frame_len;
        jobs[k].hop = hop; jobs[k].order = order; jobs[k].step = step; jobs[k].sample_rate = sample_rate;
        jobs[k].seconds = seconds; jobs[k].w_hann = w; jobs[k].next = &next;
        jobs[k].max_frames = max_frames ? max_frames : ~0ul;
    }
    const double t0 = now_s();
    for (int k = 1; k < n_threads; k++) pthread_create(&th[k], NULL, worker, &jobs[k]);
    worker(&jobs[0]);
    for (int k = 1; k < n_threads; k++) pthread_join(th[k], NULL);
    const double dt = now_s() - t0;
    unsigned long total = 0; double cs = 0.0;
    for (int k = 0; k < n_threads; k++) { total += jobs[k].done; cs += jobs[k].checksum; }
    if (frames_done) *frames_done = total;
    if (elapsed_s) *elapsed_s = dt;
    if (checksum) *checksum = cs;
    free(jobs); free(th); free(w);
    return 0;
}
