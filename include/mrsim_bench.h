/*
 * mrsim_bench.h -- measurement and test aids of libmrsim.so.  NOT part of the product ABI (include/mrsim.h): nothing a
 * drop-in caller of the MR_env.step() path needs is declared here.  bench.py, tools/ and tests/ bind these; the symbols live in
 * the same shared library so that what is measured is the product's own launch path with events attached, not a copy of it.
 */
#ifndef MRSIM_BENCH_H
#define MRSIM_BENCH_H
#include "mrsim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Same launch bracketed by HIP events on `stream`; *kernel_ms_host = duration of the step
 * kernel alone (synchronises the stream; measurement aid for bench.py, not a product path). */
int mrsim_step_timed(const MrsimParams* p, int64_t n, uint32_t env_id0, const MrsimState* st,
                     const MrsimStepIO* io, uint64_t seed, uint64_t step_idx, void* stream,
                     float* kernel_ms_host);

/* mrsim_rollout with HIP events attached to the dispatch: *kernel_ms_host = kernel duration
 * (synchronises the stream; measurement aid for bench.py). */
int mrsim_rollout_timed(const MrsimParams* p, int64_t n, uint32_t env_id0, const MrsimState* st,
                        const MrsimRolloutIO* io, uint64_t seed, uint64_t step_idx0, void* stream,
                        float* kernel_ms_host);

/* Non-blocking measurement aids: the same launches with two caller-owned HIP events attached to the dispatch
 * (hipExtLaunchKernelGGL), nothing synchronised.  Lets bench.py read each kernel's duration INSIDE its timed
 * region, on the stream the kernel runs on, without perturbing it.  Events come from mrsim_event_create (thin
 * wrappers over hipEventCreate / hipEventElapsedTime / hipEventDestroy so a ctypes caller needs no second HIP
 * binding); mrsim_event_elapsed_ms synchronises on `stop`. */
int mrsim_event_create(void** event_out);
int mrsim_event_destroy(void* event);
int mrsim_event_elapsed_ms(void* start_event, void* stop_event, float* ms_host);
int mrsim_rollout_events(const MrsimParams* p, int64_t n, uint32_t env_id0, const MrsimState* st,
                         const MrsimRolloutIO* io, uint64_t seed, uint64_t step_idx0, void* stream,
                         void* start_event, void* stop_event);
int mrsim_step_events(const MrsimParams* p, int64_t n, uint32_t env_id0, const MrsimState* st,
                      const MrsimStepIO* io, uint64_t seed, uint64_t step_idx, void* stream,
                      void* start_event, void* stop_event);

/* Test aid: out[n][4] = the 4 standard normals of RNG call `c0` for envs env_id0..env_id0+n-1
 * (bit-compared with the oracle's definition in tests/). */
int mrsim_debug_normals(int64_t n, uint32_t env_id0, uint64_t seed, uint64_t step_idx, uint32_t c0,
                        int32_t noise_math, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRSIM_BENCH_H */
