/*
 * vsloss_debug.h - diagnostic entry points of libvsloss.so.
 *
 * Not part of the drop-in boundary (include/vsloss.h): nothing here replaces a function of the reference.  The
 * library's own tests and its benchmark use them to look inside the fused launch.  They may change without a
 * version bump.
 */
#ifndef VSLOSS_DEBUG_H
#define VSLOSS_DEBUG_H

#include "vsloss.h"

#ifdef __cplusplus
extern "C" {
#endif

/* q[i] = a[i] / b[i] through the shared-divisor division the fused kernel uses for its sampling coordinates
 * (device arrays of n floats); the tests compare it bit for bit with IEEE float32 division. */
int vsl_debug_quot(const float* a, const float* b, float* q, long long n, void* stream);

/* Workgroups of the fused kernel the runtime expects to fit on one CU for S sources. */
int vsl_debug_fused_occupancy(int S, int avg_reprojection);

/* Runs vsl_loss_forward_heads `reps` times and returns in *avg_ms (host) the average duration of the fused tile
 * kernel alone, from HIP events recorded on `stream` directly around that launch.  The calls are enqueued back to
 * back, every launch between an event pair of its own, and the host waits once at the end (1 <= reps <= 4096).
 * Creates and destroys 2 * reps events (the one place where the library touches an allocating runtime call). */
int vsl_debug_time_fused(const vsl_loss_desc* desc, const vsl_head* heads, int n_heads, int reps, float* avg_ms,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VSLOSS_DEBUG_H */
