/* pcd_guard.h - C ABI of libpcd_guard.so: the resident-set watchdog.
 *
 * Test / tooling infrastructure of this repository, not part of the hot path
 * and with no counterpart in the reference: two GPU boxes were lost in round 3
 * to host allocations of this repository's own scripts (profiles/README.md).
 * A detached native thread (no Python GIL involved: a numpy / scipy call that
 * allocates without releasing the GIL cannot starve it) polls
 * /proc/self/statm every `interval_ms` and ends the process with
 * _exit(exit_code) once the resident set exceeds `limit_bytes`, after one
 * line on stderr.  Started at most once per process; a second call only
 * lowers the limit.  Returns 0, 1 (bad arguments) or 2 (no thread).
 * Python side: fenapack_amd/_guard.py.
 */
#ifndef PCD_GUARD_H
#define PCD_GUARD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int pcdg_watchdog_start(int64_t limit_bytes, int interval_ms, int exit_code);
/* highest resident set seen so far / the limit in force (0: not started) */
int64_t pcdg_watchdog_peak(void);
int64_t pcdg_watchdog_limit(void);

#ifdef __cplusplus
}
#endif
#endif
