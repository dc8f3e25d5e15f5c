/*
 * probes.h - measurement probes of the probe builds (libvallex_probes.so / libvallex_stamps.so, `build.py --probes|--stamps`).
 * NOT part of the product ABI (include/vallex.h) and not compiled into libvallex.so.
 */
#ifndef VALLEX_PROBES_H
#define VALLEX_PROBES_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid for DESIGN.md's launch-boundary budget: us per kernel of a dependent chain of
 * n trivial kernels, out[0] replayed as a hipGraph, out[1] launched eagerly. */
int vx_debug_launch_floor(int32_t n_kernels, int32_t grid, int32_t block, int32_t iters, double* out);
/* Probe for a single-launch decode step: `stages` dependent 1024-wide GEMV stages in ONE kernel, separated by a
 * device-wide barrier (mode 0 = barrier only, 1 = release/acquire fences, 2 = agent-scope loads/stores + counter, 3 = as 2 with the barrier among groups of 8 consecutive workgroups only, 4 = among the workgroups with equal blockIdx % 8: timing, results unchecked).
 * rows in {4,12,16} = output rows per workgroup per stage.  out[0] us/launch, out[1] us/stage, out[2] max |err|
 * against a host evaluation of the same chain, out[3] != 0 when a bounded spin ran out. */
int vx_debug_stage_chain(int32_t nwg, int32_t stages, int32_t rows, int32_t mode, int32_t iters, double* out);
/* Probe: L2 -> CU fill rate.  `grid` workgroups of `threads` lanes stream one shared region of `region_bytes` with `unroll`
 * independent 16-byte loads in flight per lane.  out[0] GB/s chip-wide, out[1] bytes/clock per busy CU, out[2] clock (GHz). */
int vx_debug_l2_fill(int32_t grid, int32_t threads, int32_t unroll, int64_t region_bytes, int32_t iters, double* out);
/* Probe builds only (csrc/build.py --stamps -> libvallex_stamps.so): phase timestamps (10 ns ticks) that workgroup 0 of the
 * last stamped kernel recorded at its VX_STAMP points.  The product library returns VX_ERR_UNSUPPORTED. */
int vx_debug_read_stamps(unsigned long long* out, int32_t n);
/* Stamps build only: per-kernel times (s_memrealtime, 100 MHz) of the AR decode step inside the hipGraph replay, recorded by one
 * extra, otherwise idle workgroup per kernel.  dst == NULL arms (allocates + zeroes) the ring, otherwise the ring is copied to dst:
 * [16 passes][64 kernel ids] uint64; a pass lands in slot pass % 16. */
int vx_debug_kstamps(void* dst, int64_t nbytes, int32_t* dims);

/* Stamps build only: phase times of the sharded decode step's two launches, [2 kinds][16 layers][256 workgroups][16] uint64 (100 MHz);
 * dst == NULL arms. */
int vx_debug_fqstamps(void* dst, int64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
