/*
 * voxgraph_amd_bench.h -- benchmark / test tooling, exported by libvoxgraph_amd_bench.so (csrc/bench/; it links
 * the product library libvoxgraph_amd.so, which carries none of this): synthetic scenes generated on the device so that
 * bench.py and the tests can fill 200 submaps of 256^3 voxels without a host round trip, memory and atomic ceilings for
 * the rooflines, and the TSDF integrator's diagnostics -- among them the racing kernel's EVENT LOG, which
 * tests/test_tsdf_replay_gpu.py replays through the oracle.  Nothing in the reference corresponds to these entry
 * points and an integration never calls them.
 */
#ifndef VOXGRAPH_AMD_BENCH_H_
#define VOXGRAPH_AMD_BENCH_H_

#include "voxgraph_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Fills a dense
 * block_dims[0..2] cube of blocks starting at block_min with the analytic
 * "city" scene of oracle/synth.py (ground plane + one box building per 25.6 m
 * cell, seeded), sampled at the submap's true pose {x,y,z,yaw}, entirely on the
 * device: TSDF = clamp(d, +-truncation) with weight tsdf_weight where
 * |d| <= 2*truncation else 0; ESDF = clamp(d, +-esdf_max), observed iff
 * |d| <= esdf_max.  Builds the ESDF sampling grid (and the TSDF one when
 * build_tsdf_grid != 0). */
VGX_API int vgx_synth_city_submap(vgx_ctx ctx, int32_t submap_id, float voxel_size,
                                  int32_t voxels_per_side, const int32_t block_min[3],
                                  const int32_t block_dims[3], float truncation,
                                  float esdf_max, float tsdf_weight,
                                  const double true_pose[4], uint32_t seed,
                                  int32_t build_tsdf_grid, vgx_submap* out);

/* One OS1-like LiDAR scan (n_el rings x n_az azimuth steps, elevation
 * -el_span/2 .. +el_span/2 rad) of the same analytic city scene, sphere-traced on the
 * device from sensor pose {x,y,z,yaw} (world frame).  Writes n_el*n_az points in the
 * SENSOR frame (float3, DEVICE pointer); rays that hit nothing within max_range get
 * length max_range * 2 (so they become clearing rays / are dropped like real returns). */
VGX_API int vgx_synth_city_scan(vgx_ctx ctx, const double sensor_pose[4], int32_t n_az,
                                int32_t n_el, float el_span, float max_range, uint32_t seed,
                                void* d_points_C);

/* Latency model of the racing TSDF kernel (bench.py `tsdf.*.roofline`): ns per step of a chain of
 * DEPENDENT device-scope 64-bit exchanges on pseudo-random words of a table of table_bytes (a power of two; an
 * approximate hash set is 8 MiB), `waves` wavefronts of 64 such chains running side by side (1 = unloaded). */
VGX_API int vgx_bench_atomic_roundtrip(vgx_ctx ctx, int64_t table_bytes, int32_t waves, int32_t chain,
                                       float* ns_per_step);

/* Device memory whose physical pages are deliberately out of order: an address range reserved through the virtual-memory
 * API, one physical allocation per chunk_bytes (rounded up to the allocation granularity), mapped in a shuffled order
 * (seed 0: in order).  For the placement experiments of profiles/r05_points_placement.txt.  Freed with
 * vgx_bench_free_scattered only. */
VGX_API int vgx_bench_alloc_scattered(vgx_ctx ctx, int64_t bytes, int64_t chunk_bytes, uint32_t seed, void** d_ptr);
VGX_API int vgx_bench_free_scattered(vgx_ctx ctx, void* d_ptr);

/* Same-run memory ceiling for the REG rooflines (bench.py `roofline.copy_ceiling_GBs`): `launches` launches that stream
 * read_bytes in (float4 loads from d_src) and write_bytes out (non-temporal float4 stores to d_dst) -- both DEVICE
 * pointers, sizes multiples of 16; read_bytes == write_bytes is a float4 copy, read_bytes == 0 a fill.  Returns the
 * average duration of one launch (HIP events on the context's stream). */
VGX_API int vgx_bench_stream_ceiling(vgx_ctx ctx, const void* d_src, int64_t read_bytes, void* d_dst,
                                     int64_t write_bytes, int32_t launches, float* ms_per_launch);

/* What the rays of the last COUNTED racing scan did (vgx_tsdf_integrate[_device] with n_updates != NULL resets the
 * statistics before the scan and makes the kernel gather them; an uncounted scan gathers nothing):
 *   stats[0]  the longest chain of DEPENDENT round trips to the observed set any ray needed (rounds of the cooperative
 *             walk; with VGX_TSDF_KERNEL=v1: voxel steps, one exchange each)
 *   stats[1]  exchanges on the observed set, all rays together       stats[2]  voxel updates that also blended a colour
 *   stats[3]  peeks (plain loads of an observed-set slot ahead of the exchanges)
 *   stats[4]  per-voxel folds (one block lookup + one {distance, weight} load + one compare-and-swap each, + colour)
 *   stats[5]  compare-and-swaps that found another workgroup's update and were folded again
 *   stats[6]  exchanges issued behind a ray's stopping step because a peeked slot changed before the exchange */
VGX_API int vgx_tsdf_integrator_walk_stats(vgx_tsdf_integrator integrator, int64_t stats[7]);

/* Where the racing kernel's time goes: every COUNTED racing scan leaves one row of 16 words per workgroup (256 points):
 * four wall_clock64 stamps -- start, rays queued (phase 1 done), walk done, end -- then its rays, rounds, per-voxel folds
 * and longest chain of repeated folds, then the eight sums behind vgx_tsdf_integrator_walk_stats.  rows[workgroup][16];
 * *clock_khz = the counter's rate. */
VGX_API int vgx_tsdf_integrator_read_trace(vgx_tsdf_integrator integrator, int64_t* rows, int64_t max_workgroups,
                                           int64_t* n_workgroups, int64_t* clock_khz);

/* Reproducible TSDF mode, bounded speculation (csrc/vgx_tsdf_det.hip): a scan whose rays' complete walks are more
 * than `threshold` voxel steps is written out `depth` steps per ray at first; rays that ran on are extended in a
 * further attempt.  Defaults 32 and 8 Mi.  The integrated layer does NOT depend on either value -- the tests use
 * small ones so that small scans exercise the extension logic (tests/test_tsdf_deterministic_gpu.py). */
VGX_API int vgx_tsdf_integrator_set_speculation(vgx_tsdf_integrator integrator, int32_t depth, int64_t threshold);

/* ---- the racing TSDF kernel's event log ----------------------------------------------------------------------------
 * After vgx_tsdf_integrator_set_event_trace(integrator, capacity_words > 0) every racing scan of the integrator runs the
 * logging instantiation of the shipped kernel (the same template, csrc/vgx_tsdf_coop_kernel.h: same decisions, other
 * timing) and appends events to a log of capacity_words 64-bit words; 0 restores the shipped launcher and frees the log.
 * An event is a run of words, the kind in the low byte of the first:
 *   1  start-set exchange   {1, point, value written, value returned}
 *   2  start-set skip       {2, point, value, the point of the lane to the left}   (same value as the left neighbour:
 *                            no exchange -- it would have found its own value)
 *   3  a cast ray           {3 | bad << 8, point, voxels its walk visits, 0}       (bad: NaN ray ends, never walked)
 *   4  observed-set exchange {4 | step << 8, point, value written, value returned}  (step: 0 = the walk's first voxel)
 *   5  a per-voxel fold     {5 | n << 8 | flags << 40, voxel key, word folded over, word published, colour folded over |
 *                            colour left << 32, voxel's index in the pool, then n records: point | step << 32, in the
 *                            order updateTsdfVoxel was applied}      voxel key: 21 bits per axis biased by 2^20, x highest;
 *                            word: distance bits | weight bits << 32; flags: 1 published (a compare-and-swap from the
 *                            one word to the other succeeded; else every record left the voxel alone and the two words
 *                            are equal), 2 the colour was written (compare-and-swap likewise), 4 some record blended
 * Peeks (plain loads that only choose which exchanges to issue) are not events. */
VGX_API int vgx_tsdf_integrator_set_event_trace(vgx_tsdf_integrator integrator, int64_t capacity_words);
/* Copies the log out (header stripped) and empties it.  *lost: events that did not fit.  A log longer than max_words:
 * VGX_ERR_INVALID with *n_words = what is needed, nothing emptied. */
VGX_API int vgx_tsdf_integrator_read_event_trace(vgx_tsdf_integrator integrator, uint64_t* words, int64_t max_words,
                                                 int64_t* n_words, int64_t* lost);
/* The two approximate sets as they are now (2^20 words each; either may be NULL); state[0], [1]: the offsets the last
 * scan's values carried, state[2]: scans since the sets were last reset (clear_checks_every_n_frames). */
VGX_API int vgx_tsdf_integrator_download_sets(vgx_tsdf_integrator integrator, uint64_t* start_set, uint64_t* observed_set,
                                              int64_t state[3]);

#ifdef __cplusplus
}
#endif
#endif /* VOXGRAPH_AMD_BENCH_H_ */
