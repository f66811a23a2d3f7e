// Diagnostics of the TSDF integrator for tests and bench.py (include/voxgraph_amd_bench.h; libvoxgraph_amd_bench.so):
// the walk statistics and per-workgroup stamps of a counted racing scan, the reproducible mode's speculation knobs, and
// the EVENT LOG of the racing kernel -- the TRACE instantiations of the one kernel template the product library ships
// (csrc/vgx_tsdf_coop_kernel.h), installed as the integrator's racing launcher on request.  tests/test_tsdf_replay_gpu.py
// replays such logs through oracle/tsdf_replay.c: every exchange on the two approximate sets, every ray's decisions,
// every per-voxel fold, bit for bit.
#include "vgx_tsdf_coop_kernel.h"
#include "voxgraph_amd_bench.h"

using namespace vgx;

namespace {
hipError_t launch_racing_scan_traced(hipStream_t stream, const TsdfLayerDev& L, const TsdfIntegratorDev& I, const float T[7],
                                     const float* d_points, const uint32_t* d_rgba, long long n, int freespace, bool stats,
                                     int cloud_width) {
  return launch_racing_scan_t<true>(stream, L, I, T, d_points, d_rgba, n, freespace, stats, cloud_width);
}
}  // namespace

extern "C" {

// test tooling (include/voxgraph_amd_bench.h): the reproducible mode's bounded speculation.  A scan whose complete
// walks exceed `threshold` steps is written out `depth` steps per ray at first and extended where a ray ran on;
// defaults 32 and 8 Mi.  Results do not depend on either (vgx_tsdf_det.hip); small values make small test scans
// go through the extension, the marks kept between scans and the warm second attempt.
int vgx_tsdf_integrator_set_speculation(vgx_tsdf_integrator I, int32_t depth, int64_t threshold) {
  if (!I || depth < 1 || threshold < 0 || threshold >= (1ll << 32)) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  I->det_cap = (uint32_t)depth;
  I->det_cap_threshold = (uint32_t)threshold;
  return VGX_OK;
}

// bench header: what the rays of the last COUNTED racing scan did (n_updates != NULL resets the statistics before the
// scan and makes the kernel gather them): see include/voxgraph_amd_bench.h for the seven numbers
int vgx_tsdf_integrator_walk_stats(vgx_tsdf_integrator I, int64_t stats[7]) {
  if (!I || !stats) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long u[kScanStatWords - 1] = {};
  VGX_HIP(ctx, hipMemcpyAsync(u, I->dev.n_updates + 1, sizeof(u), hipMemcpyDeviceToHost, ctx->tsdf_stream));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  for (int k = 0; k < kScanStatWords - 1; ++k) stats[k] = (int64_t)u[k];
  return VGX_OK;
}

// bench header: the rows the workgroups of the last counted racing scan left (stamps + work counts)
int vgx_tsdf_integrator_read_trace(vgx_tsdf_integrator I, int64_t* rows, int64_t max_workgroups, int64_t* n_workgroups,
                                   int64_t* clock_khz) {
  if (!I || !n_workgroups || max_workgroups < 0) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  *n_workgroups = I->wg_stats_rows;
  if (clock_khz) {
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device);
    *clock_khz = khz;
  }
  const long long take = I->wg_stats_rows < max_workgroups ? I->wg_stats_rows : max_workgroups;
  if (rows && take > 0) {
    VGX_HIP(ctx, hipMemcpyAsync(rows, I->d_wg_stats, (size_t)take * kWgStatWords * 8, hipMemcpyDeviceToHost, ctx->tsdf_stream));
    VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  }
  return VGX_OK;
}


// The racing kernel's event log: from now on every racing scan of this integrator runs the TRACE form of the kernel and
// appends to a log of `capacity_words` 64-bit words (0: back to the shipped launcher, log freed).  The log is emptied by
// vgx_tsdf_integrator_read_event_trace.
int vgx_tsdf_integrator_set_event_trace(vgx_tsdf_integrator I, int64_t capacity_words) {
  if (!I || capacity_words < 0) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  if (I->d_trace) (void)hipFree(I->d_trace);
  I->d_trace = nullptr;
  I->dev.trace = nullptr;
  I->dev.trace_words = 0;
  I->racing_launch = nullptr;
  if (capacity_words == 0) return VGX_OK;
  if (capacity_words < (int64_t)kTraceHeaderWords + 8)
    return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_integrator_set_event_trace: capacity below one event");
  VGX_HIP(ctx, hipMalloc(&I->d_trace, (size_t)capacity_words * 8));
  const unsigned long long header[2] = {kTraceHeaderWords, 0ull};
  VGX_HIP(ctx, hipMemcpy(I->d_trace, header, sizeof(header), hipMemcpyHostToDevice));
  I->dev.trace = I->d_trace;
  I->dev.trace_words = (unsigned long long)capacity_words;
  I->racing_launch = launch_racing_scan_traced;
  return VGX_OK;
}

// Copies the log out (words[0 .. *n_words), header stripped) and empties it; *lost = events that did not fit.
// max_words < the log's length: VGX_ERR_INVALID, nothing is emptied, *n_words says what is needed.
int vgx_tsdf_integrator_read_event_trace(vgx_tsdf_integrator I, uint64_t* words, int64_t max_words, int64_t* n_words,
                                         int64_t* lost) {
  if (!I || !n_words || max_words < 0) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  if (!I->d_trace) return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_integrator_read_event_trace: no event trace set");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  unsigned long long header[2] = {0, 0};
  VGX_HIP(ctx, hipMemcpy(header, I->d_trace, sizeof(header), hipMemcpyDeviceToHost));
  // (events that did not fit still moved the cursor)
  const unsigned long long used = header[0] < I->dev.trace_words ? header[0] : I->dev.trace_words;
  *n_words = (int64_t)(used - kTraceHeaderWords);
  if (lost) *lost = (int64_t)header[1];
  if (*n_words > max_words || (*n_words > 0 && !words)) return VGX_ERR_INVALID;
  if (*n_words > 0) VGX_HIP(ctx, hipMemcpy(words, I->d_trace + kTraceHeaderWords, (size_t)*n_words * 8, hipMemcpyDeviceToHost));
  const unsigned long long fresh[2] = {kTraceHeaderWords, 0ull};
  VGX_HIP(ctx, hipMemcpy(I->d_trace, fresh, sizeof(fresh), hipMemcpyHostToDevice));
  return VGX_OK;
}

// The integrator's two approximate sets as they are now (2^20 words each; either pointer may be NULL) and the offsets the
// NEXT scan's values will carry if it does not reset the sets (a scan adds one to each when it does:
// clear_checks_every_n_frames), plus the scans since the last reset.  state[0] start offset, [1] observed offset, [2] reset counter.
int vgx_tsdf_integrator_download_sets(vgx_tsdf_integrator I, uint64_t* start_set, uint64_t* observed_set, int64_t state[3]) {
  if (!I) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  const size_t bytes = ((size_t)1 << kSetBits) * 8;
  if (start_set) VGX_HIP(ctx, hipMemcpy(start_set, I->dev.start_set, bytes, hipMemcpyDeviceToHost));
  if (observed_set) VGX_HIP(ctx, hipMemcpy(observed_set, I->dev.observed_set, bytes, hipMemcpyDeviceToHost));
  if (state) {
    state[0] = (int64_t)I->dev.start_offset;
    state[1] = (int64_t)I->dev.observed_offset;
    state[2] = (int64_t)I->reset_counter;
  }
  return VGX_OK;
}

}  // extern "C"
