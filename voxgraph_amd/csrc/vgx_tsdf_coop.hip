// TSDF path, racing mode (the default): the launch of the cooperative kernel (vgx_tsdf_coop_kernel.h: what it does and
// why) in its two shipped forms -- counted (STATS) and uncounted; no event log (TRACE = false).
#include "vgx_tsdf_coop_kernel.h"

namespace vgx {

hipError_t launch_racing_scan(hipStream_t stream, const TsdfLayerDev& L, const TsdfIntegratorDev& I, const float T[7],
                              const float* d_points, const uint32_t* d_rgba, long long n, int freespace, bool stats,
                              int cloud_width) {
  return launch_racing_scan_t<false>(stream, L, I, T, d_points, d_rgba, n, freespace, stats, cloud_width);
}

}  // namespace vgx
