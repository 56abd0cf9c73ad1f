// pvs_direct_i8.hip — the i8 instances of k_direct_topk (pvs_direct_kernel.hpp), a translation unit of their own so that the
// element types compile in parallel.
#include "pvs_direct_kernel.hpp"

namespace pvs_direct {
hipError_t launch_i8(const DirectK &k, int metric, uint32_t nq_inst, uint32_t grid, hipStream_t s, hipEvent_t a, hipEvent_t b) {
    if (nq_inst == 1) return direct_metric<PVS_I8, 1>(k, metric, grid, s, a, b);
    if (nq_inst == 2) return direct_metric<PVS_I8, 2>(k, metric, grid, s, a, b);
    if (nq_inst == 4) return direct_metric<PVS_I8, 4>(k, metric, grid, s, a, b);
    if (nq_inst == 8) return direct_metric<PVS_I8, 8>(k, metric, grid, s, a, b);
    return hipErrorInvalidValue;
}
}  // namespace pvs_direct
