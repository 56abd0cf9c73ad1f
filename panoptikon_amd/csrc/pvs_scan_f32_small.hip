// f32 instances of the filter-scan kernel: row pitch 256..1024 B (dim <= 256).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_f32_small(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
        case 1: return scan_launch_qg<PVS_F32, 1>(k, qg, metric, mode, s);
        case 2: return scan_launch_qg<PVS_F32, 2>(k, qg, metric, mode, s);
        case 3: return scan_launch_qg<PVS_F32, 3>(k, qg, metric, mode, s);
        case 4: return scan_launch_qg<PVS_F32, 4>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
