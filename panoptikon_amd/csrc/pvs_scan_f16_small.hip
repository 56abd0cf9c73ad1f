// f16 instances of the filter-scan kernel: row pitch 256..1024 B (dim <= 512).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_f16_small(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
        case 1: return scan_launch_qg<PVS_F16, 1>(k, qg, metric, mode, s);
        case 2: return scan_launch_qg<PVS_F16, 2>(k, qg, metric, mode, s);
        case 3: return scan_launch_qg<PVS_F16, 3>(k, qg, metric, mode, s);
        case 4: return scan_launch_qg<PVS_F16, 4>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
