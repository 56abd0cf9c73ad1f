// int8 instances of the filter-scan kernel: row pitch 256..1024 B (dim <= 1024).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_i8(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
#ifndef PVS_ONLY_KS3  // (tuning experiments build the 768-B instance alone)
        case 1: return scan_launch_qg<PVS_I8, 1>(k, qg, metric, mode, s);
        case 2: return scan_launch_qg<PVS_I8, 2>(k, qg, metric, mode, s);
        case 4: return scan_launch_qg<PVS_I8, 4>(k, qg, metric, mode, s);
#endif
        case 3: return scan_launch_qg<PVS_I8, 3>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
