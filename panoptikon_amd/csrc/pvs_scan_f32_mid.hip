// f32 instances of the filter-scan kernel: row pitch 1536 / 2048 B (dim 257..512).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_f32_mid(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
        case 6: return scan_launch_qg<PVS_F32, 6>(k, qg, metric, mode, s);
        case 8: return scan_launch_qg<PVS_F32, 8>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
