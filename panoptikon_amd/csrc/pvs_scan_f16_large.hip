// f16 instances of the filter-scan kernel: row pitch 1536 / 2048 B (dim 513..1024).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_f16_large(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
        case 6: return scan_launch_qg<PVS_F16, 6>(k, qg, metric, mode, s);
        case 8: return scan_launch_qg<PVS_F16, 8>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
