// int8 instances of the filter-scan kernel: row pitch 1280 / 1536 / 2048 / 3072 B (dim 1025..3072: bigG-class CLIP,
// 1536-, 2048- and 3072-d text models).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_i8_large(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
        case 5: return scan_launch_qg<PVS_I8, 5>(k, qg, metric, mode, s);
        case 6: return scan_launch_qg<PVS_I8, 6>(k, qg, metric, mode, s);
        case 8: return scan_launch_qg<PVS_I8, 8>(k, qg, metric, mode, s);
        case 12: return scan_launch_qg<PVS_I8, 12>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
