// f16 instances of the filter-scan kernel: row pitch 2304 / 2560 / 3072 B (dim 1025..1536: SigLIP so400m 1152, bigG 1280, 1536).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_f16_xl(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
        case 9: return scan_launch_qg<PVS_F16, 9>(k, qg, metric, mode, s);
        case 10: return scan_launch_qg<PVS_F16, 10>(k, qg, metric, mode, s);
        case 12: return scan_launch_qg<PVS_F16, 12>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
