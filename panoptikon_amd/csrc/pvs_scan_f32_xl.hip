// f32 instances of the filter-scan kernel: row pitch 4608 / 5120 / 6144 B (dim 1152, 1280, 1536).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_f32_xl(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
        case 18: return scan_launch_qg<PVS_F32, 18>(k, qg, metric, mode, s);
        case 20: return scan_launch_qg<PVS_F32, 20>(k, qg, metric, mode, s);
        case 24: return scan_launch_qg<PVS_F32, 24>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
