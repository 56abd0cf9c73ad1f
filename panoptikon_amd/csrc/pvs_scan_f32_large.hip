// f32 instances of the filter-scan kernel: row pitch 3072 / 4096 B (dim 513..1024).
#include "pvs_scan_kernel.hpp"
#include "pvs_scan_dispatch.hpp"
hipError_t pvs_scan_dispatch_f32_large(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
        case 12: return scan_launch_qg<PVS_F32, 12>(k, qg, metric, mode, s);
        case 16: return scan_launch_qg<PVS_F32, 16>(k, qg, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
