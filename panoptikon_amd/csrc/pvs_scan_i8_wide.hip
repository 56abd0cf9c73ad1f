// int8 instances of the 256-query filter scan (pvs_scan_wide.hpp): row pitch 256..1024 B (dim <= 1024).
#include "pvs_scan_wide.hpp"
hipError_t pvs_scan_dispatch_i8_wide(const ScanK &k, uint32_t kslabs, int metric, int mode, hipStream_t s) {
    switch (kslabs) {
#ifndef PVS_WIDE_ONLY_KS3  // (tools/sweep_wide.sh builds the 768-B instance alone)
        case 1: return scan_wide_launch<1>(k, metric, mode, s);
        case 2: return scan_wide_launch<2>(k, metric, mode, s);
        case 4: return scan_wide_launch<4>(k, metric, mode, s);
#endif
        case 3: return scan_wide_launch<3>(k, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
bool pvs_scan_wide_serves(uint32_t qgroups, uint32_t kslabs, int mode) { return (mode == 0 || mode == 1) && qgroups == 8 && kslabs >= 1 && kslabs <= 4; }
uint32_t pvs_scan_wide_rows(uint32_t kslabs) { return kslabs <= 3 ? 64u : 32u; }
