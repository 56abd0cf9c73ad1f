// int8 instances of the 256- and 128-query filter scans (pvs_scan_wide.hpp).
#include "pvs_scan_wide.hpp"
// qgroups = 8: 256 queries, row pitch 256..1024 B (dim <= 1024); qgroups = 4: 128 queries, row pitch 256..768 B
hipError_t pvs_scan_dispatch_i8_wide(const ScanK &k, uint32_t kslabs, uint32_t qgroups, int metric, int mode, hipStream_t s) {
    if (qgroups == 8) switch (kslabs) {
#ifndef PVS_WIDE_ONLY_KS3  // (tools/sweep_wide.sh builds the 768-B instances alone)
            case 1: return scan_wide_launch<1, 1>(k, metric, mode, s);
            case 2: return scan_wide_launch<2, 1>(k, metric, mode, s);
            case 4: return scan_wide_launch<4, 1>(k, metric, mode, s);
#endif
            case 3: return scan_wide_launch<3, 1>(k, metric, mode, s);
        }
    if (qgroups == 4) switch (kslabs) {
#ifndef PVS_WIDE_ONLY_KS3
            case 1: return scan_wide_launch<1, 2>(k, metric, mode, s);
            case 2: return scan_wide_launch<2, 2>(k, metric, mode, s);
#endif
            case 3: return scan_wide_launch<3, 2>(k, metric, mode, s);
        }
    return hipErrorInvalidValue;
}
bool pvs_scan_wide_serves(uint32_t qgroups, uint32_t kslabs, int mode) {
    if (mode != 0 && mode != 1) return false;
    return (qgroups == 8 && kslabs >= 1 && kslabs <= 4) || (qgroups == 4 && kslabs >= 1 && kslabs <= 3);
}
uint32_t pvs_scan_wide_rows(uint32_t kslabs) { return kslabs <= 3 ? 64u : 32u; }
uint32_t pvs_scan_wide_segs(uint32_t qgroups) { return PVS_WIDE_SEG_PER_STREAM * (8u / qgroups); }  // per workgroup stream: lane quarters x row blocks
