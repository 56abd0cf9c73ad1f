// Calibration of rocprofv3's FETCH_SIZE for 64-byte sector reads (k_exact_wide's access pattern: four lanes read the four 16-byte
// chunks of one 64-byte sector, sectors 256 bytes apart).  Kernel A reads the FIRST 64-byte sector of every 256 bytes of a 2 GiB
// buffer, kernel B the SECOND (the other half of the same 128-byte line), kernel C both halves back to back, kernel D a plain
// coalesced 16 B/lane stream over the whole buffer.  Build: hipcc --offload-arch=gfx950 -O3 fetch_sector_calib.hip -o fetch_sector_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_sector(const uint4 *p, size_t n256, int half, uint4 *sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t seg = i >> 2;
    if (seg >= n256) return;
    const uint4 v = p[seg * 16 + (size_t)half * 4 + (i & 3)];
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[0] = v;
}
__global__ void k_both(const uint4 *p, size_t n256, uint4 *sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t seg = i >> 2;
    if (seg >= n256) return;
    const uint4 v = p[seg * 16 + (i & 3)], w = p[seg * 16 + 4 + (i & 3)];
    if (v.x == 0x12345678u && w.y == 0x9abcdef0u) sink[0] = v;
}
__global__ void k_stream(const uint4 *p, size_t n16, uint4 *sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    const uint4 v = p[i];
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[0] = v;
}
int main() {
    const size_t bytes = 2ull << 30, n256 = bytes / 256;
    uint4 *p, *sink;
    hipMalloc(&p, bytes);
    hipMalloc(&sink, 64);
    hipMemset(p, 1, bytes);
    const unsigned th = 256;
    hipLaunchKernelGGL(k_sector, dim3((unsigned)((n256 * 4 + th - 1) / th)), dim3(th), 0, 0, p, n256, 0, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_sector, dim3((unsigned)((n256 * 4 + th - 1) / th)), dim3(th), 0, 0, p, n256, 1, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_both, dim3((unsigned)((n256 * 4 + th - 1) / th)), dim3(th), 0, 0, p, n256, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_stream, dim3((unsigned)((bytes / 16 + th - 1) / th)), dim3(th), 0, 0, p, bytes / 16, sink);
    hipDeviceSynchronize();
    printf("buffer %zu bytes; A, B: %zu bytes requested each (64-byte sectors); C: %zu (128-byte lines); D: %zu (stream)\n", bytes, bytes / 4, bytes / 2, bytes);
    return 0;
}
