// mfma_rate.hip — what the int8 matrix pipe of gfx950 sustains, by instruction shape, independent chains, waves per SIMD and
// OPERAND DATA (constant vs random bytes): the clock the part holds under the load is part of the answer (s_memtime / wall).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/probe/mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4acc __attribute__((ext_vector_type(4)));

// SHAPE 0: 32x32x32 (16 acc regs), 1: 16x16x64 (4 acc regs).  CH chains per wave, NB distinct B operands cycled (register-resident),
// A operands: NA distinct.  data: constant (0) or random bytes loaded from memory (1).
template <int SHAPE, int CH, int NOPS, int PAT>
__global__ __launch_bounds__(512, 1) void k_rate(uint32_t iters, const v4i *data, int random, unsigned long long *cycles, uint32_t *sink) {
    v4i a[NOPS], b[NOPS];
#pragma unroll
    for (int i = 0; i < NOPS; i++) {
        if (random) {
            a[i] = data[(threadIdx.x + 64 * i) & 4095];
            b[i] = data[(threadIdx.x * 7 + 64 * i + 1111) & 4095];
        } else {
            a[i] = v4i{0x01010101, 0x01010101, 0x01010101, 0x01010101};
            b[i] = v4i{0x02020202, 0x02020202, 0x02020202, 0x02020202};
        }
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    int res = 0;
    if constexpr (SHAPE == 0) {
        v16i c[CH];
#pragma unroll
        for (int ch = 0; ch < CH; ch++) c[ch] = v16i{};
        for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < NOPS; i++)
#pragma unroll
                for (int ch = 0; ch < CH; ch++) c[ch] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[PAT == 2 ? i : (i + ch) % NOPS], b[PAT == 0 ? i : (i + 3 * ch + 1) % NOPS], c[ch], 0, 0, 0);
        }
#pragma unroll
        for (int ch = 0; ch < CH; ch++) res += c[ch][0] + c[ch][7];
    } else {
        v4acc c[CH];
#pragma unroll
        for (int ch = 0; ch < CH; ch++) c[ch] = v4acc{};
        for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < NOPS; i++)
#pragma unroll
                for (int ch = 0; ch < CH; ch++) c[ch] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[PAT == 2 ? i : (i + ch) % NOPS], b[PAT == 0 ? i : (i + 3 * ch + 1) % NOPS], c[ch], 0, 0, 0);
        }
#pragma unroll
        for (int ch = 0; ch < CH; ch++) res += c[ch][0] + c[ch][3];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    if (res == 0x12345678) sink[0] = res;
}

template <int SHAPE, int CH, int NOPS, int PAT = 0>
void run(const char *name, int threads, int random, const v4i *d_data, unsigned long long *d_cyc, uint32_t *d_sink) {
    const uint32_t iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<SHAPE, CH, NOPS, PAT>), dim3(256), dim3(threads), 0, 0, iters, d_data, random, d_cyc, d_sink);
    hipDeviceSynchronize();
    // sustained rate: launch back to back for >= 400 ms, report the average of the last 20 launches (the part's power
    // management averages over milliseconds: a 2-ms burst after idle runs at a clock a sustained load does not hold)
    float best = 1e9;
    unsigned long long cyc = 0;
    {
        float total = 0.f;
        int n = 0;
        while (total < 400.f && n < 2000) {
            hipEventRecord(e0);
            for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k_rate<SHAPE, CH, NOPS, PAT>), dim3(256), dim3(threads), 0, 0, iters, d_data, random, d_cyc, d_sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            total += ms;
            n++;
            best = ms / 10.f;  // the last group
        }
        hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    }
    const double waves = 256.0 * threads / 64.0;
    const double n_mfma = waves * iters * NOPS * CH;
    const double ops = n_mfma * (SHAPE == 0 ? 65536.0 : 32768.0);
    const double per_simd = n_mfma / 1024.0;
    printf("%-28s threads %3d data %-8s  %7.3f ms  %6.2f POP/s  clock %.2f GHz  %.1f cycles/MFMA/SIMD\n", name, threads, random == 0 ? "constant" : random == 1 ? "random" : "codes", best,
           ops / (best * 1e-3) / 1e15, (double)cyc / (best * 1e-3) / 1e9, (double)cyc / per_simd);
}

int main() {
    std::vector<v4i> h(4096);
    srand(7);
    for (auto &v : h)
        for (int i = 0; i < 4; i++) v[i] = (int)((unsigned)rand() * 2654435761u) ^ (rand() << 16);
    unsigned long long *d_cyc;
    uint32_t *d_sink;
    std::vector<v4i> h2(4096);  // small-magnitude codes (sum of four uniform bytes, sigma ~ 23): what a quantized unit-vector corpus looks like
    for (auto &v : h2)
        for (int i = 0; i < 4; i++) {
            unsigned w = 0;
            for (int b = 0; b < 4; b++) {
                int c = (rand() % 41 - 20) + (rand() % 41 - 20) + (rand() % 41 - 20) + (rand() % 21 - 10);
                w |= (unsigned)(c & 0xff) << (8 * b);
            }
            v[i] = (int)w;
        }
    v4i *d_data, *d_data2;
    hipMalloc(&d_data, 4096 * sizeof(v4i));
    hipMemcpy(d_data, h.data(), 4096 * sizeof(v4i), hipMemcpyHostToDevice);
    hipMalloc(&d_data2, 4096 * sizeof(v4i));
    hipMemcpy(d_data2, h2.data(), 4096 * sizeof(v4i), hipMemcpyHostToDevice);
    hipMalloc(&d_cyc, 8);
    hipMalloc(&d_sink, 4);
    for (int threads : {512}) {
        const int random = 2;
        run<0, 2, 24, 0>("32x32 2ch shareB", threads, random, d_data2, d_cyc, d_sink);
        run<0, 2, 24, 1>("32x32 2ch none", threads, random, d_data2, d_cyc, d_sink);
        run<0, 2, 24, 2>("32x32 2ch shareA", threads, random, d_data2, d_cyc, d_sink);
        run<1, 8, 24, 0>("16x16 8ch shareB", threads, random, d_data2, d_cyc, d_sink);
        run<1, 8, 24, 1>("16x16 8ch none", threads, random, d_data2, d_cyc, d_sink);
        run<1, 8, 24, 2>("16x16 8ch shareA", threads, random, d_data2, d_cyc, d_sink);
        run<1, 2, 24, 0>("16x16 2ch shareB", threads, random, d_data2, d_cyc, d_sink);
        run<1, 2, 24, 2>("16x16 2ch shareA", threads, random, d_data2, d_cyc, d_sink);
    }
    return 0;
}
