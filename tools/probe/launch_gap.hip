// launch_gap.hip — what a dependent kernel-to-kernel hand-over costs on this stack: a chain of N short kernels (each reads what the
// previous one wrote) as plain stream launches, as one hipGraph (stream capture), and with hipExtLaunchKernelGGL events bound to
// every dispatch.  hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_step(float *p, int spin) {
    float v = p[threadIdx.x & 63];
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
    p[threadIdx.x & 63] = v;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main() {
    float *d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 4, REPS = 2000;
    for (int spin : {0, 2000}) {
        auto run_plain = [&]() {
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_step, dim3(256), dim3(256), 0, s, d, spin);
        };
        // warm
        for (int r = 0; r < 50; r++) run_plain();
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REPS; r++) run_plain();
        CK(hipStreamSynchronize(s));
        double us_plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REPS * N);
        // events bound to every dispatch
        std::vector<hipEvent_t> ev(2 * N);
        for (auto &e : ev) CK(hipEventCreate(&e));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REPS; r++)
            for (int i = 0; i < N; i++) hipExtLaunchKernelGGL(k_step, dim3(256), dim3(256), 0, s, ev[2 * i], ev[2 * i + 1], 0, d, spin);
        CK(hipStreamSynchronize(s));
        double us_bound = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REPS * N);
        float kms = 0;
        CK(hipEventElapsedTime(&kms, ev[0], ev[1]));
        // markers around every kernel
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REPS; r++)
            for (int i = 0; i < N; i++) {
                hipEventRecord(ev[2 * i], s);
                hipLaunchKernelGGL(k_step, dim3(256), dim3(256), 0, s, d, spin);
                hipEventRecord(ev[2 * i + 1], s);
            }
        CK(hipStreamSynchronize(s));
        double us_mark = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REPS * N);
        // graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        run_plain();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 50; r++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REPS; r++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REPS * N);
        printf("spin %d (kernel %.1f us by its bound events): per kernel of a dependent chain of %d — plain %.2f us, bound events %.2f, markers %.2f, graph %.2f\n",
               spin, kms * 1e3, N, us_plain, us_bound, us_mark, us_graph);
    }
    return 0;
}
