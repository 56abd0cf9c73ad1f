#include <algorithm>
// wide_probe.hip — times k_scan_wide (pvs_scan_wide.hpp) alone on a synthetic tiled corpus: back-to-back launches, sustained,
// no pass A / C, no dense fallback in between (bench.py's ablation builds answer garbage and spend the step in the dense path,
// which changes the power state the next scan runs in).  Candidate rate is set by bisection on the threshold.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I panoptikon_amd/csrc [-Dtemporary ablation edits] -o wide_probe tools/probe/wide_probe.hip
//   ./wide_probe [rows=10000000] [cands_per_query=1600] [launches=40]
#ifndef PROBE_RW
#define PROBE_RW 1  // 1: 256 queries per pass, 2: 128
#endif
#include "pvs_scan_wide.hpp"

#include <cmath>
#include <cstdio>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);          \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__global__ void k_fill_codes(uint32_t *p, size_t n_words, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {  // sum of four small uniforms: sigma ~ 23, like int8 codes of unit vectors at 768-d
            uint32_t h = (uint32_t)(i * 4 + b) * 2654435761u + seed;
            h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
            const int c = (int)(h & 31) + (int)((h >> 5) & 31) + (int)((h >> 10) & 31) + (int)((h >> 15) & 31) - 62;
            w |= (uint32_t)(c & 0xff) << (8 * b);
        }
        p[i] = w;
    }
}
__global__ void k_fill_aux(float *aux, size_t n_tiles, float norm) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n_tiles * PVS_AUX_REC; t += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(t % PVS_AUX_REC);
        aux[t] = k < 32 ? 1.0f / norm : (k < 34 ? norm : 0.f);
    }
}

int main(int argc, char **argv) {
    const uint64_t n_rows = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
    const double want = argc > 2 ? atof(argv[2]) : 1600.0;
    const int launches = argc > 3 ? atoi(argv[3]) : 40;
    constexpr int KS = 3;
    const uint32_t stride = KS * 256, batch = 256 / PROBE_RW, grid = 256;
    const uint32_t wg_rows = WideGeo<KS, PVS_WIDE_NQ, PROBE_RW>::TILE_ROWS;
    const uint64_t cap = (n_rows + 127) / 128 * 128 + 128 + 1200000;  // (+ room for the PROBE_EXTRA timing experiment)
    uint8_t *rows, *qmat;
    float *aux, *thr;
    QInfo *qinfo;
    uint2 *seg;
    uint32_t *seg_cnt;
    CK(hipMalloc(&rows, cap * stride));  // (hipDeviceMallocUncached: no difference, 1.72 ms both)
    CK(hipMalloc(&aux, cap / 32 * PVS_AUX_REC * 4));
    CK(hipMalloc(&qmat, batch * stride));
    CK(hipMalloc(&qinfo, batch * sizeof(QInfo)));
    CK(hipMalloc(&thr, batch * 4));
    CK(hipMalloc(&seg, (size_t)grid * (PVS_WIDE_SEG_PER_STREAM * PROBE_RW) * batch * PVS_WIDE_SEG_CAP * 8));
    CK(hipMalloc(&seg_cnt, (size_t)batch * grid * (PVS_WIDE_SEG_PER_STREAM * PROBE_RW) * 4));
    hipLaunchKernelGGL(k_fill_codes, dim3(4096), dim3(256), 0, 0, (uint32_t *)rows, cap * stride / 4, 1u);
    hipLaunchKernelGGL(k_fill_codes, dim3(64), dim3(256), 0, 0, (uint32_t *)qmat, (size_t)batch * stride / 4, 77u);
    const float norm = 23.0f * sqrtf(768.f);
    hipLaunchKernelGGL(k_fill_aux, dim3(1024), dim3(256), 0, 0, aux, cap / 32, norm);
    std::vector<QInfo> qi(batch);
    for (auto &q : qi) {
        memset(&q, 0, sizeof(q));
        q.bb = norm * norm;
        q.qn = norm;
        q.dscale = 1.0f;
    }
    CK(hipMemcpy(qinfo, qi.data(), batch * sizeof(QInfo), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());

    ScanK k;
    memset(&k, 0, sizeof(k));
    k.rows = rows;
    k.aux = aux;
    k.qmat = qmat;
    k.qinfo = qinfo;
    k.thr = thr;
    k.seg = seg;
    k.seg_cnt = seg_cnt;
    k.seg_queries = batch;
    k.seg_cap = PVS_WIDE_SEG_CAP;
    k.seg_stride = grid * (PVS_WIDE_SEG_PER_STREAM * PROBE_RW);
    k.n_rows = n_rows;
    k.stride = stride;
    k.n_wgtiles = (uint32_t)((n_rows + wg_rows - 1) / wg_rows);
    k.tile_step = 1;
    k.grid = grid;
    std::vector<uint32_t> cnt((size_t)batch * grid * (PVS_WIDE_SEG_PER_STREAM * PROBE_RW));
    auto launch = [&]() { CK((scan_wide_launch_one<KS, PROBE_RW, PVS_COSINE, 1>(k, 0))); };
    auto candidates = [&](float t) {  // average candidates per query at threshold t (cosine: pass iff -acc/|a| <= t)
        std::vector<float> h(batch, t);
        CK(hipMemcpy(thr, h.data(), batch * 4, hipMemcpyHostToDevice));
        launch();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cnt.data(), seg_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
        double tot = 0;
        for (uint32_t c : cnt) tot += c;
        return tot / batch;
    };
    // acc/|a| is ~ N(0, 23): the pass rate at -t is the upper tail; bisect t in [-200, 0]
    float lo = -200.f, hi = 0.f;
    double got = 0;
    for (int it = 0; it < 18; it++) {
        const float mid = 0.5f * (lo + hi);
        got = candidates(mid);
        if (got > want) hi = mid; else lo = mid;
    }
    got = candidates(hi);
    printf("rows %llu  threshold %.3f  candidates/query %.1f\n", (unsigned long long)n_rows, hi, got);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float last = 0, best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < launches; i++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        last = ms / launches;
        if (last < best) best = last;
    }
    {
        unsigned long long *dbg;
        CK(hipMalloc(&dbg, 256 * 8));
        k.dense_out = (float *)dbg;
        for (int i = 0; i < 20; i++) launch();
        CK(hipDeviceSynchronize());
        launch();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(256);
        CK(hipMemcpy(t.data(), dbg, 256 * 8, hipMemcpyDeviceToHost));
        unsigned long long mn = ~0ull, mx = 0;
        for (auto v : t) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        std::vector<unsigned long long> s2(t);
        std::sort(s2.begin(), s2.end());
        printf("end-time spread over 256 workgroups: %.1f us (p10 %.1f p50 %.1f p90 %.1f us before the last)\n", (mx - mn) / 100.0,
               (mx - s2[25]) / 100.0, (mx - s2[128]) / 100.0, (mx - s2[230]) / 100.0);
        double xcd[8] = {0};
        for (int i = 0; i < 256; i++) xcd[i % 8] += (mx - t[i]) / 100.0 / 32;
        printf("mean lead per XCD (us):");
        for (int x = 0; x < 8; x++) printf(" %.1f", xcd[x]);
        printf("\n");
        k.dense_out = nullptr;
    }
    const double ops = 2.0 * n_rows * 768.0 * batch;  // (batch = 256 / PROBE_RW)
    printf("k_scan_wide<3,cos,B>: %.4f ms/launch sustained (best group %.4f)  %.2f TB/s  %.2f POP/s\n", last, best,
           n_rows * 768.0 / (last * 1e-3) / 1e12, ops / (last * 1e-3) / 1e15);
    return 0;
}
