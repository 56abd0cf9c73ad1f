// pvs_select.hip — page 1 of many dense distance columns at once, without sorting the columns.
//
// The dense path is the reference's algorithm in HBM: score every row exactly, order everything, LIMIT k
// (docs/vector-index-design.md:84-89).  Round 1 did that one query at a time with a full radix sort of N keys (1.7 ms per
// query at 10M rows), so a batch of tie-heavy queries — int8 L2 over real data ties massively, SURVEY.md §7 — cost 150x the
// filter scan.  Here the queries the filter scan hands back are scored together (int8: 128 per corpus pass on the matrix
// cores, k_scan MODE 2; floats: 4 per pass, k_dense_exact) into one [rows][queries] matrix and the page of every column
// comes from an exact radix SELECT over all columns at once:
//   4 histogram passes (8 bits each, MSB first) pin the k-th smallest key K of every column and how many of its ties the
//   page takes; ties are cut by ROW: per-block tie counts -> prefix -> the row below which exactly that many ties lie
//   (the page order is (distance, id) and ids increase with the row); one more pass emits the <= k selected (key, row)
//   pairs per column, a per-column bitonic sort in LDS orders them.
// Six streaming reads of the matrix, whatever the data looks like (all keys equal included).  Keys: f32_sort_key(distance);
// NULL distances 0xfffffffe (after every number, in row order like any tie); rows outside a candidate mask 0xffffffff and
// never selected.
#include "pvs_kernels.hpp"

namespace {
constexpr uint32_t SEL_CG = 32;       // columns per workgroup
constexpr uint32_t SEL_RB = 2048;     // rows per tie-count block
constexpr uint32_t SEL_KMAX = 8192;   // largest page the in-LDS sort takes

struct SelCol {                // per column, in HBM
    uint32_t prefix, mask;     // radix select state
    uint32_t kk;               // rank still to find inside the current prefix class (1-based)
    uint32_t ties;             // after the 4 passes: ties of K the page takes
    uint32_t rcut;             // ... = the ties with row < rcut
    uint32_t valid;            // rows that are candidates at all
    uint32_t cnt;              // emit cursor
    uint32_t pad;
};

// position r of the walk -> physical row: the rows are walked in tie order when a second sort key is set (tinv), in row order else
__device__ inline uint64_t sel_row(const uint32_t *tinv, uint64_t r) { return tinv ? (uint64_t)tinv[r] : r; }
__device__ inline uint32_t sel_key(float d, const uint8_t *mask, uint64_t row) {
    uint32_t k = f32_sort_key(d);
    if (k == 0xffffffffu) k = 0xfffffffeu;
    if (mask && !mask[row]) k = 0xffffffffu;
    return k;
}

__global__ void k_sel_init(SelCol *cols, uint32_t nq, uint32_t k, uint32_t *hist) {
    const uint32_t j = blockIdx.x;
    if (threadIdx.x == 0) {
        SelCol c;
        c.prefix = 0;
        c.mask = 0;
        c.kk = k;
        c.ties = 0;
        c.rcut = 0;
        c.valid = 0;
        c.cnt = 0;
        c.pad = 0;
        cols[j] = c;
    }
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) hist[(size_t)j * 256 + i] = 0;
    (void)nq;
}

// hist[col][digit] += rows of the column whose key matches the column's prefix so far (digit = the pass's 8 bits)
__global__ __launch_bounds__(256) void k_sel_hist(const float *m, uint64_t n, uint32_t ld, uint32_t nq, const uint8_t *mask, const uint32_t *tinv, const SelCol *cols,
                                                  int shift, uint32_t *hist, uint64_t rows_per_wg) {
    __shared__ uint32_t sh[SEL_CG * 256];
    __shared__ uint32_t s_prefix[SEL_CG], s_mask[SEL_CG];
    const uint32_t c0 = blockIdx.y * SEL_CG;
    const uint32_t nc = min(SEL_CG, nq - c0);
    for (uint32_t i = threadIdx.x; i < SEL_CG * 256; i += 256) sh[i] = 0;
    if (threadIdx.x < nc) {
        s_prefix[threadIdx.x] = cols[c0 + threadIdx.x].prefix;
        s_mask[threadIdx.x] = cols[c0 + threadIdx.x].mask;
    }
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_wg, r1 = min(n, r0 + rows_per_wg);
    // thread t handles column (t % 32) of rows r0 + t / 32, stepping 8 rows: a wave reads 2 rows x 128 contiguous bytes
    const uint32_t c = threadIdx.x & 31u;
    if (c < nc) {
        for (uint64_t r = r0 + (threadIdx.x >> 5); r < r1; r += 8) {
            const uint64_t pr = sel_row(tinv, r);
            const uint32_t key = sel_key(m[pr * ld + c0 + c], mask, pr);
            if ((key & s_mask[c]) == s_prefix[c]) atomicAdd(&sh[c * 256 + ((key >> shift) & 255u)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nc * 256; i += 256)
        if (sh[i]) atomicAdd(&hist[(size_t)(c0 + i / 256) * 256 + (i & 255u)], sh[i]);
}

// one workgroup per column: the digit that holds rank kk; narrows the prefix; clears the histogram for the next pass
__global__ __launch_bounds__(256) void k_sel_pick(SelCol *cols, uint32_t *hist, int shift, int last) {
    __shared__ uint32_t s[256];
    __shared__ uint32_t s_sel[2];
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    uint32_t *h = hist + (size_t)j * 256;
    uint32_t v = h[tid];
    s[tid] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t add = tid >= (uint32_t)off ? s[tid - off] : 0u;
        __syncthreads();
        v += add;
        s[tid] = v;
        __syncthreads();
    }
    SelCol c = cols[j];
    if (tid == 0) {
        s_sel[0] = 256;
        if (shift == 24) {  // first pass: rows that are candidates at all = everything below key 0xffffffff
            // (bin 255 may hold NULLs 0xfffffffe and excluded rows 0xffffffff; settled exactly in the last pass)
        }
    }
    __syncthreads();
    const uint32_t before = tid ? s[tid - 1] : 0u;
    if (before < c.kk && v >= c.kk) {
        s_sel[0] = tid;
        s_sel[1] = c.kk - before;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t b = s_sel[0];
        if (b >= 256) {  // fewer than kk keys in this class: the column has fewer than k rows — take everything
            c.prefix = 0xffffffffu;
            c.mask = 0xffffffffu;
            c.kk = 0;
        } else {
            c.kk = s_sel[1];
            c.prefix |= b << shift;
            c.mask |= 0xffu << shift;
        }
        if (last) {
            if (c.prefix == 0xffffffffu) {  // K = "excluded": no tie of it is ever taken
                c.ties = 0;
            } else {
                c.ties = c.kk;
            }
        }
        cols[j] = c;
    }
    __syncthreads();
    h[tid] = 0;
}

// ties of K per (row block, column)
__global__ __launch_bounds__(256) void k_sel_count_ties(const float *m, uint64_t n, uint32_t ld, uint32_t nq, const uint8_t *mask, const uint32_t *tinv, const SelCol *cols,
                                                        uint32_t *blockties, uint32_t n_blocks) {
    __shared__ uint32_t s_cnt[SEL_CG];
    __shared__ uint32_t s_K[SEL_CG];
    const uint32_t c0 = blockIdx.y * SEL_CG, nc = min(SEL_CG, nq - c0), b = blockIdx.x;
    if (threadIdx.x < SEL_CG) {
        s_cnt[threadIdx.x] = 0;
        s_K[threadIdx.x] = threadIdx.x < nc ? cols[c0 + threadIdx.x].prefix : 0u;
    }
    __syncthreads();
    const uint64_t r0 = (uint64_t)b * SEL_RB, r1 = min(n, r0 + SEL_RB);
    const uint32_t c = threadIdx.x & 31u;
    uint32_t mine = 0;
    if (c < nc)
        for (uint64_t r = r0 + (threadIdx.x >> 5); r < r1; r += 8) {
            const uint64_t pr = sel_row(tinv, r);
            mine += sel_key(m[pr * ld + c0 + c], mask, pr) == s_K[c];
        }
    if (mine) atomicAdd(&s_cnt[c], mine);
    __syncthreads();
    if (threadIdx.x < nc) blockties[(size_t)(c0 + threadIdx.x) * n_blocks + b] = s_cnt[threadIdx.x];
}

// one workgroup per column: the row below which exactly `ties` ties of K lie
__global__ __launch_bounds__(256) void k_sel_tie_cut(const float *m, uint64_t n, uint32_t ld, const uint8_t *mask, const uint32_t *tinv, SelCol *cols, const uint32_t *blockties,
                                                     uint32_t n_blocks) {
    __shared__ uint32_t s[256];
    __shared__ uint32_t s_blk, s_before;
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    SelCol c = cols[j];
    if (c.ties == 0) {
        if (tid == 0) {
            c.rcut = 0;
            cols[j] = c;
        }
        return;
    }
    const uint32_t *bt = blockties + (size_t)j * n_blocks;
    // block holding the ties-th tie: running sum over chunks of 256 blocks
    uint32_t carried = 0;
    if (tid == 0) s_blk = 0xffffffffu;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks && s_blk == 0xffffffffu; base += 256) {
        uint32_t v = base + tid < n_blocks ? bt[base + tid] : 0u;
        s[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t add = tid >= (uint32_t)off ? s[tid - off] : 0u;
            __syncthreads();
            v += add;
            s[tid] = v;
            __syncthreads();
        }
        const uint32_t before = carried + (tid ? s[tid - 1] : 0u);
        if (before < c.ties && carried + v >= c.ties && base + tid < n_blocks) {
            s_blk = base + tid;
            s_before = before;
        }
        __syncthreads();
        carried += s[255];
        __syncthreads();
    }
    const uint32_t blk = s_blk;
    if (blk == 0xffffffffu) {  // (cannot happen: the histogram counted these ties)
        if (tid == 0) {
            c.rcut = (uint32_t)n;
            cols[j] = c;
        }
        return;
    }
    // inside the block: the (ties - before)-th tie in row order; 256 threads x 8 consecutive rows each
    const uint32_t want = c.ties - s_before;
    const uint64_t r0 = (uint64_t)blk * SEL_RB;
    uint32_t flags = 0, cnt = 0;
    for (uint32_t i = 0; i < SEL_RB / 256; i++) {
        const uint64_t r = r0 + (uint64_t)tid * (SEL_RB / 256) + i;
        const uint64_t pr = r < n ? sel_row(tinv, r) : 0;
        const bool t = r < n && sel_key(m[pr * ld + j], mask, pr) == c.prefix;
        flags |= (uint32_t)t << i;
        cnt += t;
    }
    __syncthreads();
    uint32_t v = cnt;
    s[tid] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t add = tid >= (uint32_t)off ? s[tid - off] : 0u;
        __syncthreads();
        v += add;
        s[tid] = v;
        __syncthreads();
    }
    const uint32_t before = tid ? s[tid - 1] : 0u;
    if (before < want && v >= want) {
        uint32_t need = want - before;
        for (uint32_t i = 0; i < SEL_RB / 256; i++)
            if ((flags >> i) & 1u) {
                if (--need == 0) {
                    c.rcut = (uint32_t)(r0 + (uint64_t)tid * (SEL_RB / 256) + i + 1);
                    cols[j] = c;
                    break;
                }
            }
    }
}

// the selected (key, row) pairs of every column: key < K, or key == K and row < rcut
__global__ __launch_bounds__(256) void k_sel_emit(const float *m, uint64_t n, uint32_t ld, uint32_t nq, const uint8_t *mask, const uint32_t *tinv, SelCol *cols, uint32_t k,
                                                  unsigned long long *sel, uint64_t rows_per_wg) {
    __shared__ uint32_t s_K[SEL_CG], s_rcut[SEL_CG];
    const uint32_t c0 = blockIdx.y * SEL_CG, nc = min(SEL_CG, nq - c0);
    if (threadIdx.x < nc) {
        s_K[threadIdx.x] = cols[c0 + threadIdx.x].prefix;
        s_rcut[threadIdx.x] = cols[c0 + threadIdx.x].rcut;
    }
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_wg, r1 = min(n, r0 + rows_per_wg);
    const uint32_t c = threadIdx.x & 31u;
    if (c >= nc) return;
    for (uint64_t r = r0 + (threadIdx.x >> 5); r < r1; r += 8) {
        const uint64_t pr = sel_row(tinv, r);
        const uint32_t key = sel_key(m[pr * ld + c0 + c], mask, pr);
        if (key == 0xffffffffu) continue;
        if (key < s_K[c] || (key == s_K[c] && r < s_rcut[c])) {
            const uint32_t p = atomicAdd(&cols[c0 + c].cnt, 1u);
            if (p < k) sel[(size_t)(c0 + c) * k + p] = ((unsigned long long)key << 32) | (uint32_t)r;
        }
    }
}

// one workgroup per column: order the <= k selected pairs by (key, row) and write the page
__global__ __launch_bounds__(256) void k_sel_sort_emit(const unsigned long long *sel, const SelCol *cols, uint32_t k, const int64_t *ids, const uint32_t *tinv, const uint32_t *qmap,
                                                       int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    extern __shared__ unsigned long long s_sort[];
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    const uint32_t m = min(cols[j].cnt, k);
    uint32_t m2 = 1;
    while (m2 < m) m2 <<= 1;
    for (uint32_t i = tid; i < m2; i += 256) s_sort[i] = i < m ? sel[(size_t)j * k + i] : ~0ull;
    __syncthreads();
    for (uint32_t sz = 2; sz <= m2; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < m2 / 2; i += 256) {
                const uint32_t lo = 2 * i - (i & (st - 1)), hi = lo + st;
                const bool up = (lo & sz) == 0;
                const unsigned long long x = s_sort[lo], y = s_sort[hi];
                if ((x > y) == up) {
                    s_sort[lo] = y;
                    s_sort[hi] = x;
                }
            }
            __syncthreads();
        }
    const uint32_t q = qmap ? qmap[j] : j;
    int64_t *oi = out_ids + (size_t)q * k;
    float *od = out_dist + (size_t)q * k;
    for (uint32_t i = tid; i < k; i += 256) {
        if (i < m) {
            const unsigned long long v = s_sort[i];
            const uint32_t key = (uint32_t)(v >> 32);
            oi[i] = ids[sel_row(tinv, (uint32_t)v)];
            od[i] = key == 0xfffffffeu ? __builtin_nanf("") : f32_from_sort_key(key);
        } else {
            oi[i] = -1;
            od[i] = __builtin_nanf("");
        }
    }
    if (tid == 0) out_count[q] = m;
}
}  // namespace

bool pvs_select_supported(uint32_t k) { return k <= SEL_KMAX; }

// m: [n][ld] f32 distances (column j of query slot qmap[j]); writes page 1 of size k of every column.  Scratch comes
// from the cache (pvs_scratch_alloc); everything is enqueued on `s`, no host synchronisation.
pvs_status pvs_select_topk(const float *m, uint64_t n, uint32_t ld, uint32_t nq, uint32_t k, const uint8_t *mask, const int64_t *ids,
                           const uint32_t *d_qmap, int64_t *out_ids, float *out_dist, uint32_t *out_count, hipStream_t s, const uint32_t *tinv) {
    if (nq == 0) return PVS_OK;
    if (k > SEL_KMAX) return pvs_fail(PVS_ERR_UNSUPPORTED, "select: k too large");
    if (n >= 0xffffffffull) return pvs_fail(PVS_ERR_UNSUPPORTED, "select: too many rows");
    const uint32_t n_blocks = (uint32_t)((n + SEL_RB - 1) / SEL_RB);
    SelCol *cols = nullptr;
    uint32_t *hist = nullptr, *blockties = nullptr;
    unsigned long long *sel = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&cols, sizeof(SelCol) * nq));
        HIP_TRY(pvs_scratch_alloc((void **)&hist, (size_t)nq * 256 * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&blockties, (size_t)nq * std::max<uint32_t>(n_blocks, 1) * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&sel, (size_t)nq * k * 8));
        hipLaunchKernelGGL(k_sel_init, dim3(nq), dim3(256), 0, s, cols, nq, k, hist);
        const uint32_t cgs = (nq + SEL_CG - 1) / SEL_CG;
        const uint32_t row_wgs = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((n + 4095) / 4096, 1), 4096);
        const uint64_t rows_per_wg = (n + row_wgs - 1) / row_wgs;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            hipLaunchKernelGGL(k_sel_hist, dim3(row_wgs, cgs), dim3(256), 0, s, m, n, ld, nq, mask, tinv, cols, shift, hist, rows_per_wg);
            hipLaunchKernelGGL(k_sel_pick, dim3(nq), dim3(256), 0, s, cols, hist, shift, pass == 3 ? 1 : 0);
        }
        hipLaunchKernelGGL(k_sel_count_ties, dim3(n_blocks, cgs), dim3(256), 0, s, m, n, ld, nq, mask, tinv, cols, blockties, n_blocks);
        hipLaunchKernelGGL(k_sel_tie_cut, dim3(nq), dim3(256), 0, s, m, n, ld, mask, tinv, cols, blockties, n_blocks);
        hipLaunchKernelGGL(k_sel_emit, dim3(row_wgs, cgs), dim3(256), 0, s, m, n, ld, nq, mask, tinv, cols, k, sel, rows_per_wg);
        uint32_t m2 = 1;
        while (m2 < k) m2 <<= 1;
        hipLaunchKernelGGL(k_sel_sort_emit, dim3(nq), dim3(256), (size_t)m2 * 8, s, sel, cols, k, ids, tinv, d_qmap, out_ids, out_dist, out_count);
        HIP_TRY(hipGetLastError());
        return PVS_OK;
    };
    pvs_status st = body();
    // the kernels above are stream-ordered; the blocks go back to the cache once the stream has passed them
    if (hipStreamSynchronize(s) != hipSuccess && st == PVS_OK) st = pvs_fail(PVS_ERR_DEVICE, "select failed on device");
    pvs_scratch_free(cols);
    pvs_scratch_free(hist);
    pvs_scratch_free(blockties);
    pvs_scratch_free(sel);
    return st;
}
