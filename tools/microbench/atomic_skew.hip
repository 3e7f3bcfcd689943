// Does the memory-side atomic path slow down when the TARGET ROWS are skewed (Zipf positives beyond the LDS-accumulated head)?
// 16-lane groups add to the four 64-byte segments + the bias line of a row whose index is Zipf(1)-distributed over `n_items` with the
// `skip` most popular ranks left out (the engine keeps those in LDS).  Timing experiment for profiles/r05_notes.md; not product code.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_skew.hip -o atomic_skew
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// rank ~ Zipf(1) over [skip, n): P(rank <= r) = ln((r+1)/(skip+1)) / ln((n+1)/(skip+1))  (continuous approximation)
__device__ __forceinline__ uint32_t zipf_rank(uint32_t h, float lo, float span, uint32_t n) {
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    uint32_t r = (uint32_t)(__expf(lo + u * span)) - 1u;
    return r < n ? r : n - 1;
}

// frac_zipf_x256: share of the row updates that go to Zipf rows (the rest uniform), in 1/256
// SPLIT: segment k of a row lives in table k (a row's four segments are 64-byte lines far apart instead of 256 contiguous bytes)
template <bool LOADS, bool SPLIT>
__global__ void __launch_bounds__(1024) skew_kernel(float *table, float *bias, uint32_t n_items, uint32_t skip, int frac_zipf_x256, int iters, float *out) {
    const uint32_t lane = threadIdx.x & 63, sub = lane & 15;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const float lo = __logf((float)skip + 1.0f), span = __logf((float)n_items + 1.0f) - lo;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const uint32_t h = mix32(group * 7919u + it * 104729u + 1u);
        const bool z = (int)(mix32(h ^ 0x5bd1e995u) & 255u) < frac_zipf_x256;
        uint32_t row = z ? zipf_rank(h, lo, span, n_items) : h % n_items;
        row = (uint32_t)(((uint64_t)row * 2654435761ull) % n_items);          // ranks scattered over the table
        if (LOADS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += SPLIT ? table[((size_t)k * n_items + row) * 16 + sub] : table[(size_t)row * 64 + sub + 16 * k];
            if (sub == 0) acc += bias[(size_t)row * 16];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) unsafeAtomicAdd(SPLIT ? table + ((size_t)k * n_items + row) * 16 + sub : table + (size_t)row * 64 + sub + 16 * k, 1e-9f);
        if (sub == 0) unsafeAtomicAdd(bias + (size_t)row * 16, 1e-9f);
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const uint32_t n_items = 50000;
    float *table, *bias, *out;
    (void)hipMalloc(&table, (size_t)n_items * 256); (void)hipMalloc(&bias, (size_t)n_items * 64); (void)hipMalloc(&out, 64);
    (void)hipMemset(table, 0, (size_t)n_items * 256); (void)hipMemset(bias, 0, (size_t)n_items * 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int cus = 256, iters = 320;                 // 256 x 64 groups x 320 = 5.2 M row updates = 26 M requests
    const double rows = (double)cus * 64 * iters;
    const uint32_t skips[] = {0, 64, 256, 1024};
    for (int split = 0; split < 2; ++split)
    for (int loads = 0; loads < 2; ++loads)
        for (int frac : {0, 128, 256})
            for (uint32_t skip : skips) {
                if (frac == 0 && skip != 0) continue;
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipEventRecord(e0);
                    if (loads && split) skew_kernel<true, true><<<cus, 1024>>>(table, bias, n_items, skip, frac, iters, out);
                    else if (loads) skew_kernel<true, false><<<cus, 1024>>>(table, bias, n_items, skip, frac, iters, out);
                    else if (split) skew_kernel<false, true><<<cus, 1024>>>(table, bias, n_items, skip, frac, iters, out);
                    else skew_kernel<false, false><<<cus, 1024>>>(table, bias, n_items, skip, frac, iters, out);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1);
                }
                printf("%s %s zipf share %3d/256, head of %4u ranks left out: %7.3f ms  %6.2f G row updates/s  %6.2f G requests/s\n",
                       split ? "split " : "contig", loads ? "load+atomic" : "atomic only", frac, skip, ms, rows / ms * 1e-6, rows * 5 / ms * 1e-6);
            }
    return 0;
}
