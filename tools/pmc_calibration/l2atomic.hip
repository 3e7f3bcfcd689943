// Experiment: are workgroup-scope fp32 atomics on GLOBAL memory executed in the XCD's L2 (no fabric trip), and how fast?
// Every workgroup reads the XCC id it actually runs on and confines itself to that XCD's private region, so L2-local
// atomics are coherent by construction (only CUs sharing that L2 touch the region).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 15u; }

template <int SCOPE>   // 0 agent (memory side), 1 workgroup (L2?)
__global__ void atomic_region(float *table, uint32_t segs_per_region, int iters, int confine, unsigned *xcc_hist) {
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16, sub = threadIdx.x & 15;
    const uint32_t xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(xcc_hist + xcc, 1u);
    const uint32_t region = confine ? xcc : (group & 7u);
    float *base = table + (size_t)region * segs_per_region * 16;
    for (int it = 0; it < iters; ++it) {
        const uint32_t seg = mix32(group * 7919u + it * 104729u + 1u) % segs_per_region;
        if (SCOPE == 0) unsafeAtomicAdd(base + (size_t)seg * 16 + sub, 1.0f);
        else __hip_atomic_fetch_add(base + (size_t)seg * 16 + sub, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ void sum_kernel(const float *t, size_t n, double *out) {
    double s = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += t[i];
    atomicAdd(out, s);
}

int main() {
    const uint32_t segs = 24576;                       // 1.5 MiB per region
    const size_t n = (size_t)8 * segs * 16;
    float *table; unsigned *hist; double *sum;
    hipMalloc(&table, n * 4); hipMalloc(&hist, 64); hipMalloc(&sum, 8);
    const int blocks = 2048, threads = 256, iters = 512;
    const double total = (double)blocks * threads * iters;
    for (int variant = 0; variant < 3; ++variant) {
        hipMemset(table, 0, n * 4); hipMemset(hist, 0, 64); hipMemset(sum, 0, 8);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (variant == 0) atomic_region<0><<<blocks, threads>>>(table, segs, iters, 0, hist);
        if (variant == 1) atomic_region<0><<<blocks, threads>>>(table, segs, iters, 1, hist);
        if (variant == 2) atomic_region<1><<<blocks, threads>>>(table, segs, iters, 1, hist);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        sum_kernel<<<256, 256>>>(table, n, sum);
        double h; unsigned hh[16]; hipMemcpy(&h, sum, 8, hipMemcpyDeviceToHost); hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost);
        printf("variant %d (%s): %.3f ms, %.1f G requests/s, sum %.0f of %.0f %s  xcc hist %u %u %u %u %u %u %u %u\n", variant,
               variant == 0 ? "agent scope, any region" : variant == 1 ? "agent scope, own-XCD region" : "workgroup scope, own-XCD region",
               ms, total / 16 / ms / 1e6, h, total, h == total ? "OK" : "LOST UPDATES", hh[0], hh[1], hh[2], hh[3], hh[4], hh[5], hh[6], hh[7]);
    }
    return 0;
}
