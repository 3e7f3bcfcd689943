// Random 64-byte request rates of MI355X as a function of the table size (is a table that fits the 256 MB Infinity Cache, or
// the 8 x 4 MB L2s, served faster than HBM?).  16-lane groups, one 64-byte segment per request -- the SGD kernel's pattern.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics request_rate.hip -o request_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void gather64(const float *__restrict__ table, uint32_t n_seg, int iters, float *out) {
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16, sub = threadIdx.x & 15;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const uint32_t seg = mix32(group * 7919u + it * 104729u + 1u) % n_seg;
        acc += __builtin_nontemporal_load(table + (size_t)seg * 16 + sub);
    }
    if (acc == 123.456f) out[0] = acc;
}

__global__ void atomic64(float *table, uint32_t n_seg, int iters) {
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16, sub = threadIdx.x & 15;
    for (int it = 0; it < iters; ++it) {
        const uint32_t seg = mix32(group * 7919u + it * 104729u + 1u) % n_seg;
        unsafeAtomicAdd(table + (size_t)seg * 16 + sub, 1.0f);
    }
}

// the SGD mix: per iteration 2 row reads + 2 row atomics of 4 segments each
__global__ void mixed64(float *table, uint32_t n_rows, int iters, float *out) {
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16, sub = threadIdx.x & 15;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const uint32_t a = mix32(group * 7919u + it * 104729u + 1u) % n_rows, b = mix32(group * 7919u + it * 104729u + 77u) % n_rows;
        float ra[4], rb[4];
        for (int k = 0; k < 4; ++k) { ra[k] = table[(size_t)a * 64 + sub + 16 * k]; rb[k] = table[(size_t)b * 64 + sub + 16 * k]; }
        for (int k = 0; k < 4; ++k) {
            unsafeAtomicAdd(table + (size_t)a * 64 + sub + 16 * k, 1e-9f * rb[k]);
            unsafeAtomicAdd(table + (size_t)b * 64 + sub + 16 * k, 1e-9f * ra[k]);
            acc += ra[k] * rb[k];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    float *table, *out;
    const size_t max_bytes = (size_t)4 << 30;
    (void)hipMalloc(&table, max_bytes); (void)hipMalloc(&out, 64);
    (void)hipMemset(table, 0, max_bytes);
    const int blocks = 4096, threads = 256, iters = 256;
    const double groups = (double)blocks * threads / 16;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t sizes[] = {(size_t)2 << 20, (size_t)13 << 20, (size_t)52 << 20, (size_t)128 << 20, (size_t)512 << 20, (size_t)4 << 30};
    for (size_t bytes : sizes) {
        const uint32_t n_seg = (uint32_t)(bytes / 64);
        float ms[3];
        for (int k = 0; k < 3; ++k) {
            for (int rep = 0; rep < 2; ++rep) {       // second repetition is timed (first warms the caches)
                (void)hipEventRecord(e0);
                if (k == 0) gather64<<<blocks, threads>>>(table, n_seg, iters, out);
                else if (k == 1) atomic64<<<blocks, threads>>>(table, n_seg, iters);
                else mixed64<<<blocks, threads>>>(table, n_seg / 4, iters / 8, out);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms[k], e0, e1);
            }
        }
        printf("table %6zu MiB: reads %6.1f G req/s   atomics %6.1f G req/s   SGD mix (8 reads + 8 atomics per step) %6.1f G req/s = %5.2f G steps/s\n",
               bytes >> 20, groups * iters / ms[0] * 1e-6, groups * iters / ms[1] * 1e-6, groups * (iters / 8) * 16 / ms[2] * 1e-6,
               groups * (iters / 8) / ms[2] * 1e-6);
    }
    return 0;
}
