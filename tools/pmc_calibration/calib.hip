// PMC calibration for THIS engine's access pattern (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are calibrated only for
// 16 B/lane streams).  Kernels with a known byte count:
//   gather64   16-lane groups read random 64-byte segments (4 B per lane) of a table much larger than the 256 MB Infinity Cache
//   atomic64   the same groups add into random 64-byte segments with global_atomic_add_f32
// Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); compare with the printed true byte counts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void gather64(const float *__restrict__ table, uint32_t n_seg, int iters, float *out) {
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16, sub = threadIdx.x & 15;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const uint32_t seg = mix32(group * 7919u + it * 104729u + 1u) % n_seg;
        acc += table[(size_t)seg * 16 + sub];
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

int main() {
    const size_t bytes = (size_t)4 << 30;                       // 4 GiB table
    const uint32_t n_seg = (uint32_t)(bytes / 64);
    float *table, *out;
    hipMalloc(&table, bytes); hipMalloc(&out, 64);
    hipMemset(table, 0, bytes);
    const int blocks = 4096, threads = 256, iters = 256;
    const double groups = (double)blocks * threads / 16;
    hipDeviceSynchronize();
    gather64<<<blocks, threads>>>(table, n_seg, iters, out);
    hipDeviceSynchronize();
    atomic64<<<blocks, threads>>>(table, n_seg, iters);
    hipDeviceSynchronize();
    printf("true bytes: gather64 %.0f (%.0f requests of 64 B)   atomic64 %.0f (%.0f requests of 64 B)\n", groups * iters * 64, groups * iters,
           groups * iters * 64, groups * iters);
    return 0;
}
