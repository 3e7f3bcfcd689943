// How fast does MI355X gather random FACTOR ROWS (256 / 512 contiguous bytes), and does the lane layout of the loads matter?
// The SGD kernels read a row with the layout their atomics want (lane s of a 16-lane row group owns dwords s, s+16, ...: every
// load instruction covers one 64-byte segment, KPL instructions per row).  WARP's candidate scoring only READS rows (~21 per
// update), and profiles/r03_notes.md found its time linear in the 64-byte requests per candidate.  Question: does ONE 16-byte
// load per lane (16 lanes x 16 B = 256 contiguous bytes per instruction) reach L2 / the fabric as fewer, larger requests?
//   hipcc --offload-arch=gfx950 -O3 row_gather.hip -o row_gather
// Variants (rows per second, NB rows in flight per group):
//   strided   KPL x global_load_dword   (the kernels' layout)
//   vec4      KPL/4 x global_load_dwordx4, lane s owns floats 4s .. 4s+3 (+ 64 s' for the next 256 bytes)
//   lds       global_load_lds_dwordx4 (gfx950 LDS DMA, no destination registers), scored from LDS
// each with the row's 4-byte bias read beside it (unpadded table) or not.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KPL, int NB, bool BIAS>
__global__ void __launch_bounds__(256) gather_strided(const float *__restrict__ table, const float *__restrict__ bias, uint32_t n_rows, int iters, float *out) {
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16, sub = threadIdx.x & 15;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        float v[NB][KPL], w[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const uint32_t r = mix32(group * 7919u + (it * NB + q) * 104729u + 1u) % n_rows;
#pragma unroll
            for (int k = 0; k < KPL; ++k) v[q][k] = table[(size_t)r * (16 * KPL) + sub + 16 * k];
            w[q] = BIAS ? bias[r] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) acc += v[q][k];
            acc += w[q];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int KPL, int NB, bool BIAS>
__global__ void __launch_bounds__(256) gather_vec4(const float *__restrict__ table, const float *__restrict__ bias, uint32_t n_rows, int iters, float *out) {
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16, sub = threadIdx.x & 15;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        f4 v[NB][KPL / 4];
        float w[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const uint32_t r = mix32(group * 7919u + (it * NB + q) * 104729u + 1u) % n_rows;
#pragma unroll
            for (int k = 0; k < KPL / 4; ++k) v[q][k] = *(const f4 *)(table + (size_t)r * (16 * KPL) + 64 * k + 4 * sub);
            w[q] = BIAS ? bias[r] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
#pragma unroll
            for (int k = 0; k < KPL / 4; ++k) acc += (v[q][k].x + v[q][k].y) + (v[q][k].z + v[q][k].w);
            acc += w[q];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// LDS DMA: each lane's 16 bytes land at M0-base + instruction offset + 16 * lane; a wavefront's four row groups therefore fill
// 1 KiB of LDS per instruction (4 rows of 256 B).  NB instructions in flight, then read back with ds_read_b128.
template <int KPL, int NB>
__global__ void __launch_bounds__(256) gather_lds(const float *__restrict__ table, uint32_t n_rows, int iters, float *out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16, sub = threadIdx.x & 15;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *mine = lds + (size_t)wave * NB * (KPL / 4) * 256;      // per wave: NB x KPL/4 slots of 64 lanes x 16 B
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const uint32_t r = mix32(group * 7919u + (it * NB + q) * 104729u + 1u) % n_rows;
#pragma unroll
            for (int k = 0; k < KPL / 4; ++k) {
                const float *src = table + (size_t)r * (16 * KPL) + 64 * k + 4 * sub;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(mine + (q * (KPL / 4) + k) * 256), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int k = 0; k < KPL / 4; ++k) {
                const f4 v = *(const f4 *)(mine + (q * (KPL / 4) + k) * 256 + 4 * lane);
                acc += (v.x + v.y) + (v.z + v.w);
            }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <class K, class... A>
static double time_ms(K kernel, dim3 grid, dim3 block, size_t lds, A... args) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, grid, block, lds, 0, args...);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return best;
}

template <int KPL>
static void run(const char *name, float *table, float *bias, float *out, uint32_t n_rows) {
    const int blocks = 4096, threads = 256, iters = 64;
    const double groups = (double)blocks * threads / 16;
    constexpr int NB = KPL >= 8 ? 2 : 4;
    const double rows = groups * iters * NB;
    const double a0 = time_ms(gather_strided<KPL, NB, false>, dim3(blocks), dim3(threads), 0, (const float *)table, (const float *)bias, n_rows, iters, out);
    const double a1 = time_ms(gather_strided<KPL, NB, true>, dim3(blocks), dim3(threads), 0, (const float *)table, (const float *)bias, n_rows, iters, out);
    const double b0 = time_ms(gather_vec4<KPL, NB, false>, dim3(blocks), dim3(threads), 0, (const float *)table, (const float *)bias, n_rows, iters, out);
    const double b1 = time_ms(gather_vec4<KPL, NB, true>, dim3(blocks), dim3(threads), 0, (const float *)table, (const float *)bias, n_rows, iters, out);
    const double b2 = time_ms(gather_vec4<KPL, 2 * NB, true>, dim3(blocks), dim3(threads), 0, (const float *)table, (const float *)bias, n_rows, iters / 2, out);
    const size_t lds = (size_t)4 * NB * (KPL / 4) * 256 * sizeof(float);
    const double c0 = time_ms(gather_lds<KPL, NB>, dim3(blocks), dim3(threads), lds, (const float *)table, n_rows, iters, out);
    const size_t lds2 = (size_t)4 * 4 * NB * (KPL / 4) * 256 * sizeof(float);
    const double c1 = time_ms(gather_lds<KPL, 4 * NB>, dim3(blocks), dim3(threads), lds2, (const float *)table, n_rows, iters / 4, out);
    printf("%-28s rows of %3d B, %8u rows (%6.1f MiB): G rows/s  strided %5.2f (+bias %5.2f)   vec4 %5.2f (+bias %5.2f, %d in flight %5.2f)   lds-dma %5.2f (%d in flight %5.2f)\n",
           name, 64 * KPL, n_rows, (double)n_rows * 64 * KPL / 1048576.0, rows / a0 * 1e-6, rows / a1 * 1e-6, rows / b0 * 1e-6, rows / b1 * 1e-6, 2 * NB,
           rows / b2 * 1e-6, rows / c0 * 1e-6, 4 * NB, rows / c1 * 1e-6);
}

int main() {
    float *table, *bias, *out;
    const size_t max_bytes = (size_t)1 << 30;
    (void)hipMalloc(&table, max_bytes); (void)hipMalloc(&bias, (size_t)16 << 20); (void)hipMalloc(&out, 64);
    (void)hipMemset(table, 0, max_bytes); (void)hipMemset(bias, 0, (size_t)16 << 20);
    run<4>("config 2/3 item table", table, bias, out, 50000);
    run<4>("config 4 item table", table, bias, out, 200000);
    run<4>("k=64, 1 M rows", table, bias, out, 1000000);
    run<8>("config 5 item table (k=128)", table, bias, out, 1000000);
    run<8>("k=128, 50 k rows", table, bias, out, 50000);
    return 0;
}
