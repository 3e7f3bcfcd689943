// LDS fp32 atomic-add throughput on gfx950: how many cycles does one ds_add_f32 wave instruction occupy the LDS for,
// depending on active lanes and on address sharing?   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) float lds_float;

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, int iters, int stride_rows) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.0f;
    __syncthreads();
    lds_float *p = (lds_float *)lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const int row = (it * 7 + wave * stride_rows) & 63;     // 64 rows x 64 dwords
        if (MODE == 0) {            // 64 lanes, 64 consecutive dwords of one row
            __hip_atomic_fetch_add(p + row * 64 + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 1) {     // 16 active lanes
            if (lane < 16) __hip_atomic_fetch_add(p + row * 64 + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 2) {     // 4 groups of 16 lanes, all on the SAME 16 dwords
            __hip_atomic_fetch_add(p + row * 64 + (lane & 15), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 3) {     // 4 groups on 4 different rows, same dword index -> same banks
            __hip_atomic_fetch_add(p + ((row + (lane >> 4)) & 63) * 64 + (lane & 15), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 4) {     // plain read-modify-write, 64 consecutive dwords
            p[row * 64 + lane] = p[row * 64 + lane] + 1.0f;
        } else if (MODE == 5) {     // plain read only
            acc += p[row * 64 + lane];
        } else if (MODE == 6) {     // 4 groups on 4 different rows, dword index rotated by group -> distinct banks
            __hip_atomic_fetch_add(p + ((row + (lane >> 4)) & 63) * 64 + ((lane & 15) + 16 * (lane >> 4)), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 8) {     // integer atomic, 64 lanes / 64 dwords
            atomicAdd((int *)lds + row * 64 + lane, 3);
        } else if (MODE == 9) {     // integer atomic, 4 groups on the same 16 dwords
            atomicAdd((int *)lds + row * 64 + (lane & 15), 3);
        } else if (MODE == 7) {     // returning atomic
            acc += __hip_atomic_fetch_add(p + row * 64 + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (acc == 123.456f || threadIdx.x == 0) out[blockIdx.x] = lds[threadIdx.x] + acc;
}

template <int MODE>
void run(const char *name, int stride_rows) {
    float *out; hipMalloc(&out, 4096);
    const int iters = 20000, grid = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<grid, 1024, 8192 * 4>>>(out, 100, stride_rows);
    hipEventRecord(a);
    k<MODE><<<grid, 1024, 8192 * 4>>>(out, iters, stride_rows);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: 16 waves x iters instructions; clock ~2.4 GHz
    const double clk = ms * 1e-3 * 2.4e9, instr = 16.0 * iters;
    printf("%-58s stride %2d: %8.3f ms  %6.1f clk per wave instruction per CU\n", name, stride_rows, ms, clk / instr);
    hipFree(out);
}

int main() {
    for (int s : {0, 1}) {
        run<0>("ds_add_f32 64 lanes / 64 dwords", s);
        run<1>("ds_add_f32 16 lanes", s);
        run<2>("ds_add_f32 4 groups same 16 dwords", s);
        run<3>("ds_add_f32 4 groups, 4 rows, same banks", s);
        run<6>("ds_add_f32 4 groups, 4 rows, distinct banks", s);
        run<7>("ds_add_rtn_f32 64 lanes", s);
        run<8>("ds_add_u32 64 lanes / 64 dwords", s);
        run<9>("ds_add_u32 4 groups same 16 dwords", s);
        run<4>("plain ds_read + ds_write 64 lanes", s);
        run<5>("plain ds_read 64 lanes", s);
    }
    return 0;
}
