// pipe_model.hip's synthetic row loop with config 2's ADDRESS MIX: the positive item Zipf(1)-distributed over 50,000 items, its 64 most
// popular ranks accumulated in LDS (no atomics), the negative uniform; 5 atomic requests per item (4 row segments + a bias line).
// Which FORM of the loop gets closest to the atomic path's capacity for that mix?  Timing experiment (profiles/r05_notes.md).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics pipe_model2.hip -o pipe_model2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) float lds_f32;

constexpr uint32_t kItems = 50000;
constexpr int kHot = 64;

struct Args {
    float *table, *bias, *out;
    unsigned *err;
    int iters;
    int zipf;            // 1: positives Zipf(1), 0: uniform
    float lo, span;      // ln(1), ln(kItems + 1)
};

struct Row { float vi[4], vj[4], wi, wj; uint32_t i, j; int slot; };

__device__ __forceinline__ void pick(const Args &a, uint32_t group, int it, uint32_t &i, uint32_t &j, int &slot) {
    const uint32_t h = mix32(group * 7919u + it * 104729u + 1u);
    slot = -1;
    if (a.zipf) {
        const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
        uint32_t r = (uint32_t)(__expf(a.lo + u * a.span)) - 1u;
        r = r < kItems ? r : kItems - 1;
        if (r < (uint32_t)kHot) slot = (int)r;
        i = (uint32_t)(((uint64_t)r * 2654435761ull) % kItems);
    } else i = h % kItems;
    j = mix32(h ^ 0x9E3779B9u) % kItems;
}

__device__ __forceinline__ void gather(const Args &a, uint32_t group, int it, uint32_t sub, Row &r) {
    pick(a, group, it, r.i, r.j, r.slot);
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.vi[k] = a.table[(size_t)r.i * 64 + sub + 16 * k]; r.vj[k] = a.table[(size_t)r.j * 64 + sub + 16 * k]; }
    r.wi = a.bias[(size_t)r.i * 16 + (sub & 1)];
    r.wj = a.bias[(size_t)r.j * 16 + (sub & 1)];
}

template <int VALU_N>
__device__ __forceinline__ float arith(const Row &r, float (&vu)[4], float (&di)[4], float (&dj)[4], lds_i32 *hot, uint32_t sub) {
    float p = 0.0f;
    float vi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) vi[k] = r.vi[k];
    if (r.slot >= 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) vi[k] += (float)hot[r.slot * 65 + sub + 16 * k] * 1e-9f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) p += vu[k] * (vi[k] - r.vj[k]);
    p += __shfl_xor(p, 8); p += __shfl_xor(p, 4); p += __shfl_xor(p, 2); p += __shfl_xor(p, 1);
    float x = p + r.wi - r.wj, y = 0.5f, z = 0.25f, w = 0.125f;
#pragma unroll 8
    for (int n = 0; n < VALU_N / 4; ++n) { x = x * 0.999f + 0.001f; y = y * 0.998f + x * 1e-9f; z = z * 0.997f + 0.002f; w = w * 0.996f + 0.003f; }
    const float g = 1e-6f * (x + y + z + w);
#pragma unroll
    for (int k = 0; k < 4; ++k) { di[k] = g * vu[k]; dj[k] = -g * vu[k]; vu[k] += g * (vi[k] - r.vj[k]); }
    return g;
}

__device__ __forceinline__ void hot_add(lds_i32 *hot, int slot, const float (&di)[4], float g, uint32_t sub) {
#pragma unroll
    for (int k = 0; k < 4; ++k) __hip_atomic_fetch_add(hot + slot * 65 + sub + 16 * k, (int)(di[k] * 1e9f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (sub == 0) __hip_atomic_fetch_add(hot + slot * 65 + 64, (int)(g * 1e9f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the positive's update: LDS sums for a hot slot, atomics otherwise (varying instruction count)
__device__ __forceinline__ void scatter_pos(const Args &a, lds_i32 *hot, uint32_t i, int slot, const float (&di)[4], float g, uint32_t sub) {
    if (slot >= 0) hot_add(hot, slot, di, g, sub);
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) unsafeAtomicAdd(a.table + (size_t)i * 64 + sub + 16 * k, di[k]);
        if (sub == 0) unsafeAtomicAdd(a.bias + (size_t)i * 16, g);
    }
}
__device__ __forceinline__ void scatter_neg_rows(const Args &a, uint32_t j, const float (&dj)[4], uint32_t sub) {
#pragma unroll
    for (int k = 0; k < 4; ++k) unsafeAtomicAdd(a.table + (size_t)j * 64 + sub + 16 * k, dj[k]);
}
__device__ __forceinline__ void scatter_neg_bias(const Args &a, uint32_t j, float g, uint32_t sub) {
    if (sub == 0) unsafeAtomicAdd(a.bias + (size_t)j * 16, -g);
}

// FORM 0  serial
// FORM 1  gathers one row ahead, vmcnt(0) per row
// FORM 2  gathers two rows ahead; everything whose instruction count varies in front of the gathers, the negative's four row atomics behind
// FORM 3  serial, no atomics at all (floor)
template <int FORM, int VALU_N>
__global__ void __launch_bounds__(1024) row_kernel(const Args a) {
    __shared__ int s_hot[kHot * 65];
    lds_i32 *hot = (lds_i32 *)s_hot;
    for (int k = threadIdx.x; k < kHot * 65; k += blockDim.x) s_hot[k] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, sub = lane & 15;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    float vu[4] = {0.1f, 0.2f, 0.3f, 0.4f}, di[4], dj[4];
    if (FORM == 0 || FORM == 3) {
        for (int it = 0; it < a.iters; ++it) {
            Row r;
            gather(a, group, it, sub, r);
            const float g = arith<VALU_N>(r, vu, di, dj, hot, sub);
            if (FORM == 0) { scatter_pos(a, hot, r.i, r.slot, di, g, sub); scatter_neg_bias(a, r.j, g, sub); scatter_neg_rows(a, r.j, dj, sub); }
            else if (r.slot >= 0) hot_add(hot, r.slot, di, g, sub);
        }
    } else if (FORM == 1) {
        Row cur, nxt;
        gather(a, group, 0, sub, cur);
        for (int it = 0; it < a.iters; ++it) {
            gather(a, group, it + 1, sub, nxt);
            const float g = arith<VALU_N>(cur, vu, di, dj, hot, sub);
            scatter_pos(a, hot, cur.i, cur.slot, di, g, sub); scatter_neg_bias(a, cur.j, g, sub); scatter_neg_rows(a, cur.j, dj, sub);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            cur = nxt;
        }
    } else {
        Row ra, rb;
        gather(a, group, 0, sub, ra); gather(a, group, 1, sub, rb);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int it = 0; it < a.iters; it += 2) {
            float g; uint32_t j0;
            g = arith<VALU_N>(ra, vu, di, dj, hot, sub); scatter_pos(a, hot, ra.i, ra.slot, di, g, sub); scatter_neg_bias(a, ra.j, g, sub); j0 = ra.j;
            gather(a, group, it + 2, sub, ra); scatter_neg_rows(a, j0, dj, sub);
            g = arith<VALU_N>(rb, vu, di, dj, hot, sub); scatter_pos(a, hot, rb.i, rb.slot, di, g, sub); scatter_neg_bias(a, rb.j, g, sub); j0 = rb.j;
            gather(a, group, it + 3, sub, rb); scatter_neg_rows(a, j0, dj, sub);
        }
    }
    const float acc = vu[0] + vu[1] + vu[2] + vu[3];
    if (acc == 123.456f) a.out[0] = acc;
}

// writer form: 16 - NW compute wavefronts (loads only, gathers one row ahead), NW writers issuing every atomic; see pipe_model.hip
constexpr int kCap = 64, kRec = 4 + 128;
template <int VALU_N, int NW>
__global__ void __launch_bounds__(1024) row_kernel_w(const Args a) {
    __shared__ int s_hot[kHot * 65];
    __shared__ float s_ring[kCap * kRec];
    __shared__ unsigned s_ctl[8 + kCap];
    lds_i32 *hot = (lds_i32 *)s_hot;
    lds_f32 *ring = (lds_f32 *)s_ring;
    lds_u32 *ctl = (lds_u32 *)s_ctl;
    const uint32_t lane = threadIdx.x & 63, sub = lane & 15, wave = threadIdx.x >> 6;
    const int n_comp = 16 - NW;
    for (int k = threadIdx.x; k < kHot * 65; k += blockDim.x) s_hot[k] = 0;
    for (int k = threadIdx.x; k < 8 + kCap; k += blockDim.x) s_ctl[k] = 0;
    __syncthreads();
    if ((int)wave >= n_comp) {
        const unsigned w = wave - n_comp;
        unsigned t = w;
        for (;;) {
            lds_u32 *seq = ctl + 8 + (t % kCap);
            unsigned spin = 0;
            bool quit = false;
            while (__hip_atomic_load(seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t + 1u) {
                if (__hip_atomic_load(ctl + 4, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)n_comp &&
                    __hip_atomic_load(ctl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= t) { quit = true; break; }
                if (++spin > (1u << 24)) { if (lane == 0) atomicOr(a.err, 1u); quit = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (quit) break;
            lds_f32 *rec = ring + (t % kCap) * kRec;
            const uint32_t i = __float_as_uint(rec[0]), j = __float_as_uint(rec[1]);
            const bool pos = (i & 0x80000000u) == 0u;               // top bit: the positive went to the LDS sums
            const float x = rec[4 + lane], y = rec[4 + 64 + lane];
            const float dw = rec[2 + (lane & 1)];
            if (pos) unsafeAtomicAdd(a.table + (size_t)i * 64 + lane, x);
            unsafeAtomicAdd(a.table + (size_t)j * 64 + lane, y);
            if (lane < 2 && (lane == 1 || pos)) unsafeAtomicAdd(a.bias + (size_t)(lane ? j : (i & 0x7fffffffu)) * 16, dw);
            t += NW;
            if (lane == 0) __hip_atomic_store(ctl + 1 + w, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    const uint32_t group = (blockIdx.x * (uint32_t)n_comp * 4u) + wave * 4u + (lane >> 4);
    float vu[4] = {0.1f, 0.2f, 0.3f, 0.4f}, di[4], dj[4];
    Row cur, nxt;
    gather(a, group, 0, sub, cur);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int it = 0; it < a.iters; ++it) {
        gather(a, group, it + 1, sub, nxt);
        const float g = arith<VALU_N>(cur, vu, di, dj, hot, sub);
        if (cur.slot >= 0) hot_add(hot, cur.slot, di, g, sub);
        unsigned ticket = 0;
        if (sub == 0) ticket = __hip_atomic_fetch_add(ctl + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ticket = __shfl(ticket, lane & 48);
        unsigned spin = 0;
        while (ticket - __hip_atomic_load(ctl + 1 + (ticket % NW), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)kCap) {
            if (++spin > (1u << 24)) { if (sub == 0) atomicOr(a.err, 2u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        lds_f32 *rec = ring + (ticket % kCap) * kRec;
#pragma unroll
        for (int k = 0; k < 4; ++k) { rec[4 + sub + 16 * k] = di[k]; rec[4 + 64 + sub + 16 * k] = dj[k]; }
        if (sub == 0) { rec[0] = __uint_as_float(cur.i | (cur.slot >= 0 ? 0x80000000u : 0u)); rec[1] = __uint_as_float(cur.j); rec[2] = g; rec[3] = -g; }
        if (sub == 0) __hip_atomic_store(ctl + 8 + (ticket % kCap), ticket + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        cur = nxt;
    }
    if (lane == 0) __hip_atomic_fetch_add(ctl + 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const float acc = vu[0] + vu[1] + vu[2] + vu[3];
    if (acc == 123.456f) a.out[0] = acc;
}

int main() {
    Args a;
    (void)hipMalloc(&a.table, (size_t)kItems * 256); (void)hipMalloc(&a.bias, (size_t)kItems * 64); (void)hipMalloc(&a.out, 64); (void)hipMalloc(&a.err, 4);
    (void)hipMemset(a.table, 0, (size_t)kItems * 256); (void)hipMemset(a.bias, 0, (size_t)kItems * 64); (void)hipMemset(a.err, 0, 4);
    a.lo = 0.0f; a.span = logf((float)kItems + 1.0f);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int cus = 256;
    auto run = [&](const char *name, auto kernel, int threads, int groups_per_block, int zipf) {
        a.zipf = zipf;
        a.iters = (int)(5000000.0 / ((double)cus * groups_per_block)) & ~1;          // ~5 M rows per launch like config 2
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            kernel<<<cus, threads>>>(a);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned h = 0; (void)hipMemcpy(&h, a.err, 4, hipMemcpyDeviceToHost);
        const double r = (double)cus * groups_per_block * a.iters;
        printf("%-7s %-66s %4d thr: %7.3f ms  %6.3f G rows/s  %5.2f us per row step  err %u\n", zipf ? "zipf" : "uniform", name, threads, ms, r / ms * 1e-6,
               ms * 1e3 / a.iters, h);
    };
    for (int zipf = 1; zipf >= 0; --zipf) {
        run("serial, 600 VALU", row_kernel<0, 600>, 1024, 64, zipf);
        run("serial, 600 VALU, 8 waves per CU", row_kernel<0, 600>, 512, 32, zipf);
        run("one row ahead, vmcnt(0) per row, 600 VALU", row_kernel<1, 600>, 1024, 64, zipf);
        run("two rows ahead, counted waits, 600 VALU", row_kernel<2, 600>, 1024, 64, zipf);
        run("two rows ahead, counted waits, 600 VALU, 8 waves per CU", row_kernel<2, 600>, 512, 32, zipf);
        run("serial, no atomics, 600 VALU", row_kernel<3, 600>, 1024, 64, zipf);
        run("writers: 14 compute + 2, 600 VALU", row_kernel_w<600, 2>, 1024, 56, zipf);
        run("writers: 13 compute + 3, 600 VALU", row_kernel_w<600, 3>, 1024, 52, zipf);
        run("serial, 300 VALU", row_kernel<0, 300>, 1024, 64, zipf);
        run("two rows ahead, 300 VALU", row_kernel<2, 300>, 1024, 64, zipf);
        run("serial no atomics, 300 VALU", row_kernel<3, 300>, 1024, 64, zipf);
        run("writers: 14 + 2, 300 VALU", row_kernel_w<300, 2>, 1024, 56, zipf);
    }
    return 0;
}
