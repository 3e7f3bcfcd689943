// What bounds a BPR row loop on MI355X: a synthetic row step (2 random 256-byte row gathers + 2 bias lines, ~VALU_N vector
// instructions, 2 row atomics + 2 bias atomics per 16-lane row group) in the forms the engine could take, and the shapes an
// atomic instruction can have.  Timing experiment for profiles/r05_notes.md; not product code.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics pipe_model.hip -o pipe_model
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// ---------------------------------------------------------------- atomic instruction shapes
// mode 0: 4 groups x 16 lanes -> 4 random 64-B segments     (the engine's row atomics)
// mode 1: 64 lanes -> one random 256-B row
// mode 2: 2 x 32 lanes -> 2 random 128-B lines
// mode 3: lane 0 of each 16-lane group -> 4 random dwords     (the engine's bias atomics)
// mode 4: one group of 16 lanes -> 1 random 64-B segment
// mode 5: like 0 with RETURNING atomics (result discarded)
// mode 6: like 0, plain stores
// mode 7: like 0, loads
template <int MODE>
__global__ void __launch_bounds__(1024) shape_kernel(float *table, uint32_t n_rows, int iters, float *out) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        uint32_t unit, off;
        if (MODE == 1) { unit = wave; off = lane; }
        else if (MODE == 2) { unit = wave * 2 + (lane >> 5); off = lane & 31; }
        else { unit = wave * 4 + (lane >> 4); off = lane & 15; }
        const uint32_t h = mix32(unit * 7919u + it * 104729u + 1u);
        float *p;
        if (MODE == 1) p = table + (size_t)(h % n_rows) * 64 + off;
        else if (MODE == 2) p = table + (size_t)(h % (n_rows * 2)) * 32 + off;
        else p = table + (size_t)(h % (n_rows * 4)) * 16 + off;
        bool on = true;
        if (MODE == 3) on = (lane & 15) == 0;
        if (MODE == 4) on = lane < 16;
        if (on) {
            if (MODE == 5) acc += atomicAdd(p, 1.0f);
            else if (MODE == 6) *p = (float)it;
            else if (MODE == 7) acc += *p;
            else unsafeAtomicAdd(p, 1.0f);
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// ---------------------------------------------------------------- synthetic row loop
// FORM 0  serial: gathers -> wait -> arithmetic -> atomics; the next row's gathers are issued behind the atomics
// FORM 1  gathers of row t+1 issued before the arithmetic of row t, atomics at the row's end, vmcnt(0) at the top of every row
// FORM 2  same, counted wait (the compiler's own: vmcnt(10)) -- the atomics of row t stay in flight over row t+1
// FORM 3  FORM 0 without atomics
// FORM 4  FORM 2 without atomics
// FORM 6  gathers TWO rows ahead, the bias atomics (a branch) in front of them: see the code
// FORM 5  FORM 2, atomics deferred: issued at the top of row t+1 (behind the gathers of row t+2), so that they are two rows old
//         when anything waits for them
template <int FORM, int VALU_N>
__global__ void __launch_bounds__(1024) row_kernel(float *table, float *bias, uint32_t n_rows, int iters, float *out) {
    const uint32_t lane = threadIdx.x & 63, sub = lane & 15;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    float vu[4] = {0.1f, 0.2f, 0.3f, 0.4f};
    float acc = 0.0f;
    struct Row { float vi[4], vj[4], wi, wj; uint32_t i, j; };
    auto gather = [&](int it, Row &r) {
        r.i = mix32(group * 7919u + it * 104729u + 1u) % n_rows;
        r.j = mix32(group * 7919u + it * 104729u + 77u) % n_rows;
#pragma unroll
        for (int k = 0; k < 4; ++k) { r.vi[k] = table[(size_t)r.i * 64 + sub + 16 * k]; r.vj[k] = table[(size_t)r.j * 64 + sub + 16 * k]; }
        r.wi = bias[(size_t)r.i * 16 + (sub & 1)];
        r.wj = bias[(size_t)r.j * 16 + (sub & 1)];
    };
    float di[4], dj[4], dwi = 0.0f, dwj = 0.0f;
    auto arith = [&](const Row &r) {
        float p = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) p += vu[k] * (r.vi[k] - r.vj[k]);
        p += __shfl_xor(p, 8); p += __shfl_xor(p, 4); p += __shfl_xor(p, 2); p += __shfl_xor(p, 1);
        float x = p + r.wi - r.wj, y = 0.5f, z = 0.25f, w = 0.125f;
        // four independent chains of dependent multiply-adds: VALU_N vector instructions in all
#pragma unroll 8
        for (int n = 0; n < VALU_N / 4; ++n) { x = x * 0.999f + 0.001f; y = y * 0.998f + x * 1e-9f; z = z * 0.997f + 0.002f; w = w * 0.996f + 0.003f; }
        const float g = 1e-6f * (x + y + z + w);
#pragma unroll
        for (int k = 0; k < 4; ++k) { di[k] = g * vu[k]; dj[k] = -g * vu[k]; vu[k] += g * (r.vi[k] - r.vj[k]); }
        dwi = g; dwj = -g;
    };
    auto scatter_rows = [&](uint32_t i, uint32_t j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { unsafeAtomicAdd(table + (size_t)i * 64 + sub + 16 * k, di[k]); unsafeAtomicAdd(table + (size_t)j * 64 + sub + 16 * k, dj[k]); }
    };
    auto scatter_bias = [&](uint32_t i, uint32_t j) {
        if (FORM == 2) {
            // a FIXED number of atomic instructions per row (no branch the static wait count would have to cover): the bias lines take
            // a 16-lane add whose lanes 1 .. 15 add zero -- still one 64-byte request per line
            unsafeAtomicAdd(bias + (size_t)i * 16 + sub, sub == 0 ? dwi : 0.0f);
            unsafeAtomicAdd(bias + (size_t)j * 16 + sub, sub == 0 ? dwj : 0.0f);
        } else if (sub == 0) { unsafeAtomicAdd(bias + (size_t)i * 16, dwi); unsafeAtomicAdd(bias + (size_t)j * 16, dwj); }
    };
    auto scatter = [&](uint32_t i, uint32_t j) { scatter_rows(i, j); scatter_bias(i, j); };
    if (FORM == 0 || FORM == 3) {
        for (int it = 0; it < iters; ++it) {
            Row r;
            gather(it, r);
            arith(r);
            if (FORM == 0) scatter(r.i, r.j);
        }
    } else if (FORM == 5) {
        Row cur, nxt;
        gather(0, cur);
        uint32_t pi = 0, pj = 0;
        bool pending = false;
        for (int it = 0; it < iters; ++it) {
            gather(it + 1, nxt);
            if (pending) scatter(pi, pj);                 // row it-1's atomics, behind row it+1's gathers
            arith(cur);
            pi = cur.i; pj = cur.j; pending = true;
            cur = nxt;
        }
        scatter(pi, pj);
    } else if (FORM == 6) {
        // two rows ahead: [arithmetic t] [bias atomics t: the only instructions whose number varies] [gathers t+2] [row atomics t].
        // The static wait for row t+1's gathers then leaves 26 younger instructions in flight; what it can over-wait for is a row old.
        Row ra, rb;
        gather(0, ra); gather(1, rb);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int it = 0; it < iters; it += 2) {
            uint32_t i0, j0;
            arith(ra); scatter_bias(ra.i, ra.j); i0 = ra.i; j0 = ra.j; gather(it + 2, ra); scatter_rows(i0, j0);
            arith(rb); scatter_bias(rb.i, rb.j); i0 = rb.i; j0 = rb.j; gather(it + 3, rb); scatter_rows(i0, j0);
        }
    } else {
        Row cur, nxt;
        gather(0, cur);
        __builtin_amdgcn_s_waitcnt(0x0F70);               // (the loop's static wait counts must not have to cover this entry path)
        for (int it = 0; it < iters; ++it) {
            gather(it + 1, nxt);
            arith(cur);
            if (FORM != 4) scatter(cur.i, cur.j);
            if (FORM == 1) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), lgkmcnt / expcnt untouched
            cur = nxt;
        }
    }
    acc = vu[0] + vu[1] + vu[2] + vu[3];
    if (acc == 123.456f) out[0] = acc;
}

// ---------------------------------------------------------------- writer-wavefront form
// Wavefronts 0 .. 15-NW of a 1024-thread workgroup compute and carry LOADS ONLY; every row's deltas go through an LDS ring to NW writer
// wavefronts, which issue the atomics (a whole 256-byte row per instruction).  Multi-producer ring: a ticket from `tail`, the slot is
// free once `head` has passed ticket - CAP, the record is published by storing ticket + 1 into the slot's sequence word; writer w takes
// the tickets = w mod NW in order.  Spin limits turn a protocol error into a flag instead of a hang.
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) float lds_f32;
constexpr int kCap = 64, kRec = 4 + 128;
template <int VALU_N, int NW, bool ATOMICS>
__global__ void __launch_bounds__(1024) row_kernel_w(float *table, float *bias, uint32_t n_rows, int iters, float *out, unsigned *err) {
    __shared__ float s_ring[kCap * kRec];
    __shared__ unsigned s_ctl[8 + kCap];          // [0] tail | [1 + w] head of writer w | [4] producers done | [8 ..] sequence words
    lds_f32 *ring = (lds_f32 *)s_ring;
    lds_u32 *ctl = (lds_u32 *)s_ctl;
    const uint32_t lane = threadIdx.x & 63, sub = lane & 15, wave = threadIdx.x >> 6;
    const int n_comp = 16 - NW;
    for (int k = threadIdx.x; k < 8 + kCap; k += blockDim.x) s_ctl[k] = 0;
    __syncthreads();
    if ((int)wave >= n_comp) {
        // ---- writer
        const unsigned w = wave - n_comp;
        unsigned t = w;
        for (;;) {
            lds_u32 *seq = ctl + 8 + (t % kCap);
            unsigned spin = 0;
            bool quit = false;
            while (__hip_atomic_load(seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t + 1u) {
                if (__hip_atomic_load(ctl + 4, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)n_comp &&
                    __hip_atomic_load(ctl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= t) { quit = true; break; }
                if (++spin > (1u << 24)) { if (lane == 0) atomicOr(err, 1u); quit = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (quit) break;
            lds_f32 *rec = ring + (t % kCap) * kRec;
            const uint32_t i = __float_as_uint(rec[0]), j = __float_as_uint(rec[1]);
            const float a = rec[4 + lane], b = rec[4 + 64 + lane];
            const float dw = rec[2 + (lane & 1)];
            if (ATOMICS) {
                unsafeAtomicAdd(table + (size_t)i * 64 + lane, a);
                unsafeAtomicAdd(table + (size_t)j * 64 + lane, b);
                if (lane < 2) unsafeAtomicAdd(bias + (size_t)(lane ? j : i) * 16, dw);
            } else if (a + b + dw == 123.456f) out[1] = a;
            t += NW;
            if (lane == 0) __hip_atomic_store(ctl + 1 + w, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     // (the record is in registers)
        }
        return;
    }
    // ---- compute wavefronts
    const uint32_t group = (blockIdx.x * (uint32_t)n_comp * 4u) + wave * 4u + (lane >> 4);
    float vu[4] = {0.1f, 0.2f, 0.3f, 0.4f};
    struct Row { float vi[4], vj[4], wi, wj; uint32_t i, j; };
    auto gather = [&](int it, Row &r) {
        r.i = mix32(group * 7919u + it * 104729u + 1u) % n_rows;
        r.j = mix32(group * 7919u + it * 104729u + 77u) % n_rows;
#pragma unroll
        for (int k = 0; k < 4; ++k) { r.vi[k] = table[(size_t)r.i * 64 + sub + 16 * k]; r.vj[k] = table[(size_t)r.j * 64 + sub + 16 * k]; }
        r.wi = bias[(size_t)r.i * 16 + (sub & 1)];
        r.wj = bias[(size_t)r.j * 16 + (sub & 1)];
    };
    Row cur, nxt;
    gather(0, cur);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int it = 0; it < iters; ++it) {
        gather(it + 1, nxt);
        float p = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) p += vu[k] * (cur.vi[k] - cur.vj[k]);
        p += __shfl_xor(p, 8); p += __shfl_xor(p, 4); p += __shfl_xor(p, 2); p += __shfl_xor(p, 1);
        float x = p + cur.wi - cur.wj, y = 0.5f, z = 0.25f, w = 0.125f;
#pragma unroll 8
        for (int n = 0; n < VALU_N / 4; ++n) { x = x * 0.999f + 0.001f; y = y * 0.998f + x * 1e-9f; z = z * 0.997f + 0.002f; w = w * 0.996f + 0.003f; }
        const float g = 1e-6f * (x + y + z + w);
        // hand the row's deltas to the writers
        unsigned ticket = 0;
        if (sub == 0) ticket = __hip_atomic_fetch_add(ctl + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ticket = __shfl(ticket, lane & 48);
        unsigned spin = 0;
        while (ticket - __hip_atomic_load(ctl + 1 + (ticket % NW), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)kCap) {
            if (++spin > (1u << 24)) { if (sub == 0) atomicOr(err, 2u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        lds_f32 *rec = ring + (ticket % kCap) * kRec;
#pragma unroll
        for (int k = 0; k < 4; ++k) { rec[4 + sub + 16 * k] = g * vu[k]; rec[4 + 64 + sub + 16 * k] = -g * vu[k]; vu[k] += g * (cur.vi[k] - cur.vj[k]); }
        if (sub == 0) { rec[0] = __uint_as_float(cur.i); rec[1] = __uint_as_float(cur.j); rec[2] = g; rec[3] = -g; }
        if (sub == 0) __hip_atomic_store(ctl + 8 + (ticket % kCap), ticket + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        cur = nxt;
    }
    if (lane == 0) __hip_atomic_fetch_add(ctl + 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const float acc = vu[0] + vu[1] + vu[2] + vu[3];
    if (acc == 123.456f) out[0] = acc;
}

static double time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); return ms; }

int main(int argc, char **argv) {
    const uint32_t n_rows = 150000;                          // 38 MB of 256-byte rows: config 2's tables (Infinity-Cache resident)
    float *table, *bias, *out;
    (void)hipMalloc(&table, (size_t)n_rows * 256); (void)hipMalloc(&bias, (size_t)n_rows * 64); (void)hipMalloc(&out, 64);
    (void)hipMemset(table, 0, (size_t)n_rows * 256); (void)hipMemset(bias, 0, (size_t)n_rows * 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);

    // ---- shapes: wave instructions per second and per CU
    const char *names[8] = {"4 x 64 B segments (row atomics)", "1 x 256 B row", "2 x 128 B lines", "4 single dwords (bias atomics)", "1 x 64 B segment (16 lanes)",
                            "4 x 64 B, RETURNING atomics", "4 x 64 B, plain stores", "4 x 64 B, loads"};
    const int grids[4] = {cus, cus / 2, cus / 4, cus / 8};
    const int tpb[3] = {1024, 512, 256};
    for (int mode = 0; mode < 8; ++mode) {
        for (int gi = 0; gi < 4; ++gi) {
            for (int ti = 0; ti < 3; ++ti) {
                if (gi > 0 && ti > 0) continue;
                if (mode != 0 && mode != 1 && mode != 7 && (gi > 0 || ti > 0)) continue;
                const int iters = 2048;
                double ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipEventRecord(e0);
                    switch (mode) {
                    case 0: shape_kernel<0><<<grids[gi], tpb[ti]>>>(table, n_rows, iters, out); break;
                    case 1: shape_kernel<1><<<grids[gi], tpb[ti]>>>(table, n_rows, iters, out); break;
                    case 2: shape_kernel<2><<<grids[gi], tpb[ti]>>>(table, n_rows, iters, out); break;
                    case 3: shape_kernel<3><<<grids[gi], tpb[ti]>>>(table, n_rows, iters, out); break;
                    case 4: shape_kernel<4><<<grids[gi], tpb[ti]>>>(table, n_rows, iters, out); break;
                    case 5: shape_kernel<5><<<grids[gi], tpb[ti]>>>(table, n_rows, iters, out); break;
                    case 6: shape_kernel<6><<<grids[gi], tpb[ti]>>>(table, n_rows, iters, out); break;
                    default: shape_kernel<7><<<grids[gi], tpb[ti]>>>(table, n_rows, iters, out); break;
                    }
                    (void)hipEventRecord(e1);
                    ms = time_ms(e0, e1);
                }
                const double winstr = (double)grids[gi] * (tpb[ti] / 64) * iters;
                printf("shape %-34s grid %3d x %4d: %7.3f ms  %7.2f G wave-instr/s  %6.1f ns per wave-instr and CU\n", names[mode], grids[gi], tpb[ti], ms,
                       winstr / ms * 1e-6, ms * 1e6 / (winstr / grids[gi]));
            }
        }
    }

    // ---- synthetic row loop, one 1024-thread workgroup per CU (16 wavefronts, 64 rows in flight per CU = the engine's geometry)
    const int iters = 400;
    const double rows = (double)cus * 64 * iters;
    auto run = [&](const char *name, auto kernel, int threads) {
        double ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            kernel<<<cus, threads>>>(table, bias, n_rows, iters, out);
            (void)hipEventRecord(e1);
            ms = time_ms(e0, e1);
        }
        const double r = (double)cus * (threads / 16) * iters;
        printf("rows  %-64s %4d thr: %7.3f ms  %6.2f G rows/s  %6.2f us per row step\n", name, threads, ms, r / ms * 1e-6, ms * 1e3 / iters);
        (void)rows;
    };
    run("serial (gather, wait, 300 VALU, atomics)", row_kernel<0, 300>, 1024);
    run("gathers one row ahead, vmcnt(0) per row", row_kernel<1, 300>, 1024);
    run("gathers one row ahead, counted wait (atomics stay in flight)", row_kernel<2, 300>, 1024);
    run("serial, no atomics", row_kernel<3, 300>, 1024);
    run("gathers one row ahead, no atomics", row_kernel<4, 300>, 1024);
    run("serial, 600 VALU", row_kernel<0, 600>, 1024);
    run("ahead + counted, 600 VALU", row_kernel<2, 600>, 1024);
    run("serial no atomics, 600 VALU", row_kernel<3, 600>, 1024);
    run("ahead no atomics, 600 VALU", row_kernel<4, 600>, 1024);
    run("serial, 100 VALU", row_kernel<0, 100>, 1024);
    run("ahead + counted, 100 VALU", row_kernel<2, 100>, 1024);
    run("ahead no atomics, 100 VALU", row_kernel<4, 100>, 1024);
    run("serial, 300 VALU, 8 waves per CU", row_kernel<0, 300>, 512);
    run("ahead + counted, 300 VALU, 8 waves per CU", row_kernel<2, 300>, 512);
    run("two rows ahead (bias atomics in front of the gathers), 300 VALU", row_kernel<6, 300>, 1024);
    run("two rows ahead, 600 VALU", row_kernel<6, 600>, 1024);
    run("two rows ahead, 100 VALU", row_kernel<6, 100>, 1024);
    run("atomics deferred one row, 300 VALU", row_kernel<5, 300>, 1024);
    unsigned *err; (void)hipMalloc(&err, 4); (void)hipMemset(err, 0, 4);
    auto run_w = [&](const char *name, auto kernel, int n_comp) {
        double ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            kernel<<<cus, 1024>>>(table, bias, n_rows, iters, out, err);
            (void)hipEventRecord(e1);
            ms = time_ms(e0, e1);
        }
        unsigned h = 0; (void)hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
        const double r = (double)cus * (n_comp * 4) * iters;
        printf("rows  %-64s %4d thr: %7.3f ms  %6.2f G rows/s  %6.2f us per row step  (err %u)\n", name, 1024, ms, r / ms * 1e-6, ms * 1e3 / iters, h);
    };
    run_w("writer wavefront: 15 compute + 1 writer, 300 VALU", row_kernel_w<300, 1, true>, 15);
    run_w("writer wavefronts: 14 compute + 2 writers, 300 VALU", row_kernel_w<300, 2, true>, 14);
    run_w("writer wavefront, 600 VALU", row_kernel_w<600, 1, true>, 15);
    run_w("2 writers, 600 VALU", row_kernel_w<600, 2, true>, 14);
    run_w("writer wavefront, 100 VALU", row_kernel_w<100, 1, true>, 15);
    run_w("2 writers, 100 VALU", row_kernel_w<100, 2, true>, 14);
    run_w("2 writers that drop the atomics, 300 VALU", row_kernel_w<300, 2, false>, 14);
    return 0;
}
