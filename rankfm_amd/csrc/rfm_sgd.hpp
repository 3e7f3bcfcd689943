// rfm_sgd.hpp -- the BPR/WARP SGD wavefront kernel of the MI355X RankFM engine (gfx950 only).
//
// Replaces the reference's sequential row loop rankfm/_rankfm.pyx:230-326 (one SGD step per shuffled
// interaction: score positive, draw/score negative(s), sigmoid gradient, in-place update of
// w_i, w_if, v_u, v_i, v_uf, v_if) together with compute_ui_utility (:48-89), the rejection sampler
// (:250-253 + lsearch :20-27) and the WARP early-exit loop (:244-270).
//
// Mapping onto CDNA4:
//   * A "row group" of G lanes (G = 4..64, power of two) owns one interaction; lane s of the group owns
//     factor chunks c = s, s+G, ... of VEC floats (VEC = 4 -> one 16-byte load per chunk, so a k=64 row
//     is one fully coalesced 256-byte segment read by 16 lanes; VEC = 1 for factor counts that are not
//     a multiple of 4).  A 64-lane wavefront therefore carries 64/G interactions at once.
//   * The k-dimension dot products are xor-butterfly reductions inside the group (DPP / ds_bpermute);
//     every lane ends with the bit-identical sum, so the WARP control flow is group-uniform.
//   * Negative draws are counter based (include/rfm_rng.h): no shared RNG state, any lane can draw.
//   * HOGWILD mode: factor/bias updates are fp32 hardware atomics (global_atomic_add_f32), computed as
//     delta = eta * (grad - 2*reg*w_loaded) from the values the step loaded (the update is not a pure
//     add, it reads w).  SERIAL mode: ONE wavefront, ONE group active, plain
//     read-modify-write in the reference's exact order -- the reference's sequential semantics, used
//     by the parity tests against the golden vectors and for debugging.
//   * No MFMA anywhere: ~2 flops per 4-byte factor element; this is an HBM/L2 gather-scatter.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rfm_rng.h"

namespace rfm {

struct SgdArgs {
    const int32_t *__restrict__ interactions;   // [N,2]
    const float *__restrict__ sample_weight;    // [N]
    const int64_t *__restrict__ csr_off;        // [U+1]
    const int32_t *__restrict__ csr_items;      // [nnz]
    const float *__restrict__ x_uf;             // [U,P]
    const float *__restrict__ x_if;             // [I,Q]
    float *w_i, *w_if, *v_u, *v_i, *v_uf, *v_if;
    const int32_t *__restrict__ perm;           // this epoch's visiting order [N] or nullptr
    const float *__restrict__ multiplier;       // [max_samples+1]: log((I-1)/s)/log(I), s = 1..max_samples (host, double)
    uint32_t *mt_state;                         // [625] MT19937 words + index (serial + MT only)
    double *ll;                                 // this epoch's log-likelihood accumulator
    unsigned long long *draws;                  // this epoch's accepted-draw counter
    unsigned int *error_flags;                  // bit 0: rejection sampler gave up
    int64_t pos_begin, pos_end;                 // positions of the epoch handled by this launch
    int64_t n_rows;                             // N
    int32_t n_items, n_uf, n_if, n_factors;     // I, P, Q, F
    int32_t has_uf, has_if;
    int32_t max_samples;
    int32_t rng;                                // RFM_RNG_*
    uint32_t epoch_key, perm_bits;
    float eta, reg_a, reg_b;                    // learning rate of the epoch, 2*alpha, 2*beta
    // Hogwild step damping (DESIGN.md "staleness"): a row that n in-flight updates touch at once receives n steps computed
    // from the same stale value; above ~M of them the combined step overshoots.  The step on such a row is scaled by
    // min(1, M / n), with n = in-flight rows x the row's share of the data.  All 1 / null in serial mode.
    const float *__restrict__ pos_scale;        // [I] scale for the positive item's row (by item popularity), or nullptr
    float user_cap;                             // a user of degree d gets min(1, user_cap / d)
    float feat_scale;                           // scale for the dense feature tables (every row touches them)
    int32_t update_mode;                        // hogwild experiments: 0 all atomics, 1 v_u plain RMW, 2 everything plain RMW
};

constexpr float kMargin = 1.0f;                 // rankfm/_rankfm.pyx:149
constexpr uint32_t kMaxAttempts = 1u << 22;     // safety net; the host rejects saturated users up front

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) x += __shfl_xor(x, m);
    return x;
}

template <int VEC>
__device__ __forceinline__ void load_chunk(const float *p, float (&r)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else {
        r[0] = *p;
    }
}

template <int VEC>
__device__ __forceinline__ void store_chunk(float *p, const float (&r)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
        *p = r[0];
    }
}

// fp32 hardware atomic add, no return value (global_atomic_add_f32)
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

template <bool SERIAL, int VEC>
__device__ __forceinline__ void apply_chunk(float *p, const float (&oldv)[VEC], const float (&delta)[VEC], bool plain = false) {
    if (SERIAL || plain) {
        float n[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) n[e] = oldv[e] + delta[e];
        store_chunk<VEC>(p, n);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) atomic_add_f32(p + e, delta[e]);
    }
}

// membership of `item` in the user's sorted list: the predicate of lsearch (rankfm/_rankfm.pyx:20-27),
// evaluated by binary search
__device__ __forceinline__ bool is_member(const int32_t *__restrict__ items, int64_t lo, int64_t hi, int32_t item) {
    while (lo < hi) {
        const int64_t md = lo + ((hi - lo) >> 1);
        const int32_t v = items[md];
        if (v == item) return true;
        if (v < item) lo = md + 1; else hi = md;
    }
    return false;
}

// MT19937 step on a state kept in global memory (serial mode, one lane).  Published algorithm of
// Matsumoto & Nishimura; the reference vendors it as rankfm/mt19937ar/mt19937ar.c:105-140.
__device__ inline uint32_t mt_next_global(uint32_t *st) {
    uint32_t idx = st[624];
    if (idx >= 624u) {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (st[k] & 0x80000000u) | (st[(k + 1) % 624] & 0x7fffffffu);
            st[k] = st[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t y = st[idx];
    st[624] = idx + 1;
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ __forceinline__ float log_sigmoid(float x) {
    // log(1 / (1 + exp(-x)))  (rankfm/_rankfm.pyx:270), overflow-free form
    return fminf(x, 0.0f) - log1pf(__expf(-fabsf(x)));
}

// ---------------------------------------------------------------------------------------------
// one SGD step for one interaction, executed by the G lanes of a row group
// ---------------------------------------------------------------------------------------------
template <int VEC, int G, int KPL, bool SERIAL, bool FEAT>
struct RowStep {
    static constexpr int CH = KPL;   // chunks per lane

    const SgdArgs &a;
    const int sub;                   // lane index inside the group
    const int F;

    __device__ __forceinline__ RowStep(const SgdArgs &args, int sub_) : a(args), sub(sub_), F(args.n_factors) {}

    __device__ __forceinline__ int chunk_f(int k) const { return (sub + G * k) * VEC; }
    __device__ __forceinline__ bool chunk_ok(int k) const { return chunk_f(k) < F; }

    __device__ __forceinline__ void load_row(const float *base, float (&r)[KPL][VEC]) const {
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            if (chunk_ok(k)) {
                load_chunk<VEC>(base + chunk_f(k), r[k]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) r[k][e] = 0.0f;
            }
        }
    }

    // acc[f] = sum_r x[r] * table[r, f]   (feature projection into factor space, this lane's chunks)
    __device__ __forceinline__ void project(const float *__restrict__ x, int n, const float *table,
                                            float (&acc)[KPL][VEC]) const {
#pragma unroll
        for (int k = 0; k < KPL; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[k][e] = 0.0f;
        for (int r = 0; r < n; ++r) {
            const float xr = x[r];
            if (xr == 0.0f) continue;     // zero entries contribute nothing (and are skipped by the reference, :73,:81)
            float t[KPL][VEC];
            load_row(table + (size_t)r * F, t);
#pragma unroll
            for (int k = 0; k < KPL; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[k][e] += xr * t[k][e];
        }
    }

    // compute_ui_utility (rankfm/_rankfm.pyx:48-89) for item `it` given the user-side registers:
    //   w_i[it] + sum_q x_if[it,q] w_if[q] + sum_f [ (vu_f + A_f) * vi_f + B_f(it) * vu_f ]
    // A = x_uf[u] . v_uf  (user-feature projection), B(it) = x_if[it] . v_if  (item-feature projection)
    __device__ __forceinline__ float utility(const float (&vu)[KPL][VEC], const float (&A)[KPL][VEC], int32_t it,
                                             float (&vi)[KPL][VEC], float (&B)[KPL][VEC], float &wi) const {
        load_row(a.v_i + (size_t)it * F, vi);
        wi = a.w_i[it];
        float part = 0.0f, scalar = 0.0f;
        if constexpr (FEAT) {
            if (a.has_if) {
                const float *xi = a.x_if + (size_t)it * a.n_if;
                project(xi, a.n_if, a.v_if, B);
                for (int q = 0; q < a.n_if; ++q) scalar += xi[q] * a.w_if[q];
            } else {
#pragma unroll
                for (int k = 0; k < KPL; ++k)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) B[k][e] = 0.0f;
            }
#pragma unroll
            for (int k = 0; k < KPL; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) part += (vu[k][e] + A[k][e]) * vi[k][e] + B[k][e] * vu[k][e];
        } else {
#pragma unroll
            for (int k = 0; k < KPL; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) part += vu[k][e] * vi[k][e];
        }
        return wi + scalar + group_sum<G>(part);
    }

    // draw the next unobserved item for the user (rankfm/_rankfm.pyx:250-253)
    __device__ __forceinline__ int32_t next_negative(int64_t lo, int64_t hi, uint32_t row_key, uint32_t &attempt) const {
        int32_t j = 0;
        if (SERIAL && a.rng == 0 /* RFM_RNG_MT19937 */) {
            if (sub == 0) {
                do { j = (int32_t)(mt_next_global(a.mt_state) % (uint32_t)a.n_items); } while (is_member(a.csr_items, lo, hi, j));
            }
            j = __shfl(j, (threadIdx.x & 63) - sub);
        } else {
            for (;;) {
                j = (int32_t)rfm_draw_to_item(rfm_draw(row_key, attempt), (uint32_t)a.n_items);
                ++attempt;
                if (!is_member(a.csr_items, lo, hi, j)) break;
                if (attempt >= kMaxAttempts) { if (sub == 0) atomicOr(a.error_flags, 1u); break; }
            }
        }
        return j;
    }

    __device__ __forceinline__ void operator()(int64_t row, double &ll_acc, unsigned &draw_acc) const {
        const int32_t u = a.interactions[2 * row];                       // :233-235
        const int32_t i = a.interactions[2 * row + 1];
        const float sw = a.sample_weight[row];                           // :236
        const int64_t lo = a.csr_off[u], hi = a.csr_off[u + 1];
        const uint32_t row_key = rfm_row_key(a.epoch_key, (uint32_t)row);
        uint32_t attempt = 0;

        float vu[KPL][VEC], A[KPL][VEC];
        load_row(a.v_u + (size_t)u * F, vu);
        if constexpr (FEAT) {
            if (a.has_uf) project(a.x_uf + (size_t)u * a.n_uf, a.n_uf, a.v_uf, A);
            else {
#pragma unroll
                for (int k = 0; k < KPL; ++k)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) A[k][e] = 0.0f;
            }
        }

        float vi[KPL][VEC], Bi[KPL][VEC], wi;
        const float ut_ui = utility(vu, A, i, vi, Bi, wi);               // :239

        // WARP sampling loop (:244-264); BPR is max_samples == 1
        float vj[KPL][VEC], Bj[KPL][VEC], wj = 0.0f;
        float min_pu = 1e6f;
        int32_t j = -1;
        int sampled = 0;
        for (int s = 1; s <= a.max_samples; ++s) {
            const int32_t cand = next_negative(lo, hi, row_key, attempt);
            float vc[KPL][VEC], Bc[KPL][VEC], wc;
            const float pu = ut_ui - utility(vu, A, cand, vc, Bc, wc);   // :256-257
            sampled = s;
            if (pu < min_pu || j < 0) {                                   // :259-261 (j < 0: keep a valid index under NaN)
                if (pu < min_pu) min_pu = pu;
                j = cand; wj = wc;
#pragma unroll
                for (int k = 0; k < KPL; ++k)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { vj[k][e] = vc[k][e]; if constexpr (FEAT) Bj[k][e] = Bc[k][e]; }
            }
            if (pu < kMargin) break;                                      // :263-264
        }
        const float pu = min_pu;                                          // :267-268
        const float multiplier = a.multiplier[sampled];                   // :269 (integer division inside the log)
        if (sub == 0) { ll_acc += (double)log_sigmoid(pu); draw_acc += (unsigned)sampled; }   // :270
        const float d_outer = 1.0f / (__expf(pu) + 1.0f);                 // :276
        const float g = sw * multiplier;
        const float eta = a.eta, reg_a = a.reg_a, reg_b = a.reg_b;
        float eta_u = eta, eta_i = eta, eta_f = eta;
        if constexpr (!SERIAL) {
            eta_u = eta * fminf(1.0f, a.user_cap / (float)(hi - lo));
            if (a.pos_scale) eta_i = eta * a.pos_scale[i];
            eta_f = eta * a.feat_scale;
        }

        // item biases (:279-280) -- one lane per group
        if (sub == 0) {
            const float dwi = eta_i * (g * (d_outer * 1.0f) - reg_a * wi);
            const float dwj = eta * (g * (d_outer * -1.0f) - reg_a * wj);
            if constexpr (SERIAL) { a.w_i[i] = wi + dwi; a.w_i[j] = wj + dwj; }
            else { atomic_add_f32(a.w_i + i, dwi); atomic_add_f32(a.w_i + j, dwj); }
        }

        // item-feature weights (:283-286): every q shrinks, lanes split the q range
        if constexpr (FEAT) {
            if (a.has_if) {
                const float *xi = a.x_if + (size_t)i * a.n_if, *xj = a.x_if + (size_t)j * a.n_if;
                for (int q = sub; q < a.n_if; q += G) {
                    const float w = a.w_if[q];
                    const float d = eta_f * (g * (d_outer * (xi[q] - xj[q])) - reg_b * w);
                    if constexpr (SERIAL) a.w_if[q] = w + d; else atomic_add_f32(a.w_if + q, d);
                }
            }
        }

        // factor updates (:289-326), this lane's chunks
        float nvu[KPL][VEC], dij[KPL][VEC];     // updated v_u, updated (v_i - v_j)
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            float d_u[VEC], d_i[VEC], d_j[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float g_u = vi[k][e] - vj[k][e];                          // :292
                float g_i = vu[k][e];                                     // :293-294 (d_v_j = -d_v_i)
                if constexpr (FEAT) { g_i += A[k][e]; g_u += Bi[k][e] - Bj[k][e]; }   // :297-305
                d_u[e] = eta_u * (g * (d_outer * g_u) - reg_a * vu[k][e]);  // :308
                d_i[e] = eta_i * (g * (d_outer * g_i) - reg_a * vi[k][e]);  // :309
                d_j[e] = eta * (g * (d_outer * -g_i) - reg_a * vj[k][e]); // :310
                nvu[k][e] = vu[k][e] + d_u[e];
                dij[k][e] = (vi[k][e] + d_i[e]) - (vj[k][e] + d_j[e]);
            }
            if (chunk_ok(k)) {
                const int f0 = chunk_f(k);
                apply_chunk<SERIAL, VEC>(a.v_u + (size_t)u * F + f0, vu[k], d_u, a.update_mode >= 1);
                apply_chunk<SERIAL, VEC>(a.v_i + (size_t)i * F + f0, vi[k], d_i, a.update_mode >= 2);
                apply_chunk<SERIAL, VEC>(a.v_i + (size_t)j * F + f0, vj[k], d_j, a.update_mode >= 2);
            }
        }

        if constexpr (FEAT) {
            // user-feature factors (:313-318): rows p with x_uf[u,p] != 0, using the UPDATED v_i[i]-v_i[j]
            if (a.has_uf) {
                const float *xu = a.x_uf + (size_t)u * a.n_uf;
                for (int p = 0; p < a.n_uf; ++p) {
                    const float xp = xu[p];
                    if (xp == 0.0f) continue;
                    float *trow = a.v_uf + (size_t)p * F;
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        if (!chunk_ok(k)) continue;
                        float t[VEC], d[VEC];
                        load_chunk<VEC>(trow + chunk_f(k), t);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) d[e] = eta_f * (g * (d_outer * (xp * dij[k][e])) - reg_b * t[e]);
                        apply_chunk<SERIAL, VEC>(trow + chunk_f(k), t, d);
                    }
                }
            }
            // item-feature factors (:321-326): rows q with x_if[i,q] != x_if[j,q], using the UPDATED v_u[u]
            if (a.has_if) {
                const float *xi = a.x_if + (size_t)i * a.n_if, *xj = a.x_if + (size_t)j * a.n_if;
                for (int q = 0; q < a.n_if; ++q) {
                    const float dx = xi[q] - xj[q];
                    if (dx == 0.0f) continue;
                    float *trow = a.v_if + (size_t)q * F;
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        if (!chunk_ok(k)) continue;
                        float t[VEC], d[VEC];
                        load_chunk<VEC>(trow + chunk_f(k), t);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) d[e] = eta_f * (g * (d_outer * (dx * nvu[k][e])) - reg_b * t[e]);
                        apply_chunk<SERIAL, VEC>(trow + chunk_f(k), t, d);
                    }
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// the kernel: every wavefront walks the epoch's positions with a grid stride of (waves * rows-per-wave)
// ---------------------------------------------------------------------------------------------
template <int VEC, int G, int KPL, bool SERIAL, bool FEAT>
__global__ void __launch_bounds__(256) sgd_kernel(const SgdArgs a) {
    constexpr int RPW = SERIAL ? 1 : 64 / G;                    // interactions carried by one wavefront at a time
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, sub = lane % G;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const RowStep<VEC, G, KPL, SERIAL, FEAT> step(a, sub);

    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    for (int64_t pos0 = a.pos_begin + wave * RPW; pos0 < a.pos_end; pos0 += n_waves * RPW) {
        const int64_t pos = pos0 + (SERIAL ? 0 : grp);
        const bool active = (pos < a.pos_end) && (!SERIAL || grp == 0);
        if (active) {
            const int64_t row = a.perm ? (int64_t)a.perm[pos]
                                       : (int64_t)rfm_perm((uint32_t)pos, (uint32_t)a.n_rows, a.perm_bits, a.epoch_key);
            step(row, ll_acc, draw_acc);
        }
    }
    // wavefront reduction of the log-likelihood / draw counters, one atomic each per wavefront
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        ll_acc += __shfl_xor(ll_acc, m);
        draw_acc += __shfl_xor(draw_acc, m);
    }
    if (lane == 0) {
        if (ll_acc != 0.0) unsafeAtomicAdd(a.ll, ll_acc);
        if (draw_acc) atomicAdd(a.draws, (unsigned long long)draw_acc);
    }
}

// host-side launcher table (rfm_sgd_inst_*.hip)
typedef void (*sgd_launch_fn)(const SgdArgs &, int grid, hipStream_t);
struct SgdShape { int vec, group, kpl; };

}  // namespace rfm
