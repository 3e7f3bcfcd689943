// rfm_sgd_features.hpp -- models with user / item features: the tables kernel (trainer + step producers) and the row loops beside it.
#pragma once
#include "rfm_rowstep.hpp"

namespace rfm {

// ---------------------------------------------------------------------------------------------
// features kernel (production Hogwild for models with user / item features)
//
// The dense feature tables v_uf [P,F], v_if [Q,F], w_if [Q] are touched by EVERY update (rankfm/_rankfm.pyx:283-286, 313-326),
// and each touch shrinks the touched rows by 2 beta eta: in the sequential algorithm they are an exponential moving average of
// the last ~1 / (2 beta eta) = 50-170 updates' gradients, i.e. they forget within a tiny fraction of an epoch.  16 k
// interactions in flight cannot share such rows Hogwild-style (thousands of stale shrinks diverge), and per-workgroup replicas
// that evolve independently and are merged now and then drift apart (measured: profiles/r02_notes.md).  So the tables are trained
// by ONE sequential stream and read, coherently, by everybody.  Roles by workgroup index:
//   * 0: the TABLE TRAINER.  It applies the reference's table updates (:283-286, :313-326) of a stream of interactions in order
//     on a master copy in its LDS -- all table rows in parallel, one row group per table row with the row in registers (the rows
//     of the tables do not read each other), walking only the interactions that touch the row -- and publishes the copy to the
//     weight arrays after every batch.  The interactions' steps come to it ready-made:
//   * 1 .. n_producers: STEP PRODUCERS.  Each row group samples an interaction of the rank's data at random, scores it exactly
//     like a regular step and stages the step's g * d_outer, updated v_u, updated v_i - v_j, x_uf[u] and x_if[i] - x_if[j]
//     WITHOUT storing any row (the rows are trained when their own turn comes); a batch of one staged step per row group goes to
//     the trainer through a double-buffered slot in memory.  A step is a chain of ~5 dependent gathers (~15 us), applying 64 of
//     them takes ~2 us: round 2's trainer produced its own steps and so managed every ~150th row of the stream, which showed in
//     the first epoch from random weights (the item biases picked up what the tables carry in the reference); with the steps
//     produced beside it the trainer's rate is its apply rate -- a sequential SGD stream on a uniform sample of every ~20th-40th
//     row, the same process that drives the tables in the reference, with the same memory and the same noise level.
//   * the rest: the asynchronous ROW LOOP (user segments, v_u in registers, atomics for v_i / w_i) with the tables as a READ-ONLY
//     copy in the workgroup's LDS that its wavefronts keep refreshing, a slice per wavefront and row (system-scope loads: the
//     per-XCD L2s are not coherent, and a 16 KB table that is re-read all the time would otherwise never leave them).  No
//     lock-step, no barrier in the loop.  BPR models with at most 32 + 32 features on 16-lane row groups take the pipelined form
//     (FeatFast below), everything else the generic RowStep.
// One group alone (debug_flags bit 0) does everything in the reference's order -- the sequential form the parity tests pin; with
// debug_flags bit 5 (tables frozen: no trainer, no producers) one group alone runs the pipelined row loop sequentially, which
// pins THAT code to the oracle as well.
// ---------------------------------------------------------------------------------------------
#define RFM_REP8(X, O) X(O + 0) X(O + 1) X(O + 2) X(O + 3) X(O + 4) X(O + 5) X(O + 6) X(O + 7)
typedef float rfm_f4 __attribute__((ext_vector_type(4)));
typedef float rfm_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) rfm_f4 lds_f4;
typedef __attribute__((address_space(3))) rfm_f2 lds_f2;

// flags of one launch of the features kernel (SgdArgs::feat_flags, zero between launches)
constexpr int kFeatMaxProducers = 16;
constexpr int kFeatReady = 0;                                  // [2 * producers] batches written into each slot
constexpr int kFeatConsumed = 2 * kFeatMaxProducers;           // [2 * producers] batches the trainer has taken out of each slot
constexpr int kFeatStop = 4 * kFeatMaxProducers;               // the regular workgroups are done
constexpr int kFeatExited = kFeatStop + 1;                     // producers that have left
constexpr int kFeatDone = kFeatStop + 2;                       // (unused since the trainer works to a fixed quota)
constexpr int kFeatFlagWords = kFeatStop + 4;
constexpr unsigned kFeatSpinLimit = 1u << 23;                  // polls (~0.5 us each) before a waiting workgroup gives up: seconds

// (The trainer and the producers are non-inlined functions that receive SgdArgs by value: the compiler no longer knows that its
// pointers are global memory, and a FLAT load counts against the LDS counter as well -- every LDS wait of the trainer's apply loop
// waited for the batch in flight (measured: 17 us per batch instead of 6).  Their hot pointers are therefore cast to the global
// address space explicitly.)
typedef __attribute__((address_space(1))) float g_float;
typedef __attribute__((address_space(1))) unsigned int g_uint;
__device__ __forceinline__ unsigned flag_load(const g_uint *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void flag_store(g_uint *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// The pipelined row loop keeps its LDS copy of v_uf / v_if LANE-MAJOR: the KPL factor dwords lane s of a row group owns (s, s + 16,
// ...) are consecutive, so a table row costs the lane one 16-byte LDS read instead of KPL 4-byte ones.  Row stride 16 * KPL.
template <int KPL>
__device__ __forceinline__ void lds_row_load(const lds_float *p, float (&t)[KPL]) {
    if constexpr (KPL % 4 == 0) {
#pragma unroll
        for (int q = 0; q < KPL / 4; ++q) {
            const rfm_f4 v = *(const lds_f4 *)(p + 4 * q);
            t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
        }
    } else if constexpr (KPL % 2 == 0) {
#pragma unroll
        for (int q = 0; q < KPL / 2; ++q) {
            const rfm_f2 v = *(const lds_f2 *)(p + 2 * q);
            t[2 * q] = v.x; t[2 * q + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KPL; ++k) t[k] = p[k];
    }
}

// acc[k] += sum_t x[t] * table[t][this lane's dwords], t = 0 .. n-1 (n <= 32): x is held across the 16 lanes of the group (lane s:
// x[s] in xr0, x[s + 16] in xr1; entries >= n are zero) and reaches all lanes through a DPP row_share -- one VALU move per tag, no
// ballot / shuffle walk over the non-zero entries (zero entries add an exact zero; the reference skips them, :73, :81).  The table
// is padded with zero rows to a multiple of 8.
template <int KPL>
__device__ __forceinline__ void project_dense(float xr0, float xr1, int n, const lds_float *tab_lane, float (&acc)[KPL]) {
    constexpr int FS = 16 * KPL;
#define RFM_PSTEP(B)                                                                      \
    {                                                                                     \
        const float x = dpp_mov<0x150 + ((B) & 15)>(((B) < 16) ? xr0 : xr1);              \
        float t[KPL];                                                                     \
        lds_row_load<KPL>(tab_lane + (B) * FS, t);                                        \
        _Pragma("unroll") for (int k = 0; k < KPL; ++k) acc[k] = __builtin_fmaf(x, t[k], acc[k]); \
    }
    if (n > 0) { RFM_REP8(RFM_PSTEP, 0) }
    if (n > 8) { RFM_REP8(RFM_PSTEP, 8) }
    if (n > 16) { RFM_REP8(RFM_PSTEP, 16) }
    if (n > 24) { RFM_REP8(RFM_PSTEP, 24) }
#undef RFM_PSTEP
}

// locals every role of the features kernel derives from the launch (trainer, producers, row loops)
#define RFM_FEAT_LOCALS                                                                                                                   \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;                                                \
    const int sub = lane % G;                                                                                                             \
    const int F = a.n_factors;                                                                                                            \
    lds_float *lds_tables = lds;                                                                                                          \
    const int n_uf_f = a.n_uf * F, n_if_f = a.n_if * F, n_tab = n_uf_f + n_if_f + a.n_if;                                                  \
    auto table_ptr = [&](int k) { return (g_float *)(k < n_uf_f ? a.v_uf + k : (k < n_uf_f + n_if_f ? a.v_if + (k - n_uf_f) : a.w_if + (k - n_uf_f - n_if_f))); }; \
    const int NP = a.single_group ? 0 : a.n_producers;               /* (one group alone: no trainer, no producers) */                    \
    const bool trains = !a.single_group && !a.feat_frozen;                                                                                \
    g_uint *flags = (g_uint *)a.feat_flags;                                                                                               \
    const int gid = threadIdx.x / G, gpb = blockDim.x / G;                                                                                \
    const int n_slot = 1 + 2 * F + a.n_uf + a.n_if;                  /* staged step of one interaction (RowStep::stage) */                \
    const size_t batch_floats = (size_t)gpb * n_slot;                                                                                     \
    const int n_regular = (int)gridDim.x;                            /* (row-loop kernels: every workgroup walks rows) */                 \
    (void)lane; (void)wave; (void)n_waves; (void)sub; (void)gid; (void)flags; (void)batch_floats; (void)n_regular; (void)lds_tables; (void)table_ptr; \
    (void)trains; (void)NP; (void)n_tab;

// The roles of the features kernel other than the pipelined row loop are separate (non-inlined) functions: each gets a register
// allocation of its own, so that the trainer's batch in flight or the generic step's feature vectors do not cost the row loop spills.
// share of a launch's segments handed out to the row loops after which the table trainer stops (SgdArgs::table_quiet_from; see the
// trainer).  Set by the host ONLY for a caller's quota denser than the default: the default quota is done at ~0.78 of the row loops'
// time (~0.85 of the segments handed out -- the hand-out runs ahead of the rows by the segments in flight and a chunk per workgroup) and
// must keep its exact, repeatable step count; the opening launch of a fit, whose trainer is the slower side by design, has no stop either.
constexpr float kTableQuietFrom = 0.8f;

template <int G, int KPL>
__device__ __forceinline__ void feat_table_trainer(const SgdArgs &a, lds_float *lds, lds_int *s_stop_p) {
    RFM_FEAT_LOCALS
    // natural layout of the tables: [P, F] | [Q, F] | [Q]
    for (int k = threadIdx.x; k < n_tab; k += blockDim.x) lds_tables[k] = *table_ptr(k);
    // The staging area holds a batch in the TRAINER'S layout: per staged step [updated v_u | updated v_i - v_j | g * d_outer | x_uf[u] |
    // x_if[i] - x_if[j]], the two vectors lane-major and padded to the group width (lane s of a row group reads its KPL dwords with
    // one 16-byte LDS read, no per-dword bounds predicate), slots padded to a multiple of four floats.
    constexpr int FS = G * KPL;
    const int NSL = (2 * FS + 1 + a.n_uf + a.n_if + 3) & ~3;
    lds_float *stage = lds + ((n_tab + 3) & ~3);
    __syncthreads();
    // ---- the table trainer ------------------------------------------------------------------------------------------
    const float eta_f = a.eta * a.table_step, reg_b = a.reg_b;
    // rho^n, n = 0 .. gpb: w_if shrinks on EVERY interaction (:283-286), also those whose tag difference is zero, which the
    // row walk below skips
    lds_float *rho_pow = stage + (size_t)gpb * NSL;
    if (threadIdx.x <= (unsigned)gpb) rho_pow[threadIdx.x] = powf(1.0f - eta_f * reg_b, (float)threadIdx.x);
    for (size_t k = threadIdx.x; k < (size_t)gpb * NSL; k += blockDim.x) stage[k] = 0.0f;     // (the padding is read, never written)
    // The loop is software-pipelined: while batch q - 1 is being applied out of LDS, batch q is on its way from memory into
    // registers and the ready flag of batch q + 1 is being polled, so that a batch costs the trainer its apply time and two
    // barriers instead of three dependent memory round trips (flag, data, publication).
    constexpr int kPre = 14;                                  // dwords of a batch a thread keeps in flight
    float pre[kPre];
    const unsigned n_batch = (unsigned)batch_floats, n_threads = blockDim.x;
    // where dword threadIdx.x + j * blockDim.x of a batch (the producers' layout: RowStep::stage, [g d_outer | v_u | v_i - v_j | x_uf |
    // x_if diff] per step) goes in the staging area; the same for every batch, so computed once
    auto stage_index = [&](size_t k) {
        const int s2 = (int)(k / (size_t)n_slot), off = (int)(k - (size_t)s2 * n_slot);
        int dst;
        if (off == 0) dst = 2 * FS;
        else if (off < 1 + 2 * F) {
            const int v = off - 1 < F ? 0 : 1, f = off - 1 - v * F;
            dst = v * FS + (f % G) * KPL + f / G;
        } else dst = 2 * FS + 1 + (off - 1 - 2 * F);
        return s2 * NSL + dst;
    };
    // (two 16-bit staging indexes per register; 0xFFFF = none.  Register pressure matters here: a spilled value reloaded between two
    // of the batch's loads waits for every load issued before it -- the system-scope loads return in order)
    unsigned pre_dst[(kPre + 1) / 2];
#pragma unroll
    for (int j = 0; j < kPre; j += 2) {
        const unsigned k0 = threadIdx.x + (unsigned)j * n_threads, k1 = k0 + n_threads;
        const unsigned d0 = k0 < n_batch ? (unsigned)stage_index(k0) : 0xFFFFu, d1 = (j + 1 < kPre && k1 < n_batch) ? (unsigned)stage_index(k1) : 0xFFFFu;
        pre_dst[j / 2] = d0 | (d1 << 16);
    }
    const bool prefetch_ok = (size_t)gpb * NSL < 0xFFFFu;    // (else every dword of a batch takes the direct path below)
    unsigned applied = 0;
    // The trainer applies a FIXED number of staged steps per launch: the host's quota (rows of the launch / the pace a trainer keeps
    // beside that many row-loop workgroups, rfm_api.hip), in whole batches, at least one.  It does not look at the row loops: the
    // step count -- and with it the tables a given launch geometry produces -- no longer depends on timing.
    const unsigned quota = (unsigned)(((a.table_quota > (int64_t)gpb ? a.table_quota : (int64_t)gpb) + gpb - 1) / gpb) * (unsigned)gpb;
    bool staged = false;                                      // LDS holds a batch that has not been applied yet
    unsigned flag_next = 0;                                   // (thread 0) the ready counter of the next batch, loaded ahead
    auto slot_of = [&](unsigned q, int &p, unsigned &par, unsigned &m) {
        p = NP > 0 ? (int)(q % (unsigned)NP) : 0;
        // (ONE slot per producer: a staged step is scored on the tables of its time, and every batch that waits in a slot is a batch
        // of stale steps -- with two slots each, three producers cost the 3000 x 2000 feature fixture 0.6 point of hit_rate@10 against
        // one: profiles/r03_notes.md.  The second slot of the ring stays unused.)
        const unsigned n_p = NP > 0 ? q / (unsigned)NP : q;
        par = 0u; m = n_p;
    };
    if (threadIdx.x == 0 && NP > 0) flag_next = __hip_atomic_load(flags + kFeatReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (diagnostics, rfm_fit_report.feat_diag: time this workgroup waited for a batch / ran in all, in 100 MHz ticks)
    const unsigned long long t_begin = wall_clock64();
    unsigned long long t_wait = 0, t_seg[4] = {0, 0, 0, 0};      // (apply | publish | batch into LDS | slot release)
    for (unsigned q = 0;; ++q) {
        int p;
        unsigned par, m;
        slot_of(q, p, par, m);
        if (threadIdx.x == 0) {
            int stop = applied >= quota ? 1 : 0;
            // Safety net (round 5, quotas denser than the default): the trainer stops once the row loops have been handed table_quiet_from of
            // the launch's segments -- a quota denser than the trainer's pace used to keep it running to the launch's end and beyond
            // (every 223rd row on config 4's share: a 5.8 ms tables kernel beside 4.0 ms of row loops), which leaves the rows no
            // quiet period against tables that have stopped moving (profiles/r04_notes.md section 11: -3.8 points of hit_rate@10).
            // The counter of segments handed out is
            // only ever written by memory-side atomics: read through one (a load would be served from this XCD's L2).
            if (!stop && a.table_quiet_from > 0.0f && a.tickets != nullptr && a.pos_end > a.pos_begin) {
                const unsigned handed = __hip_atomic_fetch_add(a.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((double)handed >= (double)a.table_quiet_from * (double)(a.pos_end - a.pos_begin)) stop = 1;
            }
            const unsigned long long t0 = wall_clock64();
            for (unsigned spin = 0; !stop && (NP == 0 || flag_next < m + 1u); ++spin) {
                if (spin > kFeatSpinLimit) { atomicOr(a.error_flags, 8u); stop = 1; break; }      // (never observed: a hang guard)
                __builtin_amdgcn_s_sleep(4);
                if (NP > 0) flag_next = __hip_atomic_load(flags + kFeatReady + 2 * p + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            t_wait += wall_clock64() - t0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       // the batch's data is read after its flag
            *s_stop_p = stop;
        }
        __syncthreads();
        if (*s_stop_p) break;
        const unsigned long long tA = wall_clock64();
        // batch q: on its way into registers (the part beyond kPre dwords per thread goes straight to LDS below)
        const g_float *src = (const g_float *)a.feat_ring + (size_t)(2 * p + par) * batch_floats;
        {
            const g_float *pp = src + threadIdx.x;
            unsigned k = threadIdx.x;
#pragma unroll
            for (int j = 0; j < kPre; ++j) {
                pre[j] = (prefetch_ok && k < n_batch) ? __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0f;
                pp += n_threads;
                k += n_threads;
            }
        }
        if (threadIdx.x == 0 && NP > 0) {                     // ... and the flag of batch q + 1
            int p1;
            unsigned par1, m1;
            slot_of(q + 1, p1, par1, m1);
            flag_next = __hip_atomic_load(flags + kFeatReady + 2 * p1 + par1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (staged) {
            // Apply the staged steps.  Within one interaction the table rows do not read each other, so the reference's
            // sequential update of the tables over the batch (rankfm/_rankfm.pyx:283-286, 313-326) is, for each table ROW, a walk
            // over the interactions that touch it -- all rows at once, one row group per row with the row in registers, plain
            // read and write.  The interactions that touch the row are found by the group's lanes together (one ballot per G
            // staged steps); the row of v_if for tag q also carries w_if[q] (lane 0).
            for (int r = gid; r < a.n_uf + a.n_if; r += gpb) {
                const bool uf = r < a.n_uf;
                if (uf ? !a.has_uf : !a.has_if) continue;
                lds_float *row = lds + (size_t)r * F;                                  // v_uf rows, then v_if rows
                const int xoff = 2 * FS + 1 + r;                                        // the step's coefficient of this table row
                const lds_float *vec = stage + (uf ? FS : 0) + sub * KPL;               // updated v_i - v_j | updated v_u, this lane's dwords
                float tr[KPL];
#pragma unroll
                for (int k = 0; k < KPL; ++k) tr[k] = (sub + G * k < F) ? row[sub + G * k] : 0.0f;
                float wq = uf ? 0.0f : lds[n_uf_f + n_if_f + (r - a.n_uf)];
                int last = -1;                                                          // last staged step applied to w_if[q]
                // one touching step: tr <- tr + eta (c v - reg_b tr), and the shrink of w_if over the untouched steps before it
                auto one = [&](int s2, float c, const float (&v)[KPL], float rho_gap) {
#pragma unroll
                    for (int k = 0; k < KPL; ++k) tr[k] += eta_f * (c * v[k] - reg_b * tr[k]);
                    if (!uf) {
                        wq = wq * rho_gap;
                        wq += eta_f * (c - reg_b * wq);
                        last = s2;
                    }
                };
                for (int c64 = 0; c64 < gpb; c64 += 64) {
                    // the steps that touch this row, found by the group's lanes together (one ballot per G staged steps, their LDS reads
                    // in flight together)
                    unsigned long long act = 0;
                    for (int c0 = c64; c0 < gpb && c0 < c64 + 64; c0 += G) {
                        const int mine = c0 + sub;
                        const bool on = mine < gpb && stage[(size_t)mine * NSL + xoff] != 0.0f;
                        unsigned long long bits;
                        if constexpr (G == 64) bits = __ballot(on);
                        else bits = (unsigned long long)group_ballot<G>(on);
                        act |= bits << (c0 - c64);
                    }
                    // two touching steps per round: their coefficients, vectors and shrink powers are read together (one LDS round
                    // trip), then applied one after the other
                    while (act) {
                        const int sA = c64 + __ffsll((long long)act) - 1;
                        act &= act - 1;
                        const bool two = act != 0;
                        const int sB = two ? c64 + __ffsll((long long)act) - 1 : sA;
                        if (two) act &= act - 1;
                        const lds_float *stA = stage + (size_t)sA * NSL, *stB = stage + (size_t)sB * NSL;
                        float vA[KPL], vB[KPL];
                        lds_row_load<KPL>(vec + (size_t)sA * NSL, vA);
                        lds_row_load<KPL>(vec + (size_t)sB * NSL, vB);
                        const float cA = stA[2 * FS] * stA[xoff], cB = stB[2 * FS] * stB[xoff];
                        const float rA = uf ? 1.0f : rho_pow[sA - last - 1], rB = uf ? 1.0f : rho_pow[sB - sA - (two ? 1 : 0)];
                        one(sA, cA, vA, rA);
                        if (two) one(sB, cB, vB, rB);
                    }
                }
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (sub + G * k < F) row[sub + G * k] = tr[k];
                if (!uf && sub == 0) lds[n_uf_f + n_if_f + (r - a.n_uf)] = wq * rho_pow[gpb - 1 - last];
            }
            __syncthreads();
            const unsigned long long tB = wall_clock64();
            // publish the master copy (write-through to memory)
            for (int k = threadIdx.x; k < n_tab; k += blockDim.x)
                __hip_atomic_store(table_ptr(k), lds_tables[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            applied += (unsigned)gpb;
            t_seg[0] += tB - tA;
            t_seg[1] += wall_clock64() - tB;
        }
        const unsigned long long tC = wall_clock64();
        // batch q into the staging area (everybody has finished reading batch q - 1: the barrier above / the first round)
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const unsigned d = (pre_dst[j / 2] >> (16 * (j & 1))) & 0xFFFFu;
            if (prefetch_ok && d != 0xFFFFu) stage[d] = pre[j];
        }
        for (size_t k = threadIdx.x + (prefetch_ok ? (size_t)kPre * blockDim.x : 0); k < batch_floats; k += blockDim.x)
            stage[stage_index(k)] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        staged = true;
        __syncthreads();
        const unsigned long long tD = wall_clock64();
        if (threadIdx.x == 0) flag_store(flags + kFeatConsumed + 2 * p + par, m + 1u);      // (the loads have returned: the slot is free)
        t_seg[2] += tD - tC;
        t_seg[3] += wall_clock64() - tD;
    }
    if (threadIdx.x == 0) {
        a.feat_clock[0] = t_begin;
        a.feat_clock[1] = wall_clock64();
        if (applied) atomicAdd(a.error_flags + 2, applied);      // staged steps applied (rfm_fit_report.table_steps)
        atomicAdd(a.error_flags + 4, (unsigned)(t_wait / 100));
        atomicAdd(a.error_flags + 5, (unsigned)((wall_clock64() - t_begin) / 100));
        for (int k = 0; k < 4; ++k) atomicAdd(a.error_flags + 8 + k, (unsigned)(t_seg[k] / 100));
    }
    // the launch is over: release the producers, wait until they have left, and leave the flags zero for the next launch
    if (threadIdx.x == 0) {
        flag_store(flags + kFeatStop, 1u);
        for (unsigned spin = 0; flag_load(flags + kFeatExited) < (unsigned)NP && spin <= kFeatSpinLimit; ++spin) __builtin_amdgcn_s_sleep(8);
        for (int k = 0; k < kFeatFlagWords; ++k) flag_store(flags + k, 0u);
    }
    return;
}

template <int G, int KPL, bool WARPB>
__device__ __forceinline__ void feat_step_producer(const SgdArgs &a, lds_float *lds, lds_int *s_stop_p) {
    RFM_FEAT_LOCALS
    for (int k = threadIdx.x; k < n_tab; k += blockDim.x) lds_tables[k] = *table_ptr(k);
    lds_float *stage = lds + n_tab;
    __syncthreads();
    // ---- a step producer ----------------------------------------------------------------------------------------------------
    typedef RowStep<G, KPL, false, true, true, true, true, false, WARPB, 1> Train;
    Train step(a, sub, lds, lds + n_uf_f, lds + n_uf_f + n_if_f);
    const int p = (int)blockIdx.x - 1;
    double ll_unused = 0.0;
    unsigned draws_unused = 0;
    const unsigned long long t_begin = wall_clock64();
    unsigned long long t_wait = 0;
    bool paced = true;                                   // (thread 0) still pacing: see below
    constexpr unsigned kPaceGiveUp = 4096;               // polls (~1 us each) of an unmoving ticket counter before pacing is given up
    for (unsigned n = 0;; ++n) {
        const unsigned par = 0u, m = n;                  // (one slot per producer: see the trainer)
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            int stop = 0;
            for (unsigned spin = 0;; ++spin) {           // the slot must have been emptied m times
                if (flag_load(flags + kFeatStop)) { stop = 1; break; }
                if (flag_load(flags + kFeatConsumed + 2 * p + par) >= m) break;
                if (spin > kFeatSpinLimit) { atomicOr(a.error_flags, 8u); stop = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            // Pacing (round 6, SgdArgs::table_pace): the trainer's quota is SPREAD over the first `table_pace` of the launch's segments instead
            // of being worked off as fast as the producers can stage it.  Batch q of the launch (this producer's n-th) is produced once the
            // row loops have been handed q / (batches of the quota) of that share -- the launch's ticket counter, read through an atomic: it
            // is only ever written by memory-side atomics.  The default quota is done at ~0.64 of a launch by itself and barely waits here; a
            // sparser one (tune `table_every`) used to be done within the first third and left the rows of two thirds of every launch
            // without a table step (measured at config 2's shape with tags: twice the spacing -1.4 points of hit_rate@10 bunched, -0.9 when
            // the same steps took 80 % of the launch: profiles/r06_notes.md).  The staged steps stay as fresh as before: a batch is scored
            // when its turn has come, and the trainer applies it as soon as it is there.
            // (The row loops are another KERNEL: if they are not running beside this one -- a profiler that serialises kernels, a device
            //  without room -- the counter never moves; after kPaceGiveUp polls without seeing it move the producer stops pacing for the
            //  rest of the launch: slower tables, still correct.  The host sets table_pace only for row loops that take tickets.)
            if (!stop && paced && a.table_pace > 0.0f && a.tickets != nullptr && a.pos_end > a.pos_begin) {
                const unsigned quota_b = (unsigned)(((a.table_quota > (int64_t)gpb ? a.table_quota : (int64_t)gpb) + gpb - 1) / gpb);
                const double need = (double)(n * (unsigned)NP + (unsigned)p) / (double)quota_b * (double)a.table_pace * (double)(a.pos_end - a.pos_begin);
                unsigned last = 0xFFFFFFFFu, idle = 0;
                for (;;) {
                    const unsigned handed = __hip_atomic_fetch_add(a.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((double)handed >= need) break;
                    if (flag_load(flags + kFeatStop)) { stop = 1; break; }
                    idle = handed == last ? idle + 1 : 0;
                    last = handed;
                    if (idle > kPaceGiveUp) { paced = false; break; }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            t_wait += wall_clock64() - t0;
            *s_stop_p = stop;
        }
        __syncthreads();
        if (*s_stop_p) break;
        // this batch is scored on the tables as published now
        for (int k = threadIdx.x; k < n_tab; k += blockDim.x)
            lds_tables[k] = __hip_atomic_load(table_ptr(k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        // a uniformly random row: a random segment (accepted with probability length / 32) and a random row of it
        uint32_t h = rfm_mix32(a.epoch_key ^ rfm_mix32(((n * (unsigned)NP + (unsigned)p) * (unsigned)gpb + (unsigned)gid) * 0x9E3779B9U + 0x3C6EF372U + a.launch_index));
        int4 d;
        for (;;) {
            d = a.seg_desc[rfm_draw_to_item(h, (uint32_t)a.n_segments)];
            h = rfm_mix32(h + 0x632BE5ABU);
            if ((int)rfm_draw_to_item(h, (uint32_t)kSegmentRows) < d.z) break;
            h = rfm_mix32(h + 0x7F4A7C15U);
        }
        h = rfm_mix32(h ^ 0x85EBCA6BU);
        const int32_t u = d.x, pos = d.y + (int32_t)rfm_draw_to_item(h, (uint32_t)d.z);
        const int32_t i = a.csr_items[pos];
        const float sw = a.sw_csr[pos];
        const int64_t lo = a.csr_off[u], hi = a.csr_off[u + 1];
        float vu[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) vu[k] = (sub + G * k < F) ? load_f32<true>(a.v_u + (size_t)u * F + sub + G * k) : 0.0f;
        step.stage = stage + (size_t)gid * n_slot;
        step(rfm_mix32(h ^ 0xC2B2AE35U), u, i, sw, lo, hi, vu, ll_unused, draws_unused);
        __syncthreads();
        g_float *dst = (g_float *)a.feat_ring + (size_t)(2 * p + par) * batch_floats;
        for (size_t k = threadIdx.x; k < batch_floats; k += blockDim.x)
            __hip_atomic_store(dst + k, stage[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // this wavefront's stores have been performed ...
        __syncthreads();                                     // ... and everybody's, before the slot is announced
        if (threadIdx.x == 0) flag_store(flags + kFeatReady + 2 * p + par, m + 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(a.error_flags + 6, (unsigned)(t_wait / 100));
        atomicAdd(a.error_flags + 7, (unsigned)((wall_clock64() - t_begin) / 100));
        __hip_atomic_fetch_add(flags + kFeatExited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
}

// (inlined into the kernel: as a separate function its pointers would lose their global address space -- the struct is passed by
// value -- and every access of the generic step would become a FLAT instruction)
template <int G, int KPL, bool FRESH, bool WARPB>
__device__ __forceinline__ void feat_generic_rows(const SgdArgs &a, lds_float *lds) {
    RFM_FEAT_LOCALS
    const int64_t group = a.single_group ? (int64_t)threadIdx.x / G : (int64_t)blockIdx.x * gpb + threadIdx.x / G;
    int64_t n_groups = a.single_group ? gpb : (int64_t)n_regular * gpb;
    if (a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    int64_t sp = a.pos_begin + (a.single_group ? 0 : group);
    const int64_t stride = a.single_group ? 1 : n_groups;
    bool active = sp < a.pos_end && (a.single_group ? group == 0 : group < n_groups);
    bool have = false;
    int32_t u = 0, begin = 0, len = 0, t = 0, len_bits = 0;
    uint32_t seg_key = 0;
    int64_t lo = 0, hi = 0;
    float vu[KPL], vu0[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) vu[k] = vu0[k] = 0.0f;
    // ---- generic row loop (WARP, wide feature vectors, other row-group shapes; one group alone: both rows and tables) -----------
    if (threadIdx.x == 0 && blockIdx.x == 0) a.feat_clock[2] = wall_clock64();
    for (int k = threadIdx.x; k < n_tab; k += blockDim.x) lds_tables[k] = *table_ptr(k);
    __syncthreads();
    typedef RowStep<G, KPL, false, true, true, FRESH, true, false, WARPB, 0> Reg;
    typedef RowStep<G, KPL, false, true, true, FRESH, true, false, WARPB, 2> Both;
    Reg step(a, sub, lds, lds + n_uf_f, lds + n_uf_f + n_if_f);
    Both both(a, sub, lds, lds + n_uf_f, lds + n_uf_f + n_if_f);
    const bool train_here = a.single_group && !a.feat_frozen;          // one group alone trains the tables in its LDS
    while (__any(active)) {
        if (trains) {
            const int per = (n_tab + n_waves - 1) / n_waves, e0 = wave * per, e1 = e0 + per < n_tab ? e0 + per : n_tab;
            for (int k = e0 + lane; k < e1; k += 64)
                lds_tables[k] = __hip_atomic_load(table_ptr(k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (active && !have) {
            const uint32_t seg = rfm_perm((uint32_t)sp, (uint32_t)a.n_segments, a.seg_bits, a.epoch_key ^ 0x5bd1e995u);
            const int4 d = a.seg_desc[seg];
            u = d.x; begin = d.y; len = d.z;
            lo = a.csr_off[u]; hi = a.csr_off[u + 1];
            len_bits = (int32_t)rfm_perm_bits((uint32_t)len);
            seg_key = rfm_mix32(a.epoch_key ^ (seg * 0x9E3779B9u + 0x7F4A7C15u));
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                vu0[k] = (sub + G * k < F) ? load_f32<FRESH>(a.v_u + (size_t)u * F + sub + G * k) : 0.0f;
                vu[k] = vu0[k];
            }
            t = 0;
            have = true;
        }
        if (active) {
            const int32_t pos = begin + (int32_t)rfm_perm((uint32_t)t, (uint32_t)len, (uint32_t)len_bits, seg_key);
            const int32_t i = a.csr_items[pos];
            const float sw = a.sw_csr[pos];
            if (train_here) both(rfm_row_key(a.epoch_key, (uint32_t)pos), u, i, sw, lo, hi, vu, ll_acc, draw_acc);
            else step(rfm_row_key(a.epoch_key, (uint32_t)pos), u, i, sw, lo, hi, vu, ll_acc, draw_acc);
            if (++t == len) {
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (sub + G * k < F) atomic_add_f32(a.v_u + (size_t)u * F + sub + G * k, vu[k] - vu0[k]);
                have = false;
                sp += stride;
                active = sp < a.pos_end;
            }
        }
    }
    if (train_here) {             // the one group trained the tables in its LDS: store them
        for (int k = threadIdx.x; k < n_tab; k += blockDim.x) *table_ptr(k) = lds_tables[k];
    }
    flush_counters(a, ll_acc, draw_acc);
    if (threadIdx.x == 0) atomicMax(a.feat_clock + 3, wall_clock64());
}

// The table trainer and its step producers: a kernel of their own (1 + n_producers workgroups, launched on a second stream beside the
// row loops -- launch_segments in rfm_sgd_inst.inc).  Rounds 2-3 ran them as roles inside the row-loop kernel: non-inlined functions
// whose register ceiling (128 at 1024 threads) and 1 KB of stack the row loop shared -- it compiled with 36 spilled VGPRs -- and whose
// step count depended on when the row loops finished.  Here they have their own allocation, the row-loop kernels compile alone, and
// the trainer applies a fixed quota of steps (SgdArgs::table_quota).
template <int G, int KPL, bool WARPB>
__global__ void __launch_bounds__(1024) feat_tables_kernel(const SgdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_dynamic[];
    lds_float *lds = (lds_float *)lds_dynamic;
    __shared__ int s_stop;
    if (blockIdx.x == 0) feat_table_trainer<G, KPL>(a, lds, (lds_int *)&s_stop);
    else feat_step_producer<G, KPL, WARPB>(a, lds, (lds_int *)&s_stop);
}

// the generic row loop (WARP with features, wide feature vectors, 4- / 64-lane row groups; one group alone: rows AND tables)
template <int G, int KPL, bool FRESH, bool WARPB>
__global__ void __launch_bounds__(1024) sgd_features_kernel(const SgdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_dynamic[];
    feat_generic_rows<G, KPL, FRESH, WARPB>(a, (lds_float *)lds_dynamic);
}

// the pipelined row loop (BPR, <= 32 + 32 features, 16-lane row groups): every workgroup walks rows, the tables are a read-only
// lane-major copy in LDS that its wavefronts keep refreshing
// (THREADS: the largest workgroup the instantiation is launched with.  The loop wants ~176 VGPRs: at 1024 threads -- 128 registers --
//  it spills ~50 of them, at 768 -- three wavefronts per SIMD, 168 registers -- two.)
template <int G, int KPL, bool FRESH, int THREADS>
__global__ void __launch_bounds__(THREADS) sgd_features_fast_kernel(const SgdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_dynamic[];
    lds_float *lds = (lds_float *)lds_dynamic;
    RFM_FEAT_LOCALS
    const int first_regular = 0;
    const int64_t group = a.single_group ? (int64_t)threadIdx.x / G : (int64_t)blockIdx.x * gpb + threadIdx.x / G;
    int64_t n_groups = a.single_group ? gpb : (int64_t)n_regular * gpb;
    if (a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    int64_t sp = a.pos_begin + (a.single_group ? 0 : group);
    const int64_t stride = a.single_group ? 1 : n_groups;
    bool active = sp < a.pos_end && (a.single_group ? group == 0 : group < n_groups);
    bool have = false;
    int32_t u = 0, begin = 0, len = 0, t = 0, len_bits = 0;
    uint32_t seg_key = 0;
    int64_t lo = 0, hi = 0;
    float vu[KPL], vu0[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) vu[k] = vu0[k] = 0.0f;
    if (threadIdx.x == 0 && blockIdx.x == 0) a.feat_clock[2] = wall_clock64();
    stamp_clock(a, 0);

    static_assert(G == 16, "the pipelined feature row loop is written for 16-lane row groups");
    {
      {
        // ---- the pipelined row loop (BPR, <= 32 + 32 features, 16-lane row groups) -------------------------------------------
        constexpr int FS = G * KPL;                                     // LDS row stride (rows are padded to full width)
        const int P8 = (a.n_uf + 7) & ~7, Q8 = (a.n_if + 7) & ~7;     // tables padded with zero rows to a multiple of 8
        lds_float *t_uf = lds, *t_if = lds + (size_t)P8 * FS, *t_wif = lds + (size_t)(P8 + Q8) * FS;
        const int n_fast = (P8 + Q8) * FS + a.n_if;
        // LDS element e of the lane-major copy <- table element (global), or zero padding
        auto table_elem = [&](int e) {
            float v = 0.0f;
            if (e < (P8 + Q8) * FS) {
                const int r = e / FS, w = e % FS, f = (w % KPL) * G + w / KPL;     // lane w / KPL, its dword w % KPL
                if (f < F) {
                    if (r < P8) { if (r < a.n_uf) v = __hip_atomic_load(a.v_uf + (size_t)r * F + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
                    else if (r - P8 < a.n_if) v = __hip_atomic_load(a.v_if + (size_t)(r - P8) * F + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            } else v = __hip_atomic_load(a.w_if + (e - (P8 + Q8) * FS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return v;
        };
        for (int e = threadIdx.x; e < n_fast; e += blockDim.x) lds_tables[e] = table_elem(e);
        // hot positive items (SgdArgs::hot_item): pending updates of this workgroup in 32-bit fixed point, behind the tables
        const int hot_off = (n_fast + 3) & ~3, n_hot = a.n_hot;
        lds_int *hot_acc = (lds_int *)(lds + hot_off), *hot_accw = hot_acc + n_hot * F;
        for (int k = threadIdx.x; k < n_hot * (F + 1); k += blockDim.x) hot_acc[k] = 0;
        float hot_scale = 16777216.0f, hot_unit = 1.0f / 16777216.0f;
        if (n_hot > 0) {
            const float range = fmaxf(1.0f, __uint_as_float(*a.sw_max_bits) * a.eta * 10.0f);      // (see RowStep::kHotScale)
            hot_scale = 16777216.0f / range;
            hot_unit = range / 16777216.0f;
        }
        __syncthreads();
        typedef RowStep<G, KPL, false, true, true, FRESH, true, false, false, 0> Reg;
        Reg step(a, sub, lds, lds, lds);                               // (draws, membership test, user damping; its tables are unused)
        const lds_float *uf_lane = t_uf + sub * KPL, *if_lane = t_if + sub * KPL;
        const int lane_base = lane - sub;
        const float multiplier = a.multiplier[1];                       // :269 with sampled == 1
        const float eta = a.eta, reg_a = a.reg_a;
        constexpr int SEGR = (kSegmentRows + G - 1) / G;
        int32_t seg_item[SEGR], seg_pos[SEGR];
        float seg_sw[SEGR];
        // A = x_uf[u] . v_uf (:297-300), the user's projection into factor space: ONCE PER SEGMENT (round 6).  The user is the segment's, and
        // the tables this workgroup reads are a copy that trails the trainer's by a refresh cycle anyway; projecting the user's tags again
        // for every row was half of the row loop's 4096 LDS-operand multiply-adds (the other half, the item pair's difference, is the
        // row's own).  What a row sees of v_uf is then up to a segment (<= 32 rows) old instead of a refresh cycle; with the tables frozen
        // (the one-group parity mode of this loop) nothing changes at all.
        float A[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) A[k] = 0.0f;
        // the positive item's row, bias + step scale (one padded line), tags: fetched one row ahead
        struct Pos { float v[KPL]; float w, scale, x0, x1; } cur, nxt;
        auto fetch_pos = [&](int32_t it, Pos &p) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) p.v[k] = (sub + G * k < F) ? load_f32<FRESH>(a.v_i + (size_t)it * F + sub + G * k) : 0.0f;
            if (a.scale_in_pad) {
                const float x = load_f32<FRESH>(a.w_i + (size_t)it * a.w_stride + (sub & 1));
                p.w = __shfl(x, lane_base);
                p.scale = __shfl(x, lane_base + 1);
            } else {
                p.w = load_f32<FRESH>(a.w_i + (size_t)it * a.w_stride);
                p.scale = a.pos_scale ? a.pos_scale[it] : 1.0f;
            }
            p.x0 = p.x1 = 0.0f;
            if (a.has_if) {
                const float *x = a.x_if + (size_t)it * a.n_if;
                if (sub < a.n_if) p.x0 = x[sub];
                if (sub + G < a.n_if) p.x1 = x[sub + G];
            }
        };
        auto pick = [&](const int32_t (&r)[SEGR], int tt) {
            int32_t x = r[0];
#pragma unroll
            for (int k = 1; k < SEGR; ++k) x = ((unsigned)tt / G == (unsigned)k) ? r[k] : x;
            return x;
        };
        auto pickf = [&](const float (&r)[SEGR], int tt) {
            float x = r[0];
#pragma unroll
            for (int k = 1; k < SEGR; ++k) x = ((unsigned)tt / G == (unsigned)k) ? r[k] : x;
            return x;
        };
        // dynamic segment order (SegmentTickets)
        const bool dynamic = a.tickets != nullptr && !a.single_group;
        SegmentTickets tickets;
        __shared__ int s_ticket_q[kTicketLdsWords];
        tickets.q = (lds_int *)s_ticket_q;
        if (dynamic) {
            if (threadIdx.x == 0) tickets.init_block(a);
            __syncthreads();
            active = group < n_groups && a.pos_begin < a.pos_end;
            int64_t first = -1;
            if (active && sub == 0) first = tickets.take(a);
            sp = __shfl(first, lane_base);
            active = active && sp >= 0;
        }
        // (sweeping turns: wavefront w sweeps in iterations w x every, w x every + wavefronts x every, ...: SgdArgs::hot_sweep_every)
        const int sweep_mod = n_waves * (a.hot_sweep_every > 0 ? a.hot_sweep_every : 1), sweep_at = wave * (a.hot_sweep_every > 0 ? a.hot_sweep_every : 1);
        for (int iter = 0; __any(active); ++iter) {
            // Every wavefront keeps a slice of the workgroup's copy fresh, a part of it per row: the loads are issued here and land in
            // LDS at the END of the row, so that their latency (system-scope loads go to memory) is the row's, not an extra round
            // trip.  (Readers may see a row half old, half new: both are tables the trainer published.)
            constexpr int RF = 2;
            float rf_val[RF];
            int rf_e[RF];
            if (trains) {
                const int per = (n_fast + n_waves - 1) / n_waves, e0 = wave * per, e1 = e0 + per < n_fast ? e0 + per : n_fast;
                const int parts = (per + 64 * RF - 1) / (64 * RF);
#pragma unroll
                for (int k = 0; k < RF; ++k) {
                    rf_e[k] = e0 + lane + 64 * ((iter % parts) * RF + k);
                    rf_val[k] = rf_e[k] < e1 ? table_elem(rf_e[k]) : 0.0f;
                    if (rf_e[k] >= e1) rf_e[k] = -1;
                }
            }
            // bin sweeping duty (SgdArgs::hot_bins_v).  The lines are owned by the ROW-LOOP workgroups only: the trainer and the producers
            // never come here, and a line nobody sweeps -- the first lines are the hottest items' -- would stay unpublished all launch
            if (n_hot > 0 && !a.hot_direct && iter % sweep_mod == sweep_at) {
                const SgdArgs c = cold_args();                           // (the rarely executed parts read their arguments afresh: cold_args)
                for (int line = (int)blockIdx.x - first_regular; line < hot_lines(c); line += n_regular) hot_sweep_line(c, line);
            }
            if (active && !have) {
                const SgdArgs c = cold_args();
                const uint32_t seg = rfm_perm((uint32_t)sp, (uint32_t)c.n_segments, c.seg_bits, c.epoch_key ^ 0x5bd1e995u);
                const int4 d = c.seg_desc[seg];
                u = d.x; begin = d.y; len = d.z;
                lo = c.csr_off[u]; hi = c.csr_off[u + 1];
                len_bits = (int32_t)rfm_perm_bits((uint32_t)len);
                seg_key = rfm_mix32(c.epoch_key ^ (seg * 0x9E3779B9u + 0x7F4A7C15u));
#pragma unroll
                for (int k = 0; k < KPL; ++k) {
                    vu0[k] = (sub + G * k < F) ? load_f32<FRESH>(c.v_u + (size_t)u * F + sub + G * k) : 0.0f;
                    vu[k] = vu0[k];
                }
                t = 0;
                have = true;
                step.load_ulist(lo, hi);
#pragma unroll
                for (int k = 0; k < KPL; ++k) A[k] = 0.0f;
                if (c.has_uf) {
                    float xu0 = 0.0f, xu1 = 0.0f;
                    const float *x = c.x_uf + (size_t)u * c.n_uf;
                    if (sub < c.n_uf) xu0 = x[sub];
                    if (sub + G < c.n_uf) xu1 = x[sub + G];
                    project_dense<KPL>(xu0, xu1, c.n_uf, uf_lane, A);
                }
                // the segment's rows in visiting order, held across the lanes (row t in lane t % G, register t / G)
#pragma unroll
                for (int k = 0; k < SEGR; ++k) {
                    const int tt = sub + G * k;
                    seg_pos[k] = tt < len ? begin + (int32_t)rfm_perm((uint32_t)tt, (uint32_t)len, (uint32_t)len_bits, seg_key) : begin;
                    seg_item[k] = c.csr_items[seg_pos[k]];
                    seg_sw[k] = c.sw_csr[seg_pos[k]];
                }
                fetch_pos(__shfl(pick(seg_item, 0), lane_base), nxt);
            }
            if (active) {
                const int src = lane_base + (int)((unsigned)t % G);
                const int32_t pos = __shfl(pick(seg_pos, t), src), i = __shfl(pick(seg_item, t), src);
                const float sw = __shfl(pickf(seg_sw, t), src);
                cur = nxt;
                // (one group alone is a sequential program: a repeated (user, item) row must see the previous row's update)
                if (a.single_group) fetch_pos(i, cur);
                else if (t + 1 < len) fetch_pos(__shfl(pick(seg_item, t + 1), lane_base + (int)((unsigned)(t + 1) % G)), nxt);   // overlaps this row
                const uint32_t row_key = rfm_row_key(a.epoch_key, (uint32_t)pos);
                // a hot positive item: the workgroup's own pending updates of its row are part of the view (RowStep, HOT)
                int slot = -1;
                if (n_hot > 0 && cur.scale >= 2.0f) {
                    slot = (int)(cur.scale * 0.5f) - 1;
                    cur.scale -= 2.0f * (float)(slot + 1);
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (sub + G * k < F) cur.v[k] += (float)hot_acc[slot * F + sub + G * k] * hot_unit;
                    cur.w += (float)hot_accw[slot] * hot_unit;
                }
                // the negative (:250-253) and its gathers
                uint32_t attempt = 0;
                const int32_t j = step.next_negative(lo, hi, row_key, attempt);
                float vj[KPL], wj, xj0 = 0.0f, xj1 = 0.0f;
#pragma unroll
                for (int k = 0; k < KPL; ++k) vj[k] = (sub + G * k < F) ? load_f32<FRESH>(a.v_i + (size_t)j * F + sub + G * k) : 0.0f;
                // (bias and the item's step scale -- the damping scales an item's step on either side of the pair, RowStep -- in one request)
                float neg_scale_j = 1.0f;
                if (a.scale_in_pad) {
                    const float x = load_f32<FRESH>(a.w_i + (size_t)j * a.w_stride + (sub & 1));
                    wj = __shfl(x, lane_base);
                    neg_scale_j = __shfl(x, lane_base + 1);
                } else {
                    wj = load_f32<FRESH>(a.w_i + (size_t)j * a.w_stride);
                    if (a.pos_scale) neg_scale_j = a.pos_scale[j];
                }
                if (neg_scale_j >= 2.0f) neg_scale_j -= 2.0f * floorf(neg_scale_j * 0.5f);
                if (a.has_if) {
                    const float *x = a.x_if + (size_t)j * a.n_if;
                    if (sub < a.n_if) xj0 = x[sub];
                    if (sub + G < a.n_if) xj1 = x[sub + G];
                }
                float Bd[KPL];
#pragma unroll
                for (int k = 0; k < KPL; ++k) Bd[k] = 0.0f;
                // pairwise utility (:239, :256-257 regrouped: both the utility and the gradients need the item-feature terms only as
                // differences):  pu = (w_i - w_j) + (x_i - x_j).w_if + <v_u + A, v_i - v_j> + <(x_i - x_j).v_if, v_u>
                float part = 0.0f;
                const float dx0 = cur.x0 - xj0, dx1 = cur.x1 - xj1;
                if (a.has_if) {
                    project_dense<KPL>(dx0, dx1, a.n_if, if_lane, Bd);
                    if (sub < a.n_if) part = dx0 * t_wif[sub];
                    if (sub + G < a.n_if) part += dx1 * t_wif[sub + G];
                }
#pragma unroll
                for (int k = 0; k < KPL; ++k) part += (vu[k] + A[k]) * (cur.v[k] - vj[k]) + Bd[k] * vu[k];
                const float pu = (cur.w - wj) + group_sum<G>(part);
                float log_sig, d_outer;
                sigmoid_terms(pu, log_sig, d_outer);                              // :270, :276
                if (sub == 0) { ll_acc += (double)log_sig; draw_acc += 1u; }
                const float g = sw * multiplier;
                const float eta_u = eta * step.user_scale, eta_i = eta * cur.scale, eta_j = a.damp_positive_only ? eta : eta * neg_scale_j;
                float *pi = a.v_i + (size_t)i * F + sub, *pj = a.v_i + (size_t)j * F + sub;
#pragma unroll
                for (int k = 0; k < KPL; ++k) {
                    const float g_u = (cur.v[k] - vj[k]) + Bd[k];                                     // :292, :303-305
                    const float g_i = vu[k] + A[k];                                                   // :293-294, :297-300
                    const float d_u = eta_u * (g * (d_outer * g_u) - reg_a * vu[k]);                  // :308
                    const float d_i = eta_i * (g * (d_outer * g_i) - reg_a * cur.v[k]);               // :309
                    const float d_j = eta_j * (g * (d_outer * -g_i) - reg_a * vj[k]);                 // :310
                    vu[k] += d_u;
                    if (sub + G * k < F) {
                        if (slot >= 0) __hip_atomic_fetch_add(hot_acc + slot * F + sub + G * k, __float2int_rn(d_i * hot_scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else atomic_add_f32(pi + G * k, d_i);
                        atomic_add_f32(pj + G * k, d_j);
                    }
                }
                if (sub == 0) {
                    const float dwi = eta_i * (g * (d_outer * 1.0f) - reg_a * cur.w);                                  // :279
                    if (slot >= 0) __hip_atomic_fetch_add(hot_accw + slot, __float2int_rn(dwi * hot_scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else atomic_add_f32(a.w_i + (size_t)i * a.w_stride, dwi);
                    atomic_add_f32(a.w_i + (size_t)j * a.w_stride, eta_j * (g * (d_outer * -1.0f) - reg_a * wj));      // :280
                }
                // every hot_period-th toucher of a slot publishes what the workgroup has accumulated for it (a keyed coin, RowStep)
                if (slot >= 0 && __umulhi(rfm_mix32(row_key ^ 0x7A5C3B1DU), (uint32_t)a.hot_period[slot]) == 0u) {
                    const SgdArgs c = cold_args();
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        if (sub + G * k >= F) continue;
                        const float d = (float)__hip_atomic_exchange(hot_acc + slot * F + sub + G * k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * hot_unit;
                        if (d != 0.0f)
                            atomic_add_f32(c.hot_direct ? a.v_i + (size_t)i * F + sub + G * k
                                                        : c.hot_bins_v + hot_bin_v(c, blockIdx.x % kHotBins, slot, sub + G * k), d);
                    }
                    if (sub == 0) {
                        const float d = (float)__hip_atomic_exchange(hot_accw + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * hot_unit;
                        if (d != 0.0f) atomic_add_f32(c.hot_direct ? a.w_i + (size_t)i * a.w_stride : c.hot_bins_w + hot_bin_w(c, blockIdx.x % kHotBins, slot), d);
                    }
                }
                if (a.single_group) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");      // the next row reads what this one wrote
                if (++t == len) {
                    const SgdArgs c = cold_args();
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (sub + G * k < F) atomic_add_f32(c.v_u + (size_t)u * F + sub + G * k, vu[k] - vu0[k]);
                    if (c.single_group) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                    have = false;
                    if (dynamic) {
                        int64_t nx = -1;
                        if (sub == 0) nx = tickets.take(c);
                        sp = __shfl(nx, lane_base);
                        active = sp >= 0;
                    } else {
                        sp += stride;
                        active = sp < c.pos_end;
                    }
                }
            }
            if (trains) {
#pragma unroll
                for (int k = 0; k < RF; ++k)
                    if (rf_e[k] >= 0) lds_tables[rf_e[k]] = rf_val[k];
            }
        }
        if (n_hot > 0) {          // publish whatever is still pending
            __syncthreads();
            for (int k = threadIdx.x; k < n_hot * F; k += blockDim.x) {
                const float d = (float)hot_acc[k] * hot_unit;
                if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.v_i + (size_t)a.hot_item[k / F] * F + (k % F)
                                                           : a.hot_bins_v + hot_bin_v(a, blockIdx.x % kHotBins, k / F, k % F), d);
            }
            for (int k = threadIdx.x; k < n_hot; k += blockDim.x) {
                const float d = (float)hot_accw[k] * hot_unit;
                if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.w_i + (size_t)a.hot_item[k] * a.w_stride : a.hot_bins_w + hot_bin_w(a, blockIdx.x % kHotBins, k), d);
            }
        }
        flush_counters(a, ll_acc, draw_acc);
        if (threadIdx.x == 0) atomicMax(a.feat_clock + 3, wall_clock64());
        stamp_clock(a, 1);
      }
    }
}

}  // namespace rfm
