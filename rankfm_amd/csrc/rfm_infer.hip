// rfm_infer.hip -- `_predict` and `_recommend` on the device (the callers either side of the training path).
//
//   rfm_predict_*    replaces rankfm/_rankfm.pyx:345-390  (score arbitrary (u,i) pairs, NaN for unknown ids)
//   rfm_recommend_*  replaces rankfm/_rankfm.pyx:393-460  (score all items per user, rank descending, optionally
//                    skip the user's observed items, keep the first n_items)
//
// Both use the pointwise utility of compute_ui_utility (rankfm/_rankfm.pyx:48-89) in the factored form
//   U(u,i) = w_i[i] + x_if[i].w_if + < v_u[u] + x_uf[u].v_uf , v_i[i] > + < x_if[i].v_if , v_u[u] >
// `_predict` evaluates it per pair with 16-lane groups (coalesced row reads); `_recommend` evaluates all items of a chunk of
// users as one f32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32) followed by a block-wide top-n.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "../../include/rankfm_hip.h"

namespace rfm {

constexpr int kGroup = 16;

__device__ __forceinline__ float group16_sum(float x) {
#pragma unroll
    for (int m = kGroup / 2; m > 0; m >>= 1) x += __shfl_xor(x, m);
    return x;
}

// utility of (u, i); all 16 lanes of the group call it with the same (u, i) and get the same value
__device__ __forceinline__ float utility16(const rfm_model_view &m, int u, int i, int sub) {
    const int F = m.n_factors;
    const float *vu = m.v_u + (size_t)u * F, *vi = m.v_i + (size_t)i * F;
    float part = 0.0f;
    for (int f = sub; f < F; f += kGroup) {
        float eu = vu[f], bi = 0.0f;
        if (m.has_user_features) {
            const float *xu = m.x_uf + (size_t)u * m.n_user_features;
            for (int p = 0; p < m.n_user_features; ++p) {
                const float x = xu[p];
                if (x != 0.0f) eu += x * m.v_uf[(size_t)p * F + f];
            }
        }
        if (m.has_item_features) {
            const float *xi = m.x_if + (size_t)i * m.n_item_features;
            for (int q = 0; q < m.n_item_features; ++q) {
                const float x = xi[q];
                if (x != 0.0f) bi += x * m.v_if[(size_t)q * F + f];
            }
        }
        part += eu * vi[f] + bi * vu[f];
    }
    float res = m.w_i[i] + group16_sum(part);
    if (m.has_item_features) {
        const float *xi = m.x_if + (size_t)i * m.n_item_features;
        float s = 0.0f;
        for (int q = sub; q < m.n_item_features; q += kGroup) s += xi[q] * m.w_if[q];
        res += group16_sum(s);
    }
    return res;
}

__global__ void __launch_bounds__(256) predict_kernel(const rfm_model_view m, long long n_pairs,
                                                      const float *__restrict__ pairs, float *__restrict__ scores) {
    const int sub = threadIdx.x & (kGroup - 1);
    const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    const long long n_groups = ((long long)gridDim.x * blockDim.x) / kGroup;
    for (long long r = group; r < n_pairs; r += n_groups) {
        const float uf = pairs[2 * r], itf = pairs[2 * r + 1];
        float s;
        if (isnan(uf) || isnan(itf)) s = __uint_as_float(0x7fc00000u);   // rankfm/_rankfm.pyx:380-381
        else s = utility16(m, (int)uf, (int)itf, sub);
        if (sub == 0) scores[r] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// `_recommend` scoring as ONE f32 GEMM on the matrix cores (the one dense contraction of the package).
//   score(u, i) = bias[i] + < Ueff[u, :], Veff[i, :] >
//   Ueff[u] = [ v_u[u] + x_uf[u].v_uf | v_u[u] ]          Veff[i] = [ v_i[i] | x_if[i].v_if ]          (second halves only
//   bias[i] = w_i[i] + x_if[i].w_if                                                                   with item features)
// which is compute_ui_utility (rankfm/_rankfm.pyx:48-89) regrouped.  K is padded with zeros to a multiple of 32.
// mfma_f32_32x32x2f32 is exact fp32 (an fmaf chain), so scores agree with the scalar kernel to summation order.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) build_veff_kernel(const rfm_model_view m, int kp, float *__restrict__ veff, float *__restrict__ bias) {
    const int sub = threadIdx.x & (kGroup - 1);
    const int group = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup, n_groups = (gridDim.x * blockDim.x) / kGroup;
    const int F = m.n_factors;
    for (int i = group; i < m.n_items; i += n_groups) {
        float *row = veff + (size_t)i * kp;
        for (int f = sub; f < kp; f += kGroup) {
            float v = 0.0f;
            if (f < F) v = m.v_i[(size_t)i * F + f];
            else if (m.has_item_features && f < 2 * F) {
                const float *xi = m.x_if + (size_t)i * m.n_item_features;
                for (int q = 0; q < m.n_item_features; ++q)
                    if (xi[q] != 0.0f) v += xi[q] * m.v_if[(size_t)q * F + (f - F)];
            }
            row[f] = v;
        }
        float s = 0.0f;
        if (m.has_item_features) {
            const float *xi = m.x_if + (size_t)i * m.n_item_features;
            for (int q = sub; q < m.n_item_features; q += kGroup) s += xi[q] * m.w_if[q];
            s = group16_sum(s);
        }
        if (sub == 0) bias[i] = m.w_i[i] + s;
    }
}

__global__ void __launch_bounds__(256) build_ueff_kernel(const rfm_model_view m, const float *__restrict__ users, long long user_begin,
                                                        int n_slots, int kp, float *__restrict__ ueff) {
    const int sub = threadIdx.x & (kGroup - 1);
    const int group = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup, n_groups = (gridDim.x * blockDim.x) / kGroup;
    const int F = m.n_factors;
    for (int slot = group; slot < n_slots; slot += n_groups) {
        const float uf = users[user_begin + slot];
        float *row = ueff + (size_t)slot * kp;
        const bool cold = isnan(uf);
        const int u = cold ? 0 : (int)uf;
        for (int f = sub; f < kp; f += kGroup) {
            float v = 0.0f;
            if (!cold && f < F) {
                v = m.v_u[(size_t)u * F + f];
                if (m.has_user_features) {
                    const float *xu = m.x_uf + (size_t)u * m.n_user_features;
                    for (int p = 0; p < m.n_user_features; ++p)
                        if (xu[p] != 0.0f) v += xu[p] * m.v_uf[(size_t)p * F + f];
                }
            } else if (!cold && m.has_item_features && f < 2 * F) {
                v = m.v_u[(size_t)u * F + (f - F)];
            }
            row[f] = v;
        }
    }
}

// scores[slot, i] = bias[i] + sum_k ueff[slot, k] * veff[i, k].  64 x 64 output tile per workgroup, 4 wavefronts in 2 x 2,
// each wavefront one 32 x 32 accumulator (16 VGPRs) fed by v_mfma_f32_32x32x2_f32; K advances in LDS stages of 32.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kKT = 32, kLd = kKT + 1;       // +1: the 32 lanes of an operand read walk rows, stride 33 dwords is conflict-free

__global__ void __launch_bounds__(256) scores_mfma_kernel(const float *__restrict__ ueff, const float *__restrict__ veff,
                                                         const float *__restrict__ bias, int n_slots, int n_items, int kp,
                                                         float *__restrict__ scores) {
    __shared__ float sA[64 * kLd], sB[64 * kLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int slot0 = blockIdx.y * 64, item0 = blockIdx.x * 64;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < kp; k0 += kKT) {
        // stage 64 x 32 of each operand: 2048 floats per operand, 8 per thread, consecutive threads along k (coalesced 128 B rows)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int e = tid + 256 * r, row = e >> 5, col = e & 31;
            const int s = slot0 + row, it = item0 + row;
            sA[row * kLd + col] = s < n_slots ? ueff[(size_t)s * kp + k0 + col] : 0.0f;
            sB[row * kLd + col] = it < n_items ? veff[(size_t)it * kp + k0 + col] : 0.0f;
        }
        __syncthreads();
        // A operand: lane l holds A[row = l & 31][k = l >> 5]; B operand: lane l holds B[k = l >> 5][col = l & 31]
        const float *pa = sA + (wr * 32 + (lane & 31)) * kLd + (lane >> 5);
        const float *pb = sB + (wc * 32 + (lane & 31)) * kLd + (lane >> 5);
#pragma unroll
        for (int k = 0; k < kKT; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
        __syncthreads();
    }
    // C/D layout of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = item0 + wc * 32 + (lane & 31);
    if (col < n_items) {
        const float b = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = slot0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < n_slots) scores[(size_t)row * n_items + col] = acc[r] + b;
        }
    }
}

// one block per user slot: optionally knock out the user's observed items, then n_rec rounds of block-wide argmax.
// Ranking is descending by utility; equal scores may come out in any order (np.argsort in the reference is
// unstable too, rankfm/_rankfm.pyx:444).
__global__ void __launch_bounds__(256) topn_kernel(const float *__restrict__ users, long long user_begin, int n_items,
                                                   const int64_t *__restrict__ csr_off, const int32_t *__restrict__ csr_items,
                                                   int filter_previous, int n_rec, float *__restrict__ scores,
                                                   float *__restrict__ rec) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const long long slot = blockIdx.x;
    const float uf = users[user_begin + slot];
    float *out = rec + (size_t)(user_begin + slot) * n_rec;
    if (isnan(uf)) {                                                      // rankfm/_rankfm.pyx:435-437
        for (int k = threadIdx.x; k < n_rec; k += blockDim.x) out[k] = __uint_as_float(0x7fc00000u);
        return;
    }
    const int u = (int)uf;
    float *row = scores + (size_t)slot * n_items;
    const float kRemoved = -INFINITY;
    if (filter_previous) {                                                // rankfm/_rankfm.pyx:450-451
        for (int64_t k = csr_off[u] + threadIdx.x; k < csr_off[u + 1]; k += blockDim.x) row[csr_items[k]] = kRemoved;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int r = 0; r < n_rec; ++r) {
        float best = -INFINITY;
        int best_i = -1;
        for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
            const float v = row[i];
            // NaN scores never win; among equals the larger index wins, like a reversed ascending sort
            if (v > best || (v == best && v > kRemoved && i > best_i)) { best = v; best_i = i; }
        }
#pragma unroll
        for (int msk = 32; msk > 0; msk >>= 1) {
            const float ov = __shfl_xor(best, msk);
            const int oi = __shfl_xor(best_i, msk);
            if (oi >= 0 && (ov > best || (ov == best && oi > best_i))) { best = ov; best_i = oi; }
        }
        if (lane == 0) { s_val[wid] = best; s_idx[wid] = best_i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = s_val[0];
            int bi = s_idx[0];
            for (int w = 1; w < 4; ++w)
                if (s_idx[w] >= 0 && (s_val[w] > b || (s_val[w] == b && s_idx[w] > bi))) { b = s_val[w]; bi = s_idx[w]; }
            out[r] = bi >= 0 ? (float)bi : __uint_as_float(0x7fc00000u);
            if (bi >= 0) row[bi] = kRemoved;
        }
        __syncthreads();
    }
}

// Threshold selection for n_rec <= kTopLocal (the usual top-10): exact, two cheap passes over the score row instead of n_rec.
//   pass 1  the row is cut into up to kSegs interleaved segments (segment s = elements s, s + nseg, ...: coalesced); every
//           segment's best element goes to LDS; n_rec rounds of block-wide argmax over those maxima leave T = the n_rec-th
//           best segment maximum.  At least n_rec elements rank at or above T, so the top n_rec all do;
//   pass 2  every element ranking at or above T is appended to an LDS candidate list -- at most (n_rec - 1) x segment
//           length + 1 of them, typically ~n_rec -- and n_rec rounds of argmax over the list give the ranking.
// Same ordering rules as topn_kernel: descending utility, among equals the larger index first, NaN never wins, observed items
// knocked out.
constexpr int kTopLocal = 16;
constexpr int kSegs = 4096, kCands = 4096;

__device__ __forceinline__ bool ranks_before(float v, int i, float w, int j) {      // (v, i) ranks before (w, j)
    return v > w || (v == w && i > j);
}

// block-wide argmax of (val, idx) pairs held one per thread; result broadcast to every thread
__device__ __forceinline__ void block_best(float &v, int &i, float *w_val, int *w_idx) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int msk = 32; msk > 0; msk >>= 1) {
        const float ov = __shfl_xor(v, msk);
        const int oi = __shfl_xor(i, msk);
        if (oi >= 0 && (i < 0 || ranks_before(ov, oi, v, i))) { v = ov; i = oi; }
    }
    __syncthreads();
    if (lane == 0) { w_val[wid] = v; w_idx[wid] = i; }
    __syncthreads();
    v = w_val[0]; i = w_idx[0];
    for (int w = 1; w < 4; ++w)
        if (w_idx[w] >= 0 && (i < 0 || ranks_before(w_val[w], w_idx[w], v, i))) { v = w_val[w]; i = w_idx[w]; }
}

__global__ void __launch_bounds__(256) topn_select_kernel(const float *__restrict__ users, long long user_begin, int n_items,
                                                          const int64_t *__restrict__ csr_off, const int32_t *__restrict__ csr_items,
                                                          int filter_previous, int n_rec, int skip_index, float *__restrict__ scores,
                                                          float *__restrict__ rec) {
    __shared__ float s_val[kSegs];          // pass 1: segment maxima; pass 2: candidates
    __shared__ int s_idx[kSegs];
    __shared__ float w_val[4];
    __shared__ int w_idx[4];
    __shared__ int n_cand;
    const long long slot = blockIdx.x;
    const float uf = users ? users[user_begin + slot] : 0.0f;
    float *out = rec + (size_t)(user_begin + slot) * n_rec;
    if (isnan(uf)) {                                                      // rankfm/_rankfm.pyx:435-437
        for (int k = threadIdx.x; k < n_rec; k += blockDim.x) out[k] = __uint_as_float(0x7fc00000u);
        return;
    }
    float *row = scores + (size_t)slot * n_items;
    if (filter_previous) {                                                // rankfm/_rankfm.pyx:450-451
        const int u = (int)uf;
        for (int64_t k = csr_off[u] + threadIdx.x; k < csr_off[u + 1]; k += blockDim.x) row[csr_items[k]] = -INFINITY;
    }
    if (threadIdx.x == 0) n_cand = 0;
    __syncthreads();
    const int seglen = (n_items + kSegs - 1) / kSegs, nseg = (n_items + seglen - 1) / seglen;
    // ---- pass 1: segment maxima
    for (int sg = threadIdx.x; sg < nseg; sg += blockDim.x) {
        float bv = -INFINITY;
        int bi = -1;
        for (int j = 0; j < seglen; ++j) {
            const int i = sg + j * nseg;
            if (i >= n_items) break;
            const float v = row[i];
            if (i != skip_index && v > -INFINITY && (bi < 0 || ranks_before(v, i, bv, bi))) { bv = v; bi = i; }
        }
        s_val[sg] = bv; s_idx[sg] = bi;
    }
    __syncthreads();
    // the n_rec-th best segment maximum (selected maxima are struck out of the LDS copy as they are found)
    float tv = -INFINITY;
    int ti = -1;
    bool take_all = false;      // fewer than n_rec segments hold a rankable element: every rankable element is a candidate
    for (int r = 0; r < n_rec; ++r) {
        float bv = -INFINITY;
        int bi = -1, bs = -1;
        for (int sg = threadIdx.x; sg < nseg; sg += blockDim.x)
            if (s_idx[sg] >= 0 && (bi < 0 || ranks_before(s_val[sg], s_idx[sg], bv, bi))) { bv = s_val[sg]; bi = s_idx[sg]; bs = sg; }
        float gv = bv;
        int gi = bi;
        block_best(gv, gi, w_val, w_idx);
        // fewer than n_rec segments with a rankable element (a user who has seen nearly everything, NaN scores): the last maximum
        // found is NOT a lower bound of the top n_rec -- the non-maximal elements of those segments are needed to fill the list.
        // They number at most (n_rec - 1) x segment length, which the candidate list holds.
        if (gi < 0) { take_all = true; break; }
        tv = gv; ti = gi;
        if (bi == gi && bs >= 0) s_idx[bs] = -1;                           // (indexes are unique: exactly one thread strikes)
        __syncthreads();
    }
    __syncthreads();
    // ---- pass 2: everything that ranks at or above the threshold
    if (ti >= 0) {
        for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
            const float v = row[i];
            if (i == skip_index || !(v > -INFINITY)) continue;
            if (take_all || i == ti || ranks_before(v, i, tv, ti)) {
                const int c = atomicAdd(&n_cand, 1);
                if (c < kCands) { s_val[c] = v; s_idx[c] = i; }
            }
        }
    }
    __syncthreads();
    const int nc = n_cand < kCands ? n_cand : kCands;
    for (int r = 0; r < n_rec; ++r) {
        float bv = -INFINITY;
        int bi = -1, bc = -1;
        for (int c = threadIdx.x; c < nc; c += blockDim.x)
            if (s_idx[c] >= 0 && (bi < 0 || ranks_before(s_val[c], s_idx[c], bv, bi))) { bv = s_val[c]; bi = s_idx[c]; bc = c; }
        float gv = bv;
        int gi = bi;
        block_best(gv, gi, w_val, w_idx);
        if (threadIdx.x == 0) out[r] = gi >= 0 ? (float)gi : __uint_as_float(0x7fc00000u);
        if (gi >= 0 && bi == gi && bc >= 0) s_idx[bc] = -1;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// `_recommend` WITHOUT the score matrix (round 5).  The two-pass threshold selection above reads a [users, items]
// matrix twice that the GEMM has just written (1.4 GB for 9,936 users x 35 k items); the only thing its first pass needs of it are
// per-segment maxima, and those can come straight out of the accumulators:
//   seen_mask_kernel        one bit per (user slot, item): the user's observed items (rankfm/_rankfm.pyx:450-451), from the CSR lists
//   scores_blockmax_kernel  the scoring GEMM with the ROLES SWAPPED -- items are the rows of the 32 x 32 accumulator, users its columns --
//                           so that a lane holds sixteen items of ONE user: the best of a block of 32 items is fifteen compares in
//                           registers and one exchange with lane ^ 32 instead of a five-step shuffle reduction per row; it writes the
//                           best (score, item) of every (block of 32 items, user), observed items and NaN excluded, and nothing else
//   select_blocks_kernel    one workgroup per user: the n_rec-th best block maximum is a lower bound of the top n_rec (at least n_rec
//                           items rank at or above it); the blocks whose maximum reaches it -- n_rec of them, a few more under ties --
//                           are scored again (<= a few hundred dot products), and n_rec rounds of argmax give the ranking.
// Same ordering rules as topn_select_kernel (descending utility, among equals the larger index first, NaN never wins, observed items
// out; equal scores may come out in any order in the reference too: np.argsort is unstable, rankfm/_rankfm.pyx:444).  The re-scored
// candidates are fp32 FMA chains like the matrix cores' (v_mfma_f32_32x32x2_f32 is exact fp32), so the ranking is the scores' own.
// ---------------------------------------------------------------------------------------------
constexpr int kBlk = 32;                          // items per block maximum: the rows of one wavefront's accumulator
constexpr int kMaxCandBlocks = kCands / kBlk;     // candidate blocks a user can hold in LDS (128)

__global__ void __launch_bounds__(256) seen_mask_kernel(const float *__restrict__ users, long long user_begin, const int64_t *__restrict__ csr_off,
                                                       const int32_t *__restrict__ csr_items, int n_words, unsigned *__restrict__ mask) {
    const long long slot = blockIdx.x;
    const float uf = users[user_begin + slot];
    if (isnan(uf)) return;
    const int u = (int)uf;
    unsigned *row = mask + (size_t)slot * n_words;
    for (int64_t k = csr_off[u] + threadIdx.x; k < csr_off[u + 1]; k += blockDim.x) {
        const int it = csr_items[k];
        atomicOr(row + (it >> 5), 1u << (it & 31));
    }
}

__global__ void __launch_bounds__(256) scores_blockmax_kernel(const float *__restrict__ ueff, const float *__restrict__ veff,
                                                             const float *__restrict__ bias, const unsigned *__restrict__ mask, int n_slots,
                                                             int n_items, int kp, int n_words, float *__restrict__ bmax_val,
                                                             int *__restrict__ bmax_idx) {
    __shared__ float sA[64 * kLd], sB[64 * kLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                 // item half / user half of the 64 x 64 tile
    const int item0 = blockIdx.x * 64, slot0 = blockIdx.y * 64;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < kp; k0 += kKT) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int e = tid + 256 * r, row = e >> 5, col = e & 31;
            const int it = item0 + row, s = slot0 + row;
            sA[row * kLd + col] = it < n_items ? veff[(size_t)it * kp + k0 + col] : 0.0f;
            sB[row * kLd + col] = s < n_slots ? ueff[(size_t)s * kp + k0 + col] : 0.0f;
        }
        __syncthreads();
        // A operand (items): lane l holds A[row = l & 31][k = l >> 5]; B operand (users): lane l holds B[k = l >> 5][col = l & 31]
        const float *pa = sA + (wr * 32 + (lane & 31)) * kLd + (lane >> 5);
        const float *pb = sB + (wc * 32 + (lane & 31)) * kLd + (lane >> 5);
#pragma unroll
        for (int k = 0; k < kKT; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
        __syncthreads();
    }
    // accumulator register r of lane l: item row (r & 3) + 8 (r >> 2) + 4 (l >> 5), user column l & 31
    const int slot = slot0 + wc * 32 + (lane & 31);
    const int blk = (item0 >> 5) + wr;
    const bool live = slot < n_slots && blk < n_words;
    const unsigned seen = (live && mask) ? mask[(size_t)slot * n_words + blk] : 0u;
    float bv = -INFINITY;
    int bi = -1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int it = item0 + wr * 32 + row;
        if (live && it < n_items && !((seen >> row) & 1u)) {
            const float v = acc[r] + bias[it];
            if (v > -INFINITY && (bi < 0 || ranks_before(v, it, bv, bi))) { bv = v; bi = it; }
        }
    }
    const float ov = __shfl_xor(bv, 32);
    const int oi = __shfl_xor(bi, 32);
    if (oi >= 0 && (bi < 0 || ranks_before(ov, oi, bv, bi))) { bv = ov; bi = oi; }
    if (live && lane < 32) {
        bmax_val[(size_t)blk * n_slots + slot] = bv;
        bmax_idx[(size_t)blk * n_slots + slot] = bi;
    }
}

// The same block maxima with BOTH operands in registers (round 6; padded k <= 128).  scores_blockmax_kernel stages a 64 x 64 tile's
// operands through LDS for every K-step of 32 and synchronises twice per step: with k = 64 a workgroup's whole life is two such steps, the
// matrix cores wait for the loads of each (40 TF/s of the 157 TF/s fp32 peak).  Here a WAVEFRONT owns 32 users for
// good -- its B operand, kp / 2 registers per lane, is loaded once -- and walks blocks of 32 items whose A operand (another kp / 2
// registers) arrives from memory in 16-byte loads one block AHEAD of the one the matrix cores work on.  No LDS, no barrier.  What makes the
// operands register-friendly is the freedom in which k goes to which half of the wavefront: v_mfma_f32_32x32x2_f32 takes, per step, k0 from
// lanes 0 .. 31 and k1 from lanes 32 .. 63 -- any pairing of the k's into steps gives the same sum up to its order.  With k0 = s, k1 =
// kp / 2 + s a lane needs kp / 2 CONTIGUOUS floats of its row of each operand.  (The order of the summation differs from the k-ordered
// chain the candidates are re-scored with in select_blocks_kernel: the maxima only choose the candidate blocks, the ranking is the chain's --
// and tests/test_gpu_api.py::test_matrix_free_and_matrix_recommend_paths_agree holds the two paths to each other.)
// The four wavefronts of a workgroup walk the SAME item blocks for four neighbouring user blocks, so three of their four A loads hit L1.
template <int KH, bool MASK>
__global__ void __launch_bounds__(256) scores_blockmax_reg_kernel(const float *__restrict__ ueff, const float *__restrict__ veff,
                                                                 const float *__restrict__ bias, const unsigned *__restrict__ mask, int n_slots,
                                                                 int n_items, int n_words, float *__restrict__ bmax_val, int *__restrict__ bmax_idx) {
    static_assert(KH % 4 == 0 && KH >= 16 && KH <= 64, "half the padded factor count, in 16-byte loads");
    constexpr int kp = 2 * KH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int slot = (blockIdx.y * 4 + wave) * 32 + l31;
    const bool slot_ok = slot < n_slots;
    const int slot_c = slot_ok ? slot : n_slots - 1;        // (every load below is unconditional on a clamped address: no branch, so the
    //                                                          compiler can wait for one block's operands while the next one's are in flight)
    float b[KH];
    {
        const float4 *pb = reinterpret_cast<const float4 *>(ueff + (size_t)slot_c * kp + half * KH);
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            const float4 v = pb[q];
            b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
    }
    // the item biases ride in the product as one more k: A[row][kp] = bias (lanes 0 .. 31; 0 in the upper half), B[kp][col] = 1
    const float b_one = 1.0f;
    struct Operand { float a[KH]; float ab; unsigned seen; };
    auto load_a = [&](int blk, Operand &o) {
        const int bc = blk < n_words ? blk : n_words - 1;
        const int it = bc * kBlk + l31 < n_items ? bc * kBlk + l31 : n_items - 1;
        const float4 *pa = reinterpret_cast<const float4 *>(veff + (size_t)it * kp + half * KH);
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            const float4 v = pa[q];
            o.a[4 * q] = v.x; o.a[4 * q + 1] = v.y; o.a[4 * q + 2] = v.z; o.a[4 * q + 3] = v.w;
        }
        const float bv = bias[it];
        o.ab = half == 0 ? bv : 0.0f;
        o.seen = 0u;
        if constexpr (MASK) o.seen = mask[(size_t)slot_c * n_words + bc];
    };
    auto block = [&](int blk, const Operand &o) {
        f32x16 acc = {0};
#pragma unroll
        for (int s2 = 0; s2 < KH; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[s2], b[s2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(o.ab, b_one, acc, 0, 0, 0);
        // accumulator register r of lane l: item row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the block, user column l & 31.  Rows ascend with
        // r, so among equal scores the LATER register wins (ranks_before: the larger index first); selects, no branches.
        float bv = -INFINITY;
        int bi = -1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int it = blk * kBlk + row;
            const float v = acc[r];
            const bool take = (it < n_items) & !((o.seen >> row) & 1u) & (v > -INFINITY) & (v >= bv);
            bv = take ? v : bv;
            bi = take ? it : bi;
        }
        const float ov = __shfl_xor(bv, 32);
        const int oi = __shfl_xor(bi, 32);
        const bool other = (oi >= 0) & ((bi < 0) | (ov > bv) | ((ov == bv) & (oi > bi)));
        bv = other ? ov : bv;
        bi = other ? oi : bi;
        // (block-major: the 32 lanes' stores are one 128-byte line.  User-major -- a contiguous row of maxima per user for the selection --
        //  was measured: the product 0.57 -> 0.67 ms for its scattered stores, the selection 0.49 -> 0.45: not taken)
        if (slot_ok && lane < 32) {
            bmax_val[(size_t)blk * n_slots + slot] = bv;
            bmax_idx[(size_t)blk * n_slots + slot] = bi;
        }
    };
    // item blocks blockIdx.x, blockIdx.x + gridDim.x, ...: two per trip, the second one's operand loading while the first is multiplied
    Operand o0, o1;
    int blk = blockIdx.x;
    const int step = gridDim.x;
    load_a(blk, o0);
    for (; blk < n_words; blk += 2 * step) {
        load_a(blk + step, o1);
        block(blk, o0);
        load_a(blk + 2 * step, o0);
        if (blk + step < n_words) block(blk + step, o1);
    }
}

__global__ void __launch_bounds__(256) select_blocks_kernel(const float *__restrict__ users, long long user_begin, int n_slots, int n_items, int kp,
                                                           const float *__restrict__ ueff, const float *__restrict__ veff,
                                                           const float *__restrict__ bias, const unsigned *__restrict__ mask, int n_words,
                                                           const float *__restrict__ bmax_val, const int *__restrict__ bmax_idx, int n_rec,
                                                           float *__restrict__ rec) {
    __shared__ float s_val[kSegs];          // block maxima, then the candidates
    __shared__ int s_idx[kSegs];
    __shared__ int s_blk[kMaxCandBlocks];
    __shared__ float s_u[512];              // the user's effective factor row (kp <= 512)
    __shared__ float w_val[4];
    __shared__ int w_idx[4];
    __shared__ int n_cand, n_blk;
    const int slot = blockIdx.x;
    const float uf = users[user_begin + slot];
    float *out = rec + (size_t)(user_begin + slot) * n_rec;
    if (isnan(uf)) {                                                      // rankfm/_rankfm.pyx:435-437
        for (int k = threadIdx.x; k < n_rec; k += blockDim.x) out[k] = __uint_as_float(0x7fc00000u);
        return;
    }
    for (int b = threadIdx.x; b < n_words; b += blockDim.x) {
        s_val[b] = bmax_val[(size_t)b * n_slots + slot];
        s_idx[b] = bmax_idx[(size_t)b * n_slots + slot];
    }
    for (int k = threadIdx.x; k < kp; k += blockDim.x) s_u[k] = ueff[(size_t)slot * kp + k];
    if (threadIdx.x == 0) { n_cand = 0; n_blk = 0; }
    __syncthreads();
    // the n_rec-th best block maximum (selected maxima are struck out of the LDS copy: index -> -2 - index, so that they can be told from
    // blocks without a rankable item, -1)
    float tv = -INFINITY;
    int ti = -1;
    bool take_all = false;      // fewer than n_rec blocks hold a rankable item: every rankable item of those blocks is a candidate
    for (int r = 0; r < n_rec; ++r) {
        float bv = -INFINITY;
        int bi = -1, bs = -1;
        for (int b = threadIdx.x; b < n_words; b += blockDim.x)
            if (s_idx[b] >= 0 && (bi < 0 || ranks_before(s_val[b], s_idx[b], bv, bi))) { bv = s_val[b]; bi = s_idx[b]; bs = b; }
        float gv = bv;
        int gi = bi;
        block_best(gv, gi, w_val, w_idx);
        if (gi < 0) { take_all = true; break; }
        tv = gv; ti = gi;
        if (bi == gi && bs >= 0) s_idx[bs] = -2 - s_idx[bs];
        __syncthreads();
    }
    __syncthreads();
    // candidate blocks: every block whose maximum ranks at or above the threshold (the struck-out ones, and ties with the last of them)
    for (int b = threadIdx.x; b < n_words; b += blockDim.x) {
        const int raw = s_idx[b];
        const int idx = raw <= -2 ? -2 - raw : raw;
        if (idx < 0) continue;
        if (take_all || raw <= -2 || idx == ti || ranks_before(s_val[b], idx, tv, ti)) {
            const int c = atomicAdd(&n_blk, 1);
            if (c < kMaxCandBlocks) s_blk[c] = b;
        }
    }
    __syncthreads();
    const int nb = n_blk < kMaxCandBlocks ? n_blk : kMaxCandBlocks;
    __syncthreads();
    // score the candidate blocks' items again: bias + the fp32 FMA chain over the padded factor row, observed items and NaN left out
    for (int e = threadIdx.x; e < nb * kBlk; e += blockDim.x) {
        const int b = s_blk[e / kBlk], it = b * kBlk + (e % kBlk);
        if (it >= n_items) continue;
        if (mask && ((mask[(size_t)slot * n_words + b] >> (it & 31)) & 1u)) continue;
        const float4 *vr = reinterpret_cast<const float4 *>(veff + (size_t)it * kp);      // (kp is a multiple of 32: 16-byte loads, the same k order)
        float acc = 0.0f;
        for (int k = 0; k < kp; k += 4) {
            const float4 x = vr[k >> 2];
            acc = fmaf(x.x, s_u[k], acc); acc = fmaf(x.y, s_u[k + 1], acc); acc = fmaf(x.z, s_u[k + 2], acc); acc = fmaf(x.w, s_u[k + 3], acc);
        }
        const float v = acc + bias[it];
        if (!(v > -INFINITY)) continue;
        const int c = atomicAdd(&n_cand, 1);
        s_val[c] = v; s_idx[c] = it;                  // (nb * kBlk <= kCands = kSegs entries)
    }
    __syncthreads();
    const int nc = n_cand;
    for (int r = 0; r < n_rec; ++r) {
        float bv = -INFINITY;
        int bi = -1, bc = -1;
        for (int c = threadIdx.x; c < nc; c += blockDim.x)
            if (s_idx[c] >= 0 && (bi < 0 || ranks_before(s_val[c], s_idx[c], bv, bi))) { bv = s_val[c]; bi = s_idx[c]; bc = c; }
        float gv = bv;
        int gi = bi;
        block_best(gv, gi, w_val, w_idx);
        if (threadIdx.x == 0) out[r] = gi >= 0 ? (float)gi : __uint_as_float(0x7fc00000u);
        if (gi >= 0 && bi == gi && bc >= 0) s_idx[bc] = -1;
        __syncthreads();
    }
}

// select_blocks_kernel with ONE WAVEFRONT per user (round 6): the block maxima sit in registers (VPL per lane), the n_rec-th best of them
// and the final ranking are n_rec rounds of a shuffle arg-max each -- no LDS scan, no barrier (the workgroup form spends its time in 2 x n_rec
// block-wide rounds of two barriers each: 0.49 ms for 9,936 users, as long as the product that feeds it) -- and the candidate blocks are
// scored two at a time, one item per lane.  Same candidate rule, same fmaf chain over k, same ordering as select_blocks_kernel; any number
// of candidate blocks (ties) is handled sixteen blocks at a time against a running list.
template <int VPL>
__global__ void __launch_bounds__(256) select_blocks_wave_kernel(const float *__restrict__ users, long long user_begin, int n_slots, int n_items, int kp,
                                                                const float *__restrict__ ueff, const float *__restrict__ veff,
                                                                const float *__restrict__ bias, const unsigned *__restrict__ mask, int n_words,
                                                                const float *__restrict__ bmax_val, const int *__restrict__ bmax_idx, int n_rec,
                                                                float *__restrict__ rec) {
    constexpr int kChunk = 16;                        // candidate blocks scored per pass: 512 items, eight per lane
    __shared__ float s_u[4][512];                     // the users' effective factor rows (kp <= 512), one per wavefront
    __shared__ int s_blk[4][64 * VPL];                // the candidate blocks of each wavefront's user
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= n_slots) return;                      // (no barrier below: a wavefront may leave)
    const float uf = users[user_begin + slot];
    float *out = rec + (size_t)(user_begin + slot) * n_rec;
    if (isnan(uf)) {                                                      // rankfm/_rankfm.pyx:435-437
        if (lane < n_rec) out[lane] = __uint_as_float(0x7fc00000u);
        return;
    }
    float bv[VPL];
    int bi[VPL];
#pragma unroll
    for (int e = 0; e < VPL; ++e) {
        const int b = e * 64 + lane;
        const bool ok = b < n_words;
        bv[e] = ok ? bmax_val[(size_t)b * n_slots + slot] : -INFINITY;
        bi[e] = ok ? bmax_idx[(size_t)b * n_slots + slot] : -1;
    }
    float *su = s_u[wave];
    for (int k = lane; k < kp; k += 64) su[k] = ueff[(size_t)slot * kp + k];
    auto wave_best = [&](float &v, int &i) {          // arg-max over the wavefront, every lane gets it
#pragma unroll
        for (int msk = 32; msk > 0; msk >>= 1) {
            const float ov = __shfl_xor(v, msk);
            const int oi = __shfl_xor(i, msk);
            const bool t = (oi >= 0) & ((i < 0) | (ov > v) | ((ov == v) & (oi > i)));
            v = t ? ov : v;
            i = t ? oi : i;
        }
    };
    // the n_rec-th best block maximum
    unsigned struck = 0;
    float tv = -INFINITY;
    int ti = -1;
    bool take_all = false;        // fewer than n_rec blocks hold a rankable item: every rankable item of those blocks is a candidate
    for (int r = 0; r < n_rec; ++r) {
        float lv = -INFINITY;
        int li = -1, le = -1;
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const bool t = !((struck >> e) & 1u) & (bi[e] >= 0) & ((li < 0) | (bv[e] > lv) | ((bv[e] == lv) & (bi[e] > li)));
            lv = t ? bv[e] : lv; li = t ? bi[e] : li; le = t ? e : le;
        }
        float gv = lv;
        int gi = li;
        wave_best(gv, gi);
        if (gi < 0) { take_all = true; break; }
        tv = gv; ti = gi;
        if (li == gi && le >= 0) struck |= 1u << le;
    }
    // candidate blocks: every block whose maximum ranks at or above the threshold (the struck-out ones, and ties with the last of them)
    int nb = 0;
    int *sb = s_blk[wave];
#pragma unroll
    for (int e = 0; e < VPL; ++e) {
        const int idx = bi[e];
        const bool c = (idx >= 0) & (take_all | (((struck >> e) & 1u) != 0u) | (idx == ti) | (bv[e] > tv) | ((bv[e] == tv) & (idx > ti)));
        const unsigned long long m = __ballot(c);
        if (c) sb[nb + __popcll(m & ((1ull << lane) - 1ull))] = e * 64 + lane;
        nb += __popcll(m);
    }
    const unsigned *mrow = mask ? mask + (size_t)slot * n_words : nullptr;
    const int half = lane >> 5, l31 = lane & 31;
    int n_run = 0;
    float topv = -INFINITY;       // lane r < n_run: the r-th best so far
    int topi = -1;
    for (int c0 = 0; c0 < nb; c0 += kChunk) {
        float cv[kChunk / 2 + 1];
        int ci[kChunk / 2 + 1];
#pragma unroll
        for (int t = 0; t < kChunk / 2; ++t) {
            cv[t] = -INFINITY; ci[t] = -1;
            if (c0 + 2 * t >= nb) continue;                              // (wavefront-uniform)
            const int bidx = c0 + 2 * t + half;
            const bool valid = bidx < nb;
            const int b = valid ? sb[bidx] : 0;
            const int it = b * kBlk + l31;
            const bool ok = valid && it < n_items && !(mrow && ((mrow[b] >> l31) & 1u));
            const float4 *vr = reinterpret_cast<const float4 *>(veff + (size_t)(ok ? it : 0) * kp);      // (16-byte loads, the same k order as the scalar chain)
            float acc = 0.0f;
            for (int k = 0; k < kp; k += 4) {
                const float4 x = vr[k >> 2];
                acc = fmaf(x.x, su[k], acc); acc = fmaf(x.y, su[k + 1], acc); acc = fmaf(x.z, su[k + 2], acc); acc = fmaf(x.w, su[k + 3], acc);
            }
            const float v = acc + bias[ok ? it : 0];
            const bool keep = ok && v > -INFINITY;
            cv[t] = keep ? v : -INFINITY;
            ci[t] = keep ? it : -1;
        }
        cv[kChunk / 2] = lane < n_run ? topv : -INFINITY;               // the running list joins the pass as one more entry per lane
        ci[kChunk / 2] = lane < n_run ? topi : -1;
        float nv = -INFINITY;
        int ni = -1, got = 0;
        for (int r = 0; r < n_rec; ++r) {
            float lv = -INFINITY;
            int li = -1, le = -1;
#pragma unroll
            for (int t = 0; t <= kChunk / 2; ++t) {
                const bool w = (ci[t] >= 0) & ((li < 0) | (cv[t] > lv) | ((cv[t] == lv) & (ci[t] > li)));
                lv = w ? cv[t] : lv; li = w ? ci[t] : li; le = w ? t : le;
            }
            float gv = lv;
            int gi = li;
            wave_best(gv, gi);
            if (gi < 0) break;
            if (lane == r) { nv = gv; ni = gi; }
            ++got;
            if (li == gi && le >= 0) {
#pragma unroll
                for (int t = 0; t <= kChunk / 2; ++t) ci[t] = t == le ? -1 : ci[t];
            }
        }
        topv = nv; topi = ni; n_run = got;
    }
    if (lane < n_rec) out[lane] = (lane < n_run && topi >= 0) ? (float)topi : __uint_as_float(0x7fc00000u);
}

// similar_items / similar_users (rankfm/rankfm.py:405-454): latent representation rep[r] = v[r] + x[r] . v_f of every row, its
// dot product with the query row's representation -> sims [n_rows]; the ranking is topn_select_kernel with the query skipped.
__global__ void __launch_bounds__(256) similarity_kernel(const float *__restrict__ v, const float *__restrict__ x, const float *__restrict__ vf,
                                                         int has_features, int n_rows, int n_feat, int F, int query, float *__restrict__ sims) {
    const int sub = threadIdx.x & (kGroup - 1);
    const int group = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup, n_groups = (gridDim.x * blockDim.x) / kGroup;
    for (int r = group; r < n_rows; r += n_groups) {
        float part = 0.0f;
        for (int f = sub; f < F; f += kGroup) {
            float a = v[(size_t)r * F + f], q = v[(size_t)query * F + f];
            if (has_features)
                for (int p = 0; p < n_feat; ++p) {
                    const float t = vf[(size_t)p * F + f];
                    a += x[(size_t)r * n_feat + p] * t;
                    q += x[(size_t)query * n_feat + p] * t;
                }
            part += a * q;
        }
        part = group16_sum(part);
        if (sub == 0) sims[r] = part;
    }
}

static int check_model(const rfm_model_view *m) {
    if (!m || m->n_users < 1 || m->n_items < 1 || m->n_factors < 1 || m->n_user_features < 1 || m->n_item_features < 1)
        return RFM_ERR_BAD_ARG;
    if (!m->x_uf || !m->x_if || !m->w_i || !m->w_if || !m->v_u || !m->v_i || !m->v_uf || !m->v_if) return RFM_ERR_BAD_ARG;
    return RFM_OK;
}

static size_t model_bytes(const rfm_model_view *m, int k) {
    const size_t U = m->n_users, I = m->n_items, P = m->n_user_features, Q = m->n_item_features, F = m->n_factors;
    const size_t n[8] = {U * P, I * Q, I, Q, U * F, I * F, P * F, Q * F};
    return n[k] * sizeof(float);
}

static void free_all(void **p, int n) {
    for (int k = 0; k < n; ++k)
        if (p[k]) (void)hipFree(p[k]);
}

// The host entry points of predict / recommend stage the model and their inputs in ONE device allocation per device that is kept
// between calls (grown when a call needs more; released by rfm_release_cache and, above kServeKeepBytes, at the end of the call that
// needed it): a serving loop that calls `recommend` on host buffers then pays its copies and its kernels, not thirteen hipMalloc /
// hipFree pairs per call (measured: 9,936 users x 35 k items top-10, ~1 ms of 2.7).  Calls on one device are serialised.
struct ServeArena { char *ptr = nullptr; size_t bytes = 0; std::mutex mu; };
static ServeArena &serve_arena(int device) {
    static ServeArena arenas[64];
    return arenas[device >= 0 && device < 64 ? device : 0];
}
constexpr size_t kServeKeepBytes = (size_t)1 << 30;
static size_t up256b(size_t x) { return (x + 255) & ~(size_t)255; }
struct Bump {
    char *base; size_t off;
    void *take(size_t bytes) { void *p = base + off; off += up256b(bytes ? bytes : 1); return p; }
};
static int arena_reserve(ServeArena &a, size_t bytes) {
    if (a.bytes >= bytes && a.ptr) return RFM_OK;
    if (a.ptr) { (void)hipFree(a.ptr); a.ptr = nullptr; a.bytes = 0; }
    if (hipMalloc((void **)&a.ptr, bytes) != hipSuccess) { a.ptr = nullptr; return RFM_ERR_HIP; }
    a.bytes = bytes;
    return RFM_OK;
}
static void arena_done(ServeArena &a) {
    if (a.bytes > kServeKeepBytes && a.ptr) { (void)hipFree(a.ptr); a.ptr = nullptr; a.bytes = 0; }
}
static size_t model_total_bytes(const rfm_model_view *h) {
    size_t t = 0;
    for (int k = 0; k < 8; ++k) t += up256b(model_bytes(h, k) ? model_bytes(h, k) : 1);
    return t;
}
static int upload_model_into(const rfm_model_view *h, rfm_model_view *d, Bump &b) {
    *d = *h;
    const float *src[8] = {h->x_uf, h->x_if, h->w_i, h->w_if, h->v_u, h->v_i, h->v_uf, h->v_if};
    const float **dst[8] = {&d->x_uf, &d->x_if, &d->w_i, &d->w_if, &d->v_u, &d->v_i, &d->v_uf, &d->v_if};
    for (int k = 0; k < 8; ++k) {
        const size_t bytes = model_bytes(h, k);
        void *p = b.take(bytes);
        if (hipMemcpy(p, src[k], bytes, hipMemcpyHostToDevice) != hipSuccess) return RFM_ERR_HIP;
        *dst[k] = (const float *)p;
    }
    return RFM_OK;
}

constexpr long long kRecommendChunk = 1024;   // users scored per pass (workspace = chunk * n_items floats)
constexpr long long kFusedChunk = 16384;      // users per pass of the matrix-free path (select_blocks_kernel)

}  // namespace rfm

using namespace rfm;

extern "C" {

int rfm_predict_device(const rfm_model_view *m, int64_t n_pairs, const float *pairs, float *scores, void *hip_stream) {
    int rc = check_model(m);
    if (rc != RFM_OK) return rc;
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs || !scores))) return RFM_ERR_BAD_ARG;
    if (n_pairs == 0) return RFM_OK;
    long long blocks = (n_pairs * kGroup + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    predict_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream>>>(*m, (long long)n_pairs, pairs, scores);
    return hipGetLastError() == hipSuccess ? RFM_OK : RFM_ERR_HIP;
}

int rfm_predict_host(const rfm_model_view *hm, int64_t n_pairs, const float *pairs, float *scores, int device) {
    int rc = check_model(hm);
    if (rc != RFM_OK) return rc;
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs || !scores))) return RFM_ERR_BAD_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return RFM_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RFM_ERR_HIP;
    if (n_pairs == 0) return RFM_OK;
    ServeArena &arena = serve_arena(device);
    std::lock_guard<std::mutex> lock(arena.mu);
    const size_t in_bytes = sizeof(float) * 2 * (size_t)n_pairs, out_bytes = sizeof(float) * (size_t)n_pairs;
    rc = arena_reserve(arena, model_total_bytes(hm) + up256b(in_bytes) + up256b(out_bytes));
    if (rc != RFM_OK) return rc;
    Bump bump{arena.ptr, 0};
    rfm_model_view dm;
    rc = upload_model_into(hm, &dm, bump);
    float *d_pairs = (float *)bump.take(in_bytes), *d_scores = (float *)bump.take(out_bytes);
    if (rc == RFM_OK && hipMemcpy(d_pairs, pairs, in_bytes, hipMemcpyHostToDevice) != hipSuccess) rc = RFM_ERR_HIP;
    if (rc == RFM_OK) rc = rfm_predict_device(&dm, n_pairs, d_pairs, d_scores, nullptr);
    if (rc == RFM_OK && hipMemcpy(scores, d_scores, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = RFM_ERR_HIP;
    if (rc != RFM_OK) (void)hipDeviceSynchronize();      // (nothing of a failed call may still be running on the arena)
    arena_done(arena);
    return rc;
}

static int padded_k(const rfm_model_view *m) {
    const int k = m->has_item_features ? 2 * m->n_factors : m->n_factors;
    return (k + kKT - 1) / kKT * kKT;
}

// workspace: scores [chunk, I] | veff [I, Kp] | bias [I] | ueff [chunk, Kp]
size_t rfm_recommend_workspace_bytes(const rfm_model_view *m, int64_t n_rec_users, int32_t n_rec) {
    if (check_model(m) != RFM_OK || n_rec_users < 0 || n_rec < 1) return 0;
    const size_t chunk = (size_t)(n_rec_users < kRecommendChunk ? (n_rec_users > 0 ? n_rec_users : 1) : kRecommendChunk);
    const size_t kp = (size_t)padded_k(m), I = (size_t)m->n_items;
    return sizeof(float) * (chunk * I + I * kp + I + chunk * kp) + 1024;
}

int rfm_recommend_device(const rfm_model_view *m, int64_t n_users, const float *users, const int64_t *csr_off,
                         const int32_t *csr_items, int32_t n_rec, int32_t filter_previous, float *rec, void *workspace,
                         size_t workspace_bytes, void *hip_stream) {
    int rc = check_model(m);
    if (rc != RFM_OK) return rc;
    if (n_users < 0 || n_rec < 1 || n_rec > m->n_items) return RFM_ERR_BAD_ARG;
    if (n_users == 0) return RFM_OK;
    if (!users || !rec || (filter_previous && (!csr_off || !csr_items))) return RFM_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < rfm_recommend_workspace_bytes(m, n_users, n_rec)) return RFM_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)hip_stream;
    const long long chunk = n_users < kRecommendChunk ? n_users : kRecommendChunk;
    const int kp = padded_k(m);
    const size_t I = (size_t)m->n_items;
    auto up256 = [](size_t x) { return (x + 63) & ~(size_t)63; };                 // sub-buffers on 256-byte boundaries
    float *scores = (float *)workspace;
    float *veff = scores + up256((size_t)chunk * I);
    float *bias = veff + up256(I * kp);
    float *ueff = bias + up256(I);
    build_veff_kernel<<<dim3(512), dim3(256), 0, stream>>>(*m, kp, veff, bias);
    // The usual case -- a short list out of a catalogue of up to 131,072 items -- never writes the score matrix: block maxima out of
    // the GEMM's accumulators, then the few candidate blocks scored again (select_blocks_kernel).  The `scores` area of the workspace
    // (chunk x I floats) holds, for a chunk of up to kFusedChunk users, the maxima (8 bytes per block of 32 items and user), the
    // observed-item bits (4 bytes) and the users' effective factor rows.
    const int n_words = (m->n_items + kBlk - 1) / kBlk;
    if (n_rec <= kTopLocal && n_words <= kSegs && kp <= 512) {
        // (3 n_words + kp floats per user out of 1024 x I: a chunk of at least 8 x 1024 users always fits.)  The three per-block arrays
        // are each rounded up to 64 floats below, so 3 x 63 floats come off the budget FIRST: with a handful of users and a small
        // catalogue the roundings are larger than the users' own share, and a chunk sized without them ran `mask` and `ueff_f` into
        // `veff` (1 user x 42 - 192 items at kp = 32).  What does not fit takes the matrix path below.
        const size_t per_user = 3 * (size_t)n_words + (size_t)kp;
        const size_t area = (size_t)chunk * I, slack = 3 * 63;
        long long fchunk = area > slack ? (long long)((area - slack) / per_user) : 0;
        fchunk = std::min<long long>(std::min<long long>(fchunk, kFusedChunk), n_users);
        if (fchunk >= 1 && (fchunk >= n_users || fchunk >= 64)) {
            if (fchunk < n_users) fchunk &= ~63LL;
            if (3 * up256((size_t)n_words * fchunk) + (size_t)fchunk * kp > area) return RFM_ERR_WORKSPACE;      // (cannot happen: see `slack`)
            float *bmax_val = scores;
            int *bmax_idx = (int *)(bmax_val + up256((size_t)n_words * fchunk));
            unsigned *mask = (unsigned *)(bmax_idx + up256((size_t)n_words * fchunk));
            float *ueff_f = (float *)(mask + up256((size_t)n_words * fchunk));
            for (long long u0 = 0; u0 < n_users; u0 += fchunk) {
                const long long nu = (n_users - u0) < fchunk ? (n_users - u0) : fchunk;
                build_ueff_kernel<<<dim3(256), dim3(256), 0, stream>>>(*m, users, u0, (int)nu, kp, ueff_f);
                if (filter_previous) {
                    if (hipMemsetAsync(mask, 0, sizeof(unsigned) * (size_t)n_words * nu, stream) != hipSuccess) return RFM_ERR_HIP;
                    seen_mask_kernel<<<dim3((unsigned)nu), dim3(256), 0, stream>>>(users, u0, csr_off, csr_items, n_words, mask);
                }
                // block maxima: both operands in registers up to a padded k of 128 (scores_blockmax_reg_kernel), else staged through LDS.
                // Item strips per user tile: enough workgroups for ~8 per CU, at least four item blocks each.
                const unsigned user_tiles = (unsigned)((nu + 127) / 128);
                const unsigned strips = (unsigned)std::max<long long>(1, std::min<long long>((n_words + 3) / 4, (2048 + user_tiles - 1) / user_tiles));
                const dim3 rgrid(strips, user_tiles);
                const unsigned *mk = filter_previous ? mask : nullptr;
                auto reg = [&](auto with_mask, auto without_mask) {
                    if (mk) hipLaunchKernelGGL(with_mask, rgrid, dim3(256), 0, stream, ueff_f, veff, bias, mk, (int)nu, m->n_items, n_words, bmax_val, bmax_idx);
                    else hipLaunchKernelGGL(without_mask, rgrid, dim3(256), 0, stream, ueff_f, veff, bias, mk, (int)nu, m->n_items, n_words, bmax_val, bmax_idx);
                };
                switch (kp) {
                case 32: reg(scores_blockmax_reg_kernel<16, true>, scores_blockmax_reg_kernel<16, false>); break;
                case 64: reg(scores_blockmax_reg_kernel<32, true>, scores_blockmax_reg_kernel<32, false>); break;
                case 96: reg(scores_blockmax_reg_kernel<48, true>, scores_blockmax_reg_kernel<48, false>); break;
                case 128: reg(scores_blockmax_reg_kernel<64, true>, scores_blockmax_reg_kernel<64, false>); break;
                default:
                    scores_blockmax_kernel<<<dim3((unsigned)((I + 63) / 64), (unsigned)((nu + 63) / 64)), dim3(256), 0, stream>>>(
                        ueff_f, veff, bias, mk, (int)nu, m->n_items, kp, n_words, bmax_val, bmax_idx);
                }
                // the ranking: one wavefront per user while a lane can hold its share of the block maxima in registers, else one workgroup
                if (n_words <= 64 * 8)
                    select_blocks_wave_kernel<8><<<dim3((unsigned)((nu + 3) / 4)), dim3(256), 0, stream>>>(users, u0, (int)nu, m->n_items, kp, ueff_f, veff, bias, mk,
                                                                                                             n_words, bmax_val, bmax_idx, n_rec, rec);
                else if (n_words <= 64 * 24)
                    select_blocks_wave_kernel<24><<<dim3((unsigned)((nu + 3) / 4)), dim3(256), 0, stream>>>(users, u0, (int)nu, m->n_items, kp, ueff_f, veff, bias, mk,
                                                                                                              n_words, bmax_val, bmax_idx, n_rec, rec);
                else
                    select_blocks_kernel<<<dim3((unsigned)nu), dim3(256), 0, stream>>>(users, u0, (int)nu, m->n_items, kp, ueff_f, veff, bias, mk, n_words, bmax_val,
                                                                                        bmax_idx, n_rec, rec);
            }
            return hipGetLastError() == hipSuccess ? RFM_OK : RFM_ERR_HIP;
        }
    }
    for (long long u0 = 0; u0 < n_users; u0 += chunk) {
        const long long nu = (n_users - u0) < chunk ? (n_users - u0) : chunk;
        build_ueff_kernel<<<dim3(64), dim3(256), 0, stream>>>(*m, users, u0, (int)nu, kp, ueff);
        scores_mfma_kernel<<<dim3((unsigned)((I + 63) / 64), (unsigned)((nu + 63) / 64)), dim3(256), 0, stream>>>(
            ueff, veff, bias, (int)nu, m->n_items, kp, scores);
        // (threshold selection holds its candidates in LDS: worst case (n_rec - 1) x segment length + 1 of them)
        if (n_rec <= kTopLocal && (long long)(n_rec - 1) * ((m->n_items + kSegs - 1) / kSegs) + 1 <= kCands)
            topn_select_kernel<<<dim3((unsigned)nu), dim3(256), 0, stream>>>(users, u0, m->n_items, csr_off, csr_items, filter_previous,
                                                                              n_rec, -1, scores, rec);
        else
            topn_kernel<<<dim3((unsigned)nu), dim3(256), 0, stream>>>(users, u0, m->n_items, csr_off, csr_items, filter_previous,
                                                                       n_rec, scores, rec);
    }
    return hipGetLastError() == hipSuccess ? RFM_OK : RFM_ERR_HIP;
}

int rfm_recommend_host(const rfm_model_view *hm, int64_t n_users, const float *users, const int64_t *csr_off,
                       const int32_t *csr_items, int32_t n_rec, int32_t filter_previous, float *rec, int device) {
    int rc = check_model(hm);
    if (rc != RFM_OK) return rc;
    if (n_users < 0 || n_rec < 1 || n_rec > hm->n_items) return RFM_ERR_BAD_ARG;
    if (n_users > 0 && (!users || !rec || !csr_off || !csr_items)) return RFM_ERR_BAD_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return RFM_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RFM_ERR_HIP;
    if (n_users == 0) return RFM_OK;
    ServeArena &arena = serve_arena(device);
    std::lock_guard<std::mutex> lock(arena.mu);
    const size_t nnz = (size_t)csr_off[hm->n_users];
    const size_t ws = rfm_recommend_workspace_bytes(hm, n_users, n_rec);
    const size_t sizes[5] = {sizeof(float) * (size_t)n_users, sizeof(int64_t) * ((size_t)hm->n_users + 1),
                             sizeof(int32_t) * (nnz ? nnz : 1), sizeof(float) * (size_t)n_users * (size_t)n_rec, ws};
    size_t total = model_total_bytes(hm);
    for (int k = 0; k < 5; ++k) total += up256b(sizes[k]);
    rc = arena_reserve(arena, total);
    if (rc != RFM_OK) return rc;
    Bump bump{arena.ptr, 0};
    rfm_model_view dm;
    rc = upload_model_into(hm, &dm, bump);
    void *d[5];
    for (int k = 0; k < 5; ++k) d[k] = bump.take(sizes[k]);
    const void *srcs[3] = {users, csr_off, csr_items};
    for (int k = 0; k < 3 && rc == RFM_OK; ++k)
        if (hipMemcpy(d[k], srcs[k], k == 2 ? sizeof(int32_t) * nnz : sizes[k], hipMemcpyHostToDevice) != hipSuccess) rc = RFM_ERR_HIP;
    if (rc == RFM_OK)
        rc = rfm_recommend_device(&dm, n_users, (const float *)d[0], (const int64_t *)d[1], (const int32_t *)d[2],
                                  n_rec, filter_previous, (float *)d[3], d[4], ws, nullptr);
    if (rc == RFM_OK && hipMemcpy(rec, d[3], sizes[3], hipMemcpyDeviceToHost) != hipSuccess) rc = RFM_ERR_HIP;
    if (rc != RFM_OK) (void)hipDeviceSynchronize();      // (nothing of a failed call may still be running on the arena)
    arena_done(arena);
    return rc;
}

// (called by rfm_release_cache, rfm_api.hip)
__attribute__((visibility("hidden"))) void rfm_serve_release_cache(void) {
    int n = 0, cur = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return;
    (void)hipGetDevice(&cur);
    for (int dev = 0; dev < n && dev < 64; ++dev) {
        ServeArena &a = serve_arena(dev);
        std::lock_guard<std::mutex> lock(a.mu);
        if (a.ptr) { (void)hipSetDevice(dev); (void)hipFree(a.ptr); a.ptr = nullptr; a.bytes = 0; }
    }
    (void)hipSetDevice(cur);
}

int rfm_similar_host(const rfm_model_view *hm, int32_t kind, int32_t index, int32_t n, float *out, int device) {
    int rc = check_model(hm);
    if (rc != RFM_OK) return rc;
    const int rows = kind == RFM_SIMILAR_ITEMS ? hm->n_items : hm->n_users;
    if ((kind != RFM_SIMILAR_ITEMS && kind != RFM_SIMILAR_USERS) || index < 0 || index >= rows || n < 1 || n > rows - 1 || !out)
        return RFM_ERR_BAD_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return RFM_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RFM_ERR_HIP;
    const bool items = kind == RFM_SIMILAR_ITEMS;
    const int F = hm->n_factors, nf = items ? hm->n_item_features : hm->n_user_features;
    const int has = items ? hm->has_item_features : hm->has_user_features;
    const float *hv = items ? hm->v_i : hm->v_u, *hx = items ? hm->x_if : hm->x_uf, *hvf = items ? hm->v_if : hm->v_uf;
    void *a[6] = {nullptr};
    const size_t bytes[6] = {sizeof(float) * (size_t)rows * F, sizeof(float) * (size_t)rows * nf, sizeof(float) * (size_t)nf * F,
                             sizeof(float) * (size_t)rows, sizeof(float) * (size_t)n, 0};
    const void *src[3] = {hv, hx, hvf};
    for (int k = 0; k < 5 && rc == RFM_OK; ++k) {
        if (hipMalloc(&a[k], bytes[k]) != hipSuccess) rc = RFM_ERR_HIP;
        else if (k < 3 && hipMemcpy(a[k], src[k], bytes[k], hipMemcpyHostToDevice) != hipSuccess) rc = RFM_ERR_HIP;
    }
    if (rc == RFM_OK) {
        similarity_kernel<<<dim3(512), dim3(256)>>>((const float *)a[0], (const float *)a[1], (const float *)a[2], has, rows, nf, F, index, (float *)a[3]);
        if (n <= kTopLocal && (long long)(n - 1) * ((rows + kSegs - 1) / kSegs) + 1 <= kCands)
            topn_select_kernel<<<dim3(1), dim3(256)>>>(nullptr, 0, rows, nullptr, nullptr, 0, n, index, (float *)a[3], (float *)a[4]);
        else {                                                            // long lists: knock the query out, multi-round selection
            const float ninf = -INFINITY;
            if (hipMemcpy((float *)a[3] + index, &ninf, sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = RFM_ERR_HIP;
            const float zero_user = 0.0f;
            if (hipMalloc(&a[5], sizeof(float)) != hipSuccess || hipMemcpy(a[5], &zero_user, sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = RFM_ERR_HIP;
            if (rc == RFM_OK) topn_kernel<<<dim3(1), dim3(256)>>>((const float *)a[5], 0, rows, nullptr, nullptr, 0, n, (float *)a[3], (float *)a[4]);
        }
        if (rc == RFM_OK && hipGetLastError() != hipSuccess) rc = RFM_ERR_HIP;
    }
    if (rc == RFM_OK && hipMemcpy(out, a[4], bytes[4], hipMemcpyDeviceToHost) != hipSuccess) rc = RFM_ERR_HIP;
    free_all(a, 6);
    return rc;
}

}  // extern "C"
