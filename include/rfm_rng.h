/* rfm_rng.h -- counter-based negative-draw stream and on-the-fly epoch permutation.
 *
 * These two functions are NEW DESIGN (the reference has nothing like them): the reference
 * draws negatives from one process-global, strictly serial MT19937 stream
 * (rankfm/_rankfm.pyx:182,251; rankfm/mt19937ar/mt19937ar.c:56-57) and shuffles one host
 * index array with numpy's global RNG (rankfm/_rankfm.pyx:197,227).  Neither can be
 * evaluated by thousands of wavefronts at once, so the MI355X engine keys every draw by
 * (seed, epoch, interaction row, attempt) and every shuffled position by (seed, epoch,
 * position).  Any wavefront can then compute any draw with no shared state, and the CPU
 * oracle (oracle/rfm_oracle.c, rng_mode = RFM_RNG_COUNTER) reproduces the GPU's negatives
 * exactly.  The spec lives here, in plain C, and is included by both sides.
 */
#ifndef RFM_RNG_H
#define RFM_RNG_H
#include <stdint.h>

#if defined(__HIPCC__)
#define RFM_HD __host__ __device__ __forceinline__
#else
#define RFM_HD static inline
#endif

/* 32-bit finaliser (two odd multiplies, three xor-shifts): a bijection on uint32. */
RFM_HD uint32_t rfm_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* key of one training epoch (absolute epoch index, so fit_partial-style resumption with
 * epoch_begin > 0 continues the stream instead of replaying it) */
RFM_HD uint32_t rfm_epoch_key(uint32_t seed, uint32_t epoch) {
    return rfm_mix32(seed ^ (0x9E3779B9U * (epoch + 1U)));
}

/* key of one interaction row inside an epoch */
RFM_HD uint32_t rfm_row_key(uint32_t epoch_key, uint32_t row) {
    return rfm_mix32(epoch_key ^ rfm_mix32(row + 0x632BE5ABU));
}

/* t-th raw 32-bit draw of a row (t = 0, 1, 2, ... counts every attempt, rejected or not) */
RFM_HD uint32_t rfm_draw(uint32_t row_key, uint32_t t) {
    return rfm_mix32(row_key + 0x9E3779B9U * (t + 1U));
}

/* map a raw draw onto [0, n_items): multiply-high (no division on the device).  The MT
 * mode of the oracle / serial kernel keeps the reference's `% I` (rankfm/_rankfm.pyx:251). */
RFM_HD uint32_t rfm_draw_to_item(uint32_t raw, uint32_t n_items) {
    return (uint32_t)(((uint64_t)raw * (uint64_t)n_items) >> 32);
}

/* Pseudo-random bijection on [0, n): four multiply-add / xor-shift rounds on the smallest
 * power-of-two domain >= n, cycle-walking back into range.  `bits` = ceil(log2(n)) clamped
 * to >= 2 (see rfm_perm_bits).  Replaces the materialised, host-shuffled index array. */
RFM_HD uint32_t rfm_perm_bits(uint32_t n) {
    uint32_t b = 2;
    while (b < 32 && (1ULL << b) < (uint64_t)n) ++b;
    return b;
}

RFM_HD uint32_t rfm_perm(uint32_t pos, uint32_t n, uint32_t bits, uint32_t epoch_key) {
    const uint32_t mask = (bits >= 32) ? 0xFFFFFFFFU : ((1U << bits) - 1U);
    const uint32_t s1 = (bits + 1U) >> 1, s2 = (bits >> 1) > 0 ? (bits >> 1) : 1U;
    const uint32_t k0 = rfm_mix32(epoch_key ^ 0xA511E9B3U), k1 = rfm_mix32(epoch_key ^ 0x1B873593U);
    const uint32_t k2 = rfm_mix32(epoch_key ^ 0xCC9E2D51U), k3 = rfm_mix32(epoch_key ^ 0x38B34AE5U);
    uint32_t x = pos;
    do {
        x = (x * 0x9E3779B1U + k0) & mask; x ^= x >> s1;
        x = (x * 0x85EBCA77U + k1) & mask; x ^= x >> s2;
        x = (x * 0xC2B2AE3DU + k2) & mask; x ^= x >> s1;
        x = (x * 0x27D4EB2FU + k3) & mask; x ^= x >> s2;
    } while (x >= n);
    return x;
}

#endif /* RFM_RNG_H */
