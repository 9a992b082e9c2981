/*
 * ppg_rng.h — counter-based sampler shared by the HIP kernels and the CPU oracle.
 *
 * Stands in for samplers/independent.cpp:95-103 (one SFMT-19937 stream per worker thread, consumed
 * in data-dependent amounts; blocks are assigned to threads by the OS, so the reference's streams
 * are not reproducible even between two runs of the reference).  Here the n-th draw of a path is a
 * pure function of (seed, pixel, sample index, n): next1D() = ppg_rand(key, dim++), next2D() =
 * (ppg_rand(key, dim), ppg_rand(key, dim+1)), in the same call order as the reference's Li().
 * The three draws of the stochastic spatial filter for recorded vertex i (GP:1753-1755) use dimensions
 * D + 3i .. D + 3i + 2, D = the path's draw count when Li's loop ended (the reference draws them from the
 * stream in vertex order, skipping invalid vertices; any fixed assignment is statistically equivalent and
 * this one lets vertices be committed independently).
 * With next-event estimation the direct-light vertex of GP:1994-2010 is committed inside the loop; its three
 * stochastic-filter draws use dimensions PPG_DIM_NEE_COMMIT + 3·depth (depth = rRec.depth at that vertex) — a separate
 * range, so the main stream's positions do not depend on whether or where such a vertex is committed (its place in the
 * order of the sampling-fraction optimiser's records: include/ppg.h, record code = depth).
 * Floats are built from 23 random mantissa bits like random.cpp:630-639.
 */
#ifndef PPG_RNG_H
#define PPG_RNG_H

#include "ppg_detmath.h"

#define PPG_DIM_NEE_COMMIT 0x40000000u

PPG_HD uint32_t ppg_hash32(uint32_t x) { /* "lowbias32" integer finaliser */
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

PPG_HD uint32_t ppg_path_key(uint64_t seed, uint32_t pixel, uint32_t sample_index) {
    uint32_t k = ppg_hash32((uint32_t)seed ^ 0x9e3779b9u);
    k = ppg_hash32(k ^ (uint32_t)(seed >> 32));
    k = ppg_hash32(k + pixel * 0x9e3779b1u);
    k = ppg_hash32(k ^ (sample_index * 0x85ebca77u + 0xc2b2ae3du));
    return k;
}

PPG_HD float ppg_rand(uint32_t key, uint32_t dim) {
    uint32_t r = ppg_hash32(key ^ ppg_hash32(dim * 0x9e3779b1u + 0x7f4a7c15u));
    return ppg_u2f((r >> 9) | 0x3f800000u) - 1.0f;
}

#endif /* PPG_RNG_H */
