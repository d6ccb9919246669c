/*
 * trc_sobol.h -- Sobol' generator matrices and pixel-interval tables, GENERATED at run time.
 *
 * The reference's pbrt::SobolSampler (RT_Metal/Metal/SobolSampler.hh:26-167, wired at Render.metal:529-530, commented
 * out there) reads four constant tables from RT_Metal/Metal/Sobolmatrices.metal (pbrt-v3 core/sobolmatrices.cpp):
 *   SobolMatrices32   [1024 dims][52 columns]  :69      the generator matrices, top 32 bits of each column
 *   VdCSobolMatrices    [m = 1..25][52]        :26701   used by SobolIntervalToIndex (SobolSampler.hh:126-148)
 *   VdCSobolMatricesInv [m = 1..26][52]        :26827
 * None of that data is stored in this repository.  The tables are functions of published mathematics and are
 * recomputed here:
 *   - column k of dimension d is the direction number v_k = m_k / 2^(k+1) of Sobol's recurrence (Bratley & Fox,
 *     ACM TOMS 14(1), Algorithm 659) over the primitive polynomial and initial m_1..m_s of S. Joe and F. Y. Kuo,
 *     "Constructing Sobol sequences with better two-dimensional projections", SIAM J. Sci. Comput. 30 (2008),
 *     data set new-joe-kuo-6.21201 (the one the reference's table cites at Sobolmatrices.metal:59-63); dimension 0
 *     is the van der Corput sequence (identity matrix).  The first TRC_SOBOL_DIMS dimensions are provided: tracePath
 *     and traceMIS consume two per bounce (Render.metal:447,314), i.e. 16 at the reference's depth of 8.
 *   - the interval tables follow from dimensions 0 and 1 alone (L. Gruenschloss, M. Raab, A. Keller, "Enumerating
 *     Quasi-Monte Carlo Point Sequences in Elementary Intervals", MCQMC 2010): the top m bits of both coordinates
 *     are a GF(2)-linear, invertible function of the low 2m index bits; VdC holds the images of the higher index
 *     bits, Inv the inverse of the low block.
 * tests/test_sobol.py pins every generated word to the reference's tables through tests/golden/sobol_tables.json
 * (CRC-32 per dimension / per m, produced by tests/golden/make_sobol_fixture.py from the reference's file).
 *
 * Host-side C (no device code): libtracer_amd.so uploads the result, liboracle.so reads it directly.
 */
#ifndef TRC_SOBOL_H
#define TRC_SOBOL_H

#include <stdint.h>
#include <string.h>

#define TRC_SOBOL_DIMS         40u   /* dimensions generated; the reference's table has 1024 (Sobolmatrices.hh:42) */
#define TRC_SOBOL_MATRIX_SIZE  52u   /* columns per dimension (Sobolmatrices.hh:43) */
#define TRC_SOBOL_MAX_LOG2RES  26u   /* 2m <= 52 */

/* Joe & Kuo (2008) parameters of dimensions 2..40: degree s, polynomial coefficients a (the s-1 inner bits),
 * initial direction integers m_1..m_s. */
static const uint16_t trc_sobol_joe_kuo[TRC_SOBOL_DIMS - 1][10] = {
    /* s, a, m_1 ... m_s */
    {1, 0, 1},
    {2, 1, 1, 3},
    {3, 1, 1, 3, 1},
    {3, 2, 1, 1, 1},
    {4, 1, 1, 1, 3, 3},
    {4, 4, 1, 3, 5, 13},
    {5, 2, 1, 1, 5, 5, 17},
    {5, 4, 1, 1, 5, 5, 5},
    {5, 7, 1, 1, 7, 11, 19},
    {5, 11, 1, 1, 5, 1, 1},
    {5, 13, 1, 1, 1, 3, 11},
    {5, 14, 1, 3, 5, 5, 31},
    {6, 1, 1, 3, 3, 9, 7, 49},
    {6, 13, 1, 1, 1, 15, 21, 21},
    {6, 16, 1, 3, 1, 13, 27, 49},
    {6, 19, 1, 1, 1, 15, 7, 5},
    {6, 22, 1, 3, 1, 15, 13, 25},
    {6, 25, 1, 1, 5, 5, 19, 61},
    {7, 1, 1, 3, 7, 11, 23, 15, 103},
    {7, 4, 1, 3, 7, 13, 13, 15, 69},
    {7, 7, 1, 1, 3, 13, 7, 35, 63},
    {7, 8, 1, 3, 5, 9, 1, 25, 53},
    {7, 14, 1, 3, 1, 13, 9, 35, 107},
    {7, 19, 1, 3, 1, 5, 27, 61, 31},
    {7, 21, 1, 1, 5, 11, 19, 41, 61},
    {7, 28, 1, 3, 5, 3, 3, 13, 69},
    {7, 31, 1, 1, 7, 13, 1, 19, 1},
    {7, 32, 1, 3, 7, 5, 13, 19, 59},
    {7, 37, 1, 1, 3, 9, 25, 29, 41},
    {7, 41, 1, 3, 5, 13, 23, 1, 55},
    {7, 42, 1, 3, 7, 3, 13, 59, 17},
    {7, 50, 1, 3, 1, 3, 5, 53, 69},
    {7, 55, 1, 1, 5, 5, 23, 33, 13},
    {7, 56, 1, 1, 7, 7, 1, 61, 123},
    {7, 59, 1, 1, 7, 9, 13, 61, 49},
    {7, 62, 1, 3, 3, 5, 3, 55, 33},
    {8, 14, 1, 3, 1, 15, 31, 13, 49, 245},
    {8, 21, 1, 3, 5, 15, 31, 59, 63, 97},
    {8, 22, 1, 3, 1, 11, 11, 11, 77, 249},
};

/* the 52 columns of dimension `dim` as 52-bit binary fractions (bit 51 = 1/2) */
static inline void trc_sobol_columns52(uint32_t dim, uint64_t cols[TRC_SOBOL_MATRIX_SIZE]) {
    const uint32_t N = TRC_SOBOL_MATRIX_SIZE;
    if (dim == 0) {                                           /* van der Corput: identity */
        for (uint32_t k = 0; k < N; ++k) cols[k] = 1ull << (N - 1 - k);
        return;
    }
    const uint16_t* row = trc_sobol_joe_kuo[dim - 1];
    const uint32_t s = row[0], a = row[1];
    uint64_t m[TRC_SOBOL_MATRIX_SIZE];
    for (uint32_t k = 0; k < s; ++k) m[k] = row[2 + k];
    for (uint32_t k = s; k < N; ++k) {                        /* m_k = 2 a_1 m_{k-1} ^ ... ^ 2^s m_{k-s} ^ m_{k-s} */
        uint64_t v = m[k - s] ^ (m[k - s] << s);
        for (uint32_t i = 1; i < s; ++i)
            if ((a >> (s - 1 - i)) & 1u) v ^= m[k - i] << i;
        m[k] = v;
    }
    for (uint32_t k = 0; k < N; ++k) cols[k] = m[k] << (N - 1 - k);   /* m_k < 2^(k+1) */
}

/* SobolMatrices32 layout: out[dim * 52 + column] = top 32 bits of the column */
static inline void trc_sobol_matrices32(uint32_t* out /* [TRC_SOBOL_DIMS * 52] */) {
    uint64_t cols[TRC_SOBOL_MATRIX_SIZE];
    for (uint32_t d = 0; d < TRC_SOBOL_DIMS; ++d) {
        trc_sobol_columns52(d, cols);
        for (uint32_t k = 0; k < TRC_SOBOL_MATRIX_SIZE; ++k) out[d * TRC_SOBOL_MATRIX_SIZE + k] = (uint32_t)(cols[k] >> 20);
    }
}

/* VdCSobolMatrices[m-1] and VdCSobolMatricesInv[m-1] for a 2^m x 2^m pixel grid, 1 <= m <= 26; unused entries 0.
 * Pixel word of an index = (top m bits of dimension 0) << m | (top m bits of dimension 1). Returns 0 on success. */
static inline int trc_sobol_interval_tables(uint32_t m, uint64_t vdc[TRC_SOBOL_MATRIX_SIZE], uint64_t inv[TRC_SOBOL_MATRIX_SIZE]) {
    const uint32_t N = TRC_SOBOL_MATRIX_SIZE;
    memset(vdc, 0, N * sizeof(uint64_t));
    memset(inv, 0, N * sizeof(uint64_t));
    if (m < 1 || m > TRC_SOBOL_MAX_LOG2RES) return -1;
    uint64_t c0[TRC_SOBOL_MATRIX_SIZE], c1[TRC_SOBOL_MATRIX_SIZE], word[TRC_SOBOL_MATRIX_SIZE];
    trc_sobol_columns52(0, c0);
    trc_sobol_columns52(1, c1);
    for (uint32_t k = 0; k < N; ++k) word[k] = ((c0[k] >> (N - m)) << m) | (c1[k] >> (N - m));
    for (uint32_t c = 0; 2 * m + c < N; ++c) vdc[c] = word[2 * m + c];
    /* Gauss-Jordan over GF(2) on the 2m x 2m block of the low index bits: row k = (pixel word, index word) */
    const uint32_t n = 2 * m;
    uint64_t pw[TRC_SOBOL_MATRIX_SIZE], iw[TRC_SOBOL_MATRIX_SIZE];
    for (uint32_t k = 0; k < n; ++k) { pw[k] = word[k]; iw[k] = 1ull << k; }
    for (uint32_t bit = 0; bit < n; ++bit) {
        uint32_t p = bit;
        while (p < n && !((pw[p] >> bit) & 1ull)) ++p;
        if (p == n) return -2;                                 /* cannot happen: (0,2)-sequence in base 2 */
        uint64_t t = pw[p]; pw[p] = pw[bit]; pw[bit] = t;
        t = iw[p]; iw[p] = iw[bit]; iw[bit] = t;
        for (uint32_t r = 0; r < n; ++r)
            if (r != bit && ((pw[r] >> bit) & 1ull)) { pw[r] ^= pw[bit]; iw[r] ^= iw[bit]; }
    }
    for (uint32_t bit = 0; bit < n; ++bit) inv[bit] = iw[bit];  /* pw[bit] == 1 << bit */
    return 0;
}

#endif /* TRC_SOBOL_H */
