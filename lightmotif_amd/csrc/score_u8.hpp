// score_u8.hpp -- `Score<u8, A, C>` for C = 32: the scores of a DiscreteMatrix
// (lightmotif/src/pwm/mod.rs:754-791), materialised as a u8 StripedScores matrix.
//
// Reference bodies: Generic adds the M weights with `+=` on u8 (pli/mod.rs:98-102, wrapping
// in release builds), AVX2 / SSE2 / NEON with saturating byte adds (avx2.rs:336
// `_mm256_adds_epu8`).  The weights are non-negative integers, so both are functions of the
// EXACT integer sum: min(sum, 255) and sum mod 256.  The exact sum is what the packed
// 16-bit rotating-accumulator scan of score_prefilter.hpp computes (M * 255 < 65536: no
// carry between the halves), so this kernel is that scan with the DiscreteMatrix's own
// weights in the LDS image and a byte store where the prefilter compares.
//
// Traffic: 1 B read + 1 B written per cell (SURVEY 8a "32 B in, 32 B out per row").
#pragma once

#include "score_prefilter2.hpp"
#include "score_prefilter_blk.hpp"


namespace lm {

// `image` = prefilter image built from the u8 weights (score_store.hip: launch_score_u8);
// `out` = row `row_begin` of the u8 score matrix, row stride 32.
template <int M, int PF = kScorePF, int WIDE = 0>
__global__ __launch_bounds__(kBlock, 6) void score_c32_u8(
    const uint8_t *__restrict__ seq, const unsigned *__restrict__ image, const int K,
    const unsigned long long row_begin, const unsigned long long row_end,
    const unsigned long long T, const unsigned long long nstreams, uint8_t *__restrict__ out,
    const unsigned wrap_mask)
{
    constexpr int MP = prefilter_mp(M, WIDE);
    constexpr int SHIFT = MP - M;
    constexpr int NP = MP / 2;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        const uint4 *src = reinterpret_cast<const uint4 *>(image);
        const int n4 = prefilter_image_dw(M, K) / 4;
        for (int i = threadIdx.x; i < n4; i += kBlock)
            dst[i] = src[i];
    }
    __syncthreads();
    const char *tab_even = lds_raw;
    const char *tab_odd = tab_even + (size_t)K * prefilter_stride_dw(M, WIDE) * 4;

    const int lane = threadIdx.x & 63;
    const int col = lane & 31;
    unsigned long long stream =
        ((unsigned long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    if (stream >= nstreams)  // idle half-waves redo the last stream (same bytes, same values)
        stream = nstreams - 1;
    unsigned long long o0 = row_begin + stream * T;
    if (o0 + T > row_end)    // the last stream is shifted back and overlaps its neighbour
        o0 = row_end - T;

    const uint8_t *sp = seq + (long long)(o0 - SHIFT) * 32 + col;
    constexpr int PFE = prefilter_lookahead(PF, MP);
    unsigned acc2[NP];
    unsigned sym[MP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        acc2[i] = 0;
#pragma unroll
    for (int j = 0; j < MP; ++j)
        sym[j] = 0;
#pragma unroll
    for (int j = 0; j < PFE; ++j) {
        if (j < SHIFT) {
            if (o0 + j >= (unsigned long long)SHIFT)  // rows before the matrix do not exist; their weight rows are all zero anyway
                sym[j] = sp[j * 32];
        } else {
            sym[j] = sp[j * 32];
        }
    }

    // group 0 completes output 0, group g >= 1 outputs (g-1)*MP+1 .. g*MP
    const unsigned long long ngroups = (T + MP - 1) / MP;  // exact: T = q*MP + 1, >= 2
    uint8_t *op = out + (o0 - row_begin) * 32 + col;
    unsigned mx = 0;
    prefilter_group<M, PFE, PHASE_FIRST, 1, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx, op, wrap_mask);
    op += 32;
    for (unsigned long long g = 1; g + 1 < ngroups; ++g) {
        sp += MP * 32;
        prefilter_group<M, PFE, PHASE_MAIN, 1, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx, op, wrap_mask);
        op += MP * 32;
    }
    sp += MP * 32;
    prefilter_group<M, PFE, PHASE_LAST, 1, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx, op, wrap_mask);
}

// DNA, M >= 2: the same sums from the pair-symbol scan of score_prefilter2.hpp (two input rows
// per LDS lookup, dword symbol loads).  `image` = pair table of the u8 weights
// (prefilter2_pack_image); the sequence matrix must be 4-byte aligned.
template <int M>
__global__ __launch_bounds__(kBlock, prefilter2_waves(M, 5)) void score_c32_u8_pairs(
    const uint8_t *__restrict__ seq, const unsigned *__restrict__ image,
    const unsigned long long row_begin, const unsigned long long row_end,
    const unsigned long long T, const unsigned long long nstreams, uint8_t *__restrict__ out,
    const unsigned wrap_mask)
{
    constexpr int MO = prefilter2_mo(M);
    constexpr int SHIFT = MO - M;
    constexpr int RING = prefilter2_ring(M);
    constexpr int NP = prefilter2_npair(M);
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    lds_zero_based(lds_raw);
    {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        const uint4 *src = reinterpret_cast<const uint4 *>(image);
        constexpr int n4 = prefilter2_image_dw(M) / 4;
        for (int i = threadIdx.x; i < n4; i += kBlock)
            dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    constexpr bool LIN = prefilter2_lut_decode(M, 5);  // lane l of a half-wave on dword l of a block (score_prefilter2.hpp)
    const int col = LIN ? 4 * (lane & 7) + ((lane >> 3) & 3) : lane & 31;
    unsigned long long stream =
        ((unsigned long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    if (stream >= nstreams)  // idle half-waves redo the last stream (same bytes, same values)
        stream = nstreams - 1;
    unsigned long long o0 = row_begin + stream * T;
    if (o0 + T > row_end)
        o0 = row_end - T;

    const long long in0 = (long long)o0 - SHIFT;  // first input row of the stream (padding rows: weight 0)
    const unsigned shq = 8u * (col & 3);
    const uint8_t *spq = seq + (in0 + (col & 3)) * 32 + (col >> 2) * 4;  // this lane's row of a block
    constexpr int NB = RING / 4;
    constexpr int PFB = NB > kPairPFB ? kPairPFB : NB;
    unsigned acc[1][NP];
    unsigned blk[NB];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        acc[0][i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
        blk[j] = 0;
    if (in0 + (long long)(col & 3) >= 0)
        blk[0] = *reinterpret_cast<const unsigned *>(spq);
#pragma unroll
    for (int j = 1; j < PFB; ++j)
        if (4 * j >= SHIFT || in0 + 4 * j + (long long)(col & 3) >= 0)  // SHIFT > 3: block 1 may start before the matrix too
            blk[j] = *reinterpret_cast<const unsigned *>(spq + j * 128);

    // group 0 completes rows 0 and 1, group g >= 1 rows (g-1)*RING + 2 .. g*RING + 1
    const unsigned long long ngroups = (T - 2) / RING + 1;  // exact: T = q*RING + 2
    uint8_t *op = out + (o0 - row_begin) * 32 + col;
    const PairDecode pd = pair_decode_setup<prefilter2_so(M), LIN>();
    PairRows<NP> cur;
    unsigned off0, off1;
    StoreSink<LIN> sink(op, wrap_mask, col);
    pair_begin<M, 5, PFB, LIN>(blk, cur, off0, off1, spq, shq, pd);
    constexpr unsigned FAR = 2u * RING * 32u, NEAR = RING * 32u;  // (pair_items: the group two groups ahead, where the stream has one)
    pair_items<M, 5, 1, PFB, PHASE_FIRST, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, ngroups > 2 ? FAR : NEAR);
    sink.advance(2 * 32);
    for (unsigned long long g = 1; g + 1 < ngroups; ++g) {
        spq += RING * 32;
        pair_items<M, 5, 1, PFB, PHASE_MAIN, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, g + 2 < ngroups ? FAR : NEAR);
        sink.advance(RING * 32);
    }
    if (ngroups > 1) {
        spq += RING * 32;
        pair_items<M, 5, 1, PFB, PHASE_LAST, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, 0u);
    }
}

// Host-side launch shims of the two u8 kernels (one type: K is ignored by the pair scan).
using ScoreU8Launcher = hipError_t (*)(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq,
                                       const unsigned *image, int K, unsigned long long row_begin,
                                       unsigned long long row_end, unsigned long long T,
                                       unsigned long long nstreams, uint8_t *out, unsigned wrap_mask);

template <int M, int WIDE = 0>
hipError_t score_c32_u8_launch(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq,
                               const unsigned *image, int K, unsigned long long row_begin,
                               unsigned long long row_end, unsigned long long T, unsigned long long nstreams,
                               uint8_t *out, unsigned wrap_mask)
{
    hipLaunchKernelGGL((score_c32_u8<M, kScorePF, WIDE>), grid, dim3(kBlock), lds_bytes, stream, seq, image, K, row_begin,
                       row_end, T, nstreams, out, wrap_mask);
    return hipGetLastError();
}

template <int M>
hipError_t score_c32_u8_pairs_launch(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq,
                                     const unsigned *image, int K, unsigned long long row_begin,
                                     unsigned long long row_end, unsigned long long T,
                                     unsigned long long nstreams, uint8_t *out, unsigned wrap_mask)
{
    (void)K;
    hipLaunchKernelGGL((score_c32_u8_pairs<M>), grid, dim3(kBlock), lds_bytes, stream, seq, image, row_begin,
                       row_end, T, nstreams, out, wrap_mask);
    return hipGetLastError();
}

// What a score_inst_*.hip translation unit fills in for its range of motif lengths
// (arrays indexed by M; see score_plan.hip: init_registry).
struct KernelRegistry {
    ScoreC32Launcher (*c32)[kRegistrySlots];
    PrefilterLauncher *pre, *pre2;
    PrefilterLauncher *pre2_protein;  // the pair scan over the 441 residue pairs (K = 21)
    ScoreU8Launcher *u8, *u8_pairs;
    PrefilterMultiLauncher *pre2_multi;  // several motifs per pass (prefilter2_multi(M) > 1)
    // the same kernels for alphabets of more than 16 symbols (WIDE: 8-byte LDS reads, score_kernels.hpp)
    ScoreC32Launcher (*c32w)[kRegistrySlots];
    PrefilterLauncher *prew;
    ScoreU8Launcher *u8w;
    PrefilterLauncher *preblk;  // protein one-symbol scan on 4-row symbol blocks (score_prefilter_blk.hpp)
};

}  // namespace lm
