// score_rowpair.hpp -- `score_c32_rp<M, MODE>`: the materialising C = 32 kernel with the two
// halves of a wavefront on ADJACENT output rows of one region ("row pairs").
//
// What is computed is unchanged (lightmotif/src/pli/mod.rs:96-105, Generic):
//
//     out[r][c] = (((0.0f + P[0][s(r,c)]) + P[1][s(r+1,c)]) + ... ) + P[M-1][s(r+M-1,c)]
//
// M sequential IEEE f32 adds in motif order, so the scores are bit-identical to Generic and to
// score_c32<M, 0> (score_kernels.hpp), whose rotating-accumulator scheme this refines.
//
// Why a second layout.  Counters of score_c32<20, 0> on 1 Gbp (profiles/r02_stalls_*.txt): the
// LDS array is busy 22.8 cycles per 64 positions -- 20 for the M weights every position
// gathers (80 B at 256 B/clk/CU) + 14 % in the fill / drain steps of its 61-row streams -- and
// that is ~87 % of the kernel's duration at the clock the chip sustains under this load
// (GRBM_GUI_ACTIVE / duration ~ 1.5-1.65 GHz, not the nominal 2.4): the kernel is bound by the
// LDS gather first and by the HBM write pattern second (trivial kernels writing 128-byte rows
// from two distant streams per wavefront: 1.01 ms; the same rows from ONE region per wavefront,
// 256 contiguous bytes per store: 0.92 ms, tools/kbench/mix_bench `mix_rows` vs `mix_rows_w`).
// Both point the same way:
//
//   * lane (h, c), h = lane >> 5, owns column c and the output rows o0 + 2i + h of the
//     wavefront's region [o0, o0 + R).  Every lane still walks ALL input rows of the region --
//     half 1 one row ahead of half 0 -- but at step t it holds only the H = ceil(M/2) outputs
//     i with 0 <= t - 2i < M in flight and adds P[t - 2i][s] to each: all of one parity of j.
//     The PSSM is staged in LDS as two transposed tables (even j, odd j), so a step fetches
//     H floats instead of M and a pair of steps the same 4*M bytes as before;
//   * output i of both halves completes at step 2i + M - 1: ONE store instruction writes rows
//     o0 + 2i and o0 + 2i + 1 = 256 contiguous bytes, and a wavefront sweeps one compact region;
//   * the fill / drain steps are per REGION, not per half-wave stream: at the same rows per
//     lane the LDS overhead halves (R = 120: 21.4 instead of 22.8 cycles per 64 positions);
//   * H accumulators instead of M: ~44 VGPRs at M = 20 (8 waves per SIMD instead of 7), and
//     motifs up to M = 64 fit the unrolled scheme (score_c32 stops at 36).
//
// Unrolling.  G = 2H steps form a group: output u of the group (slot u) starts at step 2u and
// completes at step 2u + M - 1, i.e. in the NEXT group for u >= 1, so every (step, slot) pair of
// a group has a compile-time weight index j = k - 2u (+ G for the previous group's output) and
// register.  A region is R = q * G rows: one FIRST group (no previous outputs), q - 1 MAIN
// groups, one LAST group (no new outputs; G - 2 or G - 3 steps).
#pragma once

#include "score_kernels.hpp"

namespace lm {

constexpr int kMaxRpM = 64;  // largest motif score_c32_rp is instantiated for

#ifndef LM_RP_SCHED_FENCE
#define LM_RP_SCHED_FENCE 32  // motifs longer than this get scheduling fences between steps
#endif

constexpr int rp_slots(int m) { return (m + 1) / 2; }           // H: outputs in flight per lane
constexpr int rp_group(int m) { return 2 * rp_slots(m); }       // G: steps per unrolled group
// floats per (parity, symbol) row of the LDS image: 4 * odd >= H (16-byte reads, rows of the
// <= 16 symbols a 16-lane group can address in distinct 4-bank slots)
constexpr int rp_table_stride(int m) { return 4 * (((rp_slots(m) + 3) / 4) | 1); }
constexpr size_t rp_table_floats(int m, int k) { return (size_t)2 * k * rp_table_stride(m); }
// steps of the LAST group: the previous group's outputs complete at k = 2u - 1 (M even) or
// 2u - 2 (M odd), u = 1 .. H-1
constexpr int rp_last_steps(int m) { return rp_slots(m) < 2 ? 0 : (m % 2 == 0 ? rp_group(m) - 2 : rp_group(m) - 3); }

// Host side: the LDS image [parity][symbol][TS] of an M x K dense PSSM (row-major, stride k).
inline void rp_build_table(const float *pssm, int m, int k, float *image)
{
    const int ts = rp_table_stride(m);
    for (size_t i = 0; i < rp_table_floats(m, k); ++i)
        image[i] = 0.0f;
    for (int j = 0; j < m; ++j)
        for (int s = 0; s < k; ++s)
            image[((size_t)(j & 1) * k + s) * ts + (j >> 1)] = pssm[(size_t)j * k + s];
}

// N consecutive floats of one LDS row: whole 16-byte reads, then 8 and / or 4 bytes
template <int N, int NW>
__device__ __forceinline__ void rp_fetch(float (&w)[NW], const char *__restrict__ row)
{
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(row + 16 * q);
        w[4 * q + 0] = v.x;
        w[4 * q + 1] = v.y;
        w[4 * q + 2] = v.z;
        w[4 * q + 3] = v.w;
    }
    if (N % 4 >= 2) {
        const float2 v = *reinterpret_cast<const float2 *>(row + 16 * (N / 4));
        w[4 * (N / 4) + 0] = v.x;
        w[4 * (N / 4) + 1] = v.y;
    }
    if (N % 2 == 1)
        w[N - 1] = *reinterpret_cast<const float *>(row + 4 * (N - 1));
}

// Look-ahead of the symbol loads.  A group requests symbols of the NEXT group, which may be
// the LAST one with only rp_last_steps(M) steps: the distance is bounded by that, so that no
// request ever lies past the region's last input row (for the last region of the matrix that
// is the last wrap row, seq.rs:373-378 -- nothing may be read behind it).
constexpr int rp_pf_bytes(int m, int pf)
{
    const int g = rp_group(m), l = rp_last_steps(m);
    const int cap = g - 1 < l ? g - 1 : l;
    return pf < cap ? pf : (cap > 0 ? cap : 0);
}
constexpr int rp_pf_blocks(int m)
{
    const int nb = rp_group(m) / 4, l = rp_last_steps(m) / 4;
    const int a = nb < 3 ? nb : 3;
    return a < l ? a : l;
}
// dword symbol loads (quad-gathered, score_kernels.hpp quad_symbol): whole 4-row blocks per group
constexpr bool rp_quad_loads(int m) { return rp_group(m) % 4 == 0 && rp_pf_blocks(m) >= 1; }

// One group of one lane.  `sp`: symbol source of the group's step 0 (QL: this lane's row of
// block 0); `op`: output cell of the group's slot 0 (row o0 + 2 * g * H + h); `sym`: symbol ring
// (bytes, or dword blocks with QL).  `last_off`: QL + LAST group only -- greatest byte offset
// from `sp` this lane may load from (its row of the region's last input row).
template <int M, int MODE, int PHASE, int QL, int PF>
__device__ __forceinline__ void rp_group_steps(float (&acc)[rp_slots(M)], unsigned (&sym)[rp_group(M)],
                                               const uint8_t *__restrict__ sp, const char *__restrict__ tab,
                                               const unsigned parity_bytes, float *__restrict__ op,
                                               const unsigned shq, float &best_v, const int last_off)
{
    constexpr int H = rp_slots(M), G = rp_group(M);
    constexpr unsigned TSB = rp_table_stride(M) * 4;
    constexpr int NE = (M + 1) / 2, NO = M / 2;  // weights with even / odd j
    constexpr int NW = (NE + 3) / 4 * 4;
    constexpr int STEPS = PHASE == PHASE_LAST ? rp_last_steps(M) : G;
    constexpr int NB = G / 4 > 0 ? G / 4 : 1, PFB = rp_pf_blocks(M);
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
#if LM_RP_SCHED_FENCE
        // long motifs: keep the scheduler from hoisting the LDS reads of many steps above their
        // adds (M = 48 needed 2.6 KB of scratch per lane without it)
        if (M > LM_RP_SCHED_FENCE && k % 2 == 0)
            __builtin_amdgcn_sched_barrier(0);
#endif
        // (1) this step's symbol; request the one PF steps (QL: PFB blocks) ahead
        unsigned s_now;
        if (QL) {
            const unsigned d = sym[k / 4];
            s_now = (k % 4 == 0)   ? quad_symbol<0>(d, shq)
                    : (k % 4 == 1) ? quad_symbol<1>(d, shq)
                    : (k % 4 == 2) ? quad_symbol<2>(d, shq)
                                   : quad_symbol<3>(d, shq);
            if (k % 4 == 3) {
                const int nb = k / 4 + PFB;  // block index relative to this group's block 0
                if (PHASE != PHASE_LAST) {
                    sym[nb % NB] = *reinterpret_cast<const unsigned *>(sp + nb * 128);
                } else if (nb * 4 < STEPS) {
                    // the block's rows 4nb + ql may pass the last input row: clamp to that row
                    // (the symbols of steps >= STEPS are never used)
                    const int off = nb * 128 < last_off ? nb * 128 : last_off;
                    sym[nb % NB] = *reinterpret_cast<const unsigned *>(sp + off);
                }
            }
        } else if (PF > 0) {
            s_now = sym[k % G];
            if (PHASE != PHASE_LAST || k + PF < STEPS)
                sym[(k + PF) % G] = sp[(k + PF) * 32];
        } else {
            s_now = sp[k * 32];
        }
        // (2) the weights of this step's parity for that symbol
        float w[NW];
        const char *row = static_cast<const char *>(__builtin_assume_aligned(
            tab + __umul24(s_now, TSB) + ((k & 1) ? parity_bytes : 0u), 16));
        if ((k & 1) == 0)
            rp_fetch<NE, NW>(w, row);
        else if (NO > 0)
            rp_fetch<(NO > 0 ? NO : 1), NW>(w, row);
        // (3) slot u: output started at step 2u of this group (j = k - 2u) or of the previous
        // one (j = k - 2u + G)
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const bool cur = k >= 2 * u;
            const int j = cur ? k - 2 * u : k - 2 * u + G;
            if (j >= M)
                continue;  // odd M: the slot idles one step between two outputs
            if (PHASE == PHASE_FIRST && !cur)
                continue;  // no outputs before the region
            if (PHASE == PHASE_LAST && cur)
                continue;  // no outputs after it
            if (j == 0)
                acc[u] = 0.0f + w[0];  // T::default() + P[0][s]   (pli/mod.rs:98,101)
            else
                acc[u] = acc[u] + w[j >> 1];
            if (j == M - 1) {
                const float score = acc[u];
                float *cell = op + (cur ? u : u - H) * 64;  // 2 rows of 32 floats per output index
                if (LM_SCORE_NT_STORE)
                    __builtin_nontemporal_store(score, cell);
                else
                    *cell = score;
                if (MODE == MODE_STORE_ARGMAX)
                    best_v = __builtin_fmaxf(best_v, score);  // value only, located afterwards
            }
        }
    }
}

#ifndef LM_RP_MIN_WAVES
#define LM_RP_MIN_WAVES(M) ((M) <= 26 ? 8 : (M) <= 40 ? 6 : 4)
#endif

// C = 32, sequence stride 32 B, score stride 32 floats.  One wavefront per region of R = q * G
// rows; workgroup = 4 wavefronts = 4 consecutive regions.  The last region is shifted back to
// end at row_end (identical values where it overlaps the one before); idle wavefronts of the
// last workgroup redo it.  Requires row_end - row_begin >= R.
template <int M, int MODE, int QLREQ = 1, int PF = 12, int BLK = 256, int MINW = LM_RP_MIN_WAVES(M)>
__global__ __launch_bounds__(BLK, MINW) void score_c32_rp(
    const uint8_t *__restrict__ seq, const float *__restrict__ table, const int K,
    const unsigned long long row_begin, const unsigned long long row_end, const unsigned long long R,
    const unsigned long long nregions, float *__restrict__ out, const FusedOut fo)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    constexpr int H = rp_slots(M), G = rp_group(M);
    constexpr int QL = (QLREQ && rp_quad_loads(M)) ? 1 : 0;
    {
        float4 *dst = reinterpret_cast<float4 *>(lds_raw);
        const float4 *src = reinterpret_cast<const float4 *>(table);
        const int n4 = 2 * K * rp_table_stride(M) / 4;
        for (int i = threadIdx.x; i < n4; i += BLK)
            dst[i] = src[i];
    }
    __syncthreads();
    const unsigned parity_bytes = (unsigned)K * rp_table_stride(M) * 4;

    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    unsigned long long region = (unsigned long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (region >= nregions)
        region = nregions - 1;
    unsigned long long o0 = row_begin + region * R;
    if (o0 + R > row_end)
        o0 = row_end - R;

    const unsigned shq = 8u * (col & 3);
    // half h reads input row o0 + h + t at step t
    const uint8_t *sp = QL ? seq + (o0 + h + (col & 3)) * 32 + (col >> 2) * 4 : seq + (o0 + h) * 32 + col;
    float *op = out + ((o0 - row_begin) + h) * 32 + col;

    constexpr int PFE = rp_pf_bytes(M, PF);
    float acc[H];
    unsigned sym[G];
#pragma unroll
    for (int u = 0; u < H; ++u)
        acc[u] = 0.0f;
#pragma unroll
    for (int i = 0; i < G; ++i)
        sym[i] = 0;
    if (QL) {
#pragma unroll
        for (int b = 0; b < rp_pf_blocks(M); ++b)
            sym[b] = *reinterpret_cast<const unsigned *>(sp + b * 128);
    } else {
#pragma unroll
        for (int i = 0; i < PFE; ++i)
            sym[i] = sp[i * 32];
    }
    float best_v = __builtin_nanf("");

    const unsigned long long q = R / G;  // groups that start outputs
    // QL, LAST group: this lane's row (+ql) of the region's last input row, relative to `sp`
    const int last_off = (rp_last_steps(M) - 1 - (col & 3)) * 32;

    rp_group_steps<M, MODE, PHASE_FIRST, QL, PFE>(acc, sym, sp, lds_raw, parity_bytes, op, shq, best_v, 0);
    for (unsigned long long g = 1; g < q; ++g) {
        sp += G * 32;
        op += H * 64;
        rp_group_steps<M, MODE, PHASE_MAIN, QL, PFE>(acc, sym, sp, lds_raw, parity_bytes, op, shq, best_v, 0);
    }
    sp += G * 32;
    op += H * 64;
    rp_group_steps<M, MODE, PHASE_LAST, QL, PFE>(acc, sym, sp, lds_raw, parity_bytes, op, shq, best_v, last_off);

    if (MODE == MODE_STORE_ARGMAX) {
        // one record per wavefront; "index" = the region, ties go to the later rows
        long long idx = best_v != best_v
                            ? -1
                            : (long long)((unsigned long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6));
        best_wave_reduce(best_v, idx);
        if ((threadIdx.x & 63) == 0) {
            ArgmaxRecord *rec = fo.block_best + (size_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
            rec->value = best_v;
            rec->index = idx;
            rec->found = idx >= 0;
        }
    }
}

template <int M, int MODE>
hipError_t score_c32_rp_launch(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq,
                               const float *table, int K, unsigned long long row_begin,
                               unsigned long long row_end, unsigned long long R,
                               unsigned long long nregions, float *out, FusedOut fo)
{
    hipLaunchKernelGGL((score_c32_rp<M, MODE>), grid, dim3(256), lds_bytes, stream, seq, table, K, row_begin,
                       row_end, R, nregions, out, fo);
    return hipGetLastError();
}

}  // namespace lm
