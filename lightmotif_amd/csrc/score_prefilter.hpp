// score_prefilter.hpp -- fused score+threshold with a packed 16-bit discrete prefilter.
//
// The reference's Scanner (lightmotif/src/scan.rs:169-198) does not score in f32
// first: it scores a *discretised*, over-estimating matrix (DiscreteMatrix,
// pwm/mod.rs:665-696: weights rounded UP, threshold rounded DOWN), keeps the
// positions whose discrete score reaches the discrete threshold, and re-scores only
// those exactly (`score_position`, scan.rs:187-190).  The hit set is therefore exactly
// `{i : f32 score(i) >= t}`, whatever the discretisation.  This kernel is the MI355X
// form of that idea:
//
//   * weights are u16 (the reference uses u8 for AVX2's byte shuffles; 16 bits keep
//     the false-positive rate negligible and cannot overflow: sum <= kPrefilterTop + 2 M);
//   * two accumulators share one VGPR and are advanced by ONE 32-bit add (sums cannot carry
//     across the halves), and a
//     symbol's whole column is M*2 bytes of LDS instead of M*4 -> half the LDS
//     traffic and half the adds of the f32 kernel, which is LDS-bound;
//   * flagged row ranges go to the candidate list and are re-scored with the exact
//     f32 weights by `rescore_candidates` (score_threshold.hip), the same path the f32 fused
//     kernel uses, so results are bit-identical to it.
//
// Soundness (no false negatives): with P' = P where finite and the row minimum where
// P = -inf, off_j = min_s P'[j][s], O = sum off_j, factor = (sum_j max_s P'[j][s] - O)
// / kPrefilterTop, d[j][s] = ceil((P'[j][s] - off_j) / factor), an f32 score S >= t implies
// sum d >= floor((t - O) / factor) - ceil(E / factor) where E bounds the rounding error
// of the M sequential f32 adds (host side, pssm.hip: build_prefilter).
//
// Rotating accumulators as in score_c32, with the motif padded to an EVEN length MP by
// a leading all-zero row (SHIFT = MP - M): slot pair i = outputs (2i, 2i+1) mod MP.
// At step k the pair needs the weights (d[j], d[j-1]) with j = (k - 2i) mod MP, which
// sit in one dword of one of two pre-arranged LDS layouts chosen by the parity of k:
//   layout EVEN (k even): dword m = (lo d[2m],   hi d[2m-1 mod MP])
//   layout ODD  (k odd) : dword m = (lo d[2m+1], hi d[2m])
#pragma once

#include "score_kernels.hpp"

namespace lm {

// The discrete weights of a matrix add up to at most kPrefilterTop + 2 M (every weight is rounded up, once for the
// ceiling and once to guard it): below 0x8000 for every supported length, so bit 15 of a biased sum can serve as its
// "reached the threshold" flag in the pair scans (score_prefilter2.hpp: kFlagBits).  15 bits of resolution over the
// matrix's score range: the over-estimate is at most 2 M quanta of range / 32000.
constexpr unsigned kPrefilterTop = 32000u;
// padded motif length: even (two accumulators per register); a multiple of 4 for wide alphabets, whose scan fetches symbols
// in 4-row blocks that must not straddle a group (score_prefilter_blk.hpp) -- every user of a wide image shares that geometry
constexpr int prefilter_mp(int m, int wide = 0) { return wide ? (m + 3) / 4 * 4 : (m + 1) / 2 * 2; }
// Symbol look-ahead of the byte-load scans, in steps: at most the ring of MP symbol registers minus the one being consumed.
// (Round 5 shortened it to MP - 2 because the protein kernels of M = 7, 8 lost a quarter of their candidates at MP - 1.  The ring
// was innocent: at MP - 1 those kernels used v0..v55 of 56 allocated VGPRs and v55 held the amount of the 64-bit shift that moved
// the "groups per flag bit" mask -- a shift by the LAST allocated VGPR returns garbage on gfx950 (LLVM's Shift64HighRegBug of
// gfx11; tools/kbench/shift64_repro.hip, profiles/r06_ring_fault_experiments.txt).  The scans keep their flags in GroupNotes now,
// which has no such shift, and tools/isa_audit.py checks every kernel of the built library for the pattern: DESIGN 4.9.)
constexpr int prefilter_lookahead(int pf, int mp) { return pf < mp - 1 ? pf : mp - 1; }
// dwords per symbol row of one discrete layout: 4 * odd >= MP / 2 (conflict-free b128)
// (wide alphabets, lds_wide(k): 2 * odd, read with single ds_read_b64 -- see table_stride in score_kernels.hpp)
constexpr int prefilter_stride_dw(int m, int wide = 0)
{
    return wide ? 2 * (((prefilter_mp(m, 1) / 2 + 1) / 2) | 1) : 4 * (((prefilter_mp(m) / 2 + 3) / 4) | 1);
}
// total dwords of the LDS image: layout EVEN | layout ODD
constexpr int prefilter_image_dw(int m, int k)
{
    return 2 * k * prefilter_stride_dw(m, lds_wide(k));
}

// Host side: packs the padded discrete weights d[j * k + s], j < prefilter_mp(m, lds_wide(k)) (leading all-zero
// padding rows), into the LDS image [layout EVEN | layout ODD].
inline void prefilter_pack_image(const unsigned *d, int m, int k, unsigned *image)
{
    const int mp = prefilter_mp(m, lds_wide(k)), dsd = prefilter_stride_dw(m, lds_wide(k));
    for (int i = 0; i < prefilter_image_dw(m, k); ++i)
        image[i] = 0u;
    unsigned *even = image;
    unsigned *odd = even + (size_t)k * dsd;
    for (int s = 0; s < k; ++s)
        for (int w = 0; w < mp / 2; ++w) {
            const unsigned e_lo = d[(size_t)(2 * w) * k + s];
            const unsigned e_hi = d[(size_t)((2 * w - 1 + mp) % mp) * k + s];
            const unsigned o_lo = d[(size_t)(2 * w + 1) * k + s];
            const unsigned o_hi = d[(size_t)(2 * w) * k + s];
            even[(size_t)s * dsd + w] = e_lo | (e_hi << 16);
            odd[(size_t)s * dsd + w] = o_lo | (o_hi << 16);
        }
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Adds two pairs of u16 accumulators.  No 16-bit sum can exceed kPrefilterTop + 2M < 32768 (host
// side, build_prefilter), so nothing ever carries from the low into the high half and a
// plain 32-bit add IS the packed add -- v_add_u32 issues at twice the rate of v_pk_add_u16
// on this part (tools/kbench/valu_bench: 64 vs 38 T lane-instr/s).
__device__ __forceinline__ unsigned pk_add_u16(unsigned a, unsigned b)
{
    return a + b;
}

__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b)
{
    const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, y));
}

__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b)
{
    const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(x, y));
}

// (GroupNotes, the per-stream record of flagged groups, lives in score_kernels.hpp)

// `mx` collects (packed max) every register that holds a just-completed sum.  Its other half
// is the partial sum of an output still in flight; weights are >= 0, so a partial sum that
// reaches td belongs to an output that will reach it too -- taking it into the maximum can
// only flag a group early (an extra exact re-scoring), never hide a hit.  One v_pk_max_u16
// per step instead of extract + compare + select.
//
// STORE = 1 (score_u8.hpp): the sums are a DiscreteMatrix's u8 scores; every completed sum is
// written to `op[k * 32]` (`op` = the lane's cell of the group's first completed output;
// the FIRST group completes one output only), clamped to 255 (`sat_mask` = 0: the
// saturating adds of avx2.rs:336) or reduced mod 256 (`sat_mask` = 0xff: Generic's `+=`).
template <int M, int PF, int PHASE, int STORE = 0, int WIDE = 0>
__device__ __forceinline__ void prefilter_group(unsigned (&acc2)[prefilter_mp(M, WIDE) / 2],
                                                unsigned (&sym)[prefilter_mp(M, WIDE)],
                                                const uint8_t *__restrict__ sp,
                                                const char *__restrict__ tab_even,
                                                const char *__restrict__ tab_odd,
                                                unsigned &mx, uint8_t *__restrict__ op = nullptr,
                                                const unsigned wrap_mask = 0)
{
    // STORE: quad-transposed stores (see below).  Lane q = lane & 3 of a quad writes row k0 + q of
    // the quad's columns: `oq` = that dword for k0 = 0; the byte selectors pick byte q of two
    // neighbours' packs into bytes (0, 1) resp. (2, 3) of the result, 0x0c = constant zero.
    const unsigned q = STORE ? (threadIdx.x & 3u) : 0u;
    uint8_t *oq = STORE ? op - q + q * 32 : nullptr;
    const unsigned sel_lo = 0x0c0c0000u | q | ((4u + q) << 8);
    const unsigned sel_hi = 0x00000c0cu | (q << 16) | ((4u + q) << 24);
    unsigned pack = 0;
    constexpr int MP = prefilter_mp(M, WIDE);
    constexpr int NP = MP / 2;
    constexpr int NV = (NP + 3) / 4;
    constexpr unsigned DSB = prefilter_stride_dw(M, WIDE) * 4;
#pragma unroll
    for (int k = 0; k < MP; ++k) {
        if (PF > 0) {
            if (PHASE != PHASE_LAST || k + PF < MP)
                sym[(k + PF) % MP] = sp[(k + PF) * 32];
        } else {
            sym[k] = sp[k * 32];
        }
        const char *row = static_cast<const char *>(__builtin_assume_aligned(
            ((k & 1) ? tab_odd : tab_even) + __umul24(sym[k], DSB), WIDE ? 8 : 16));
        // NP dwords: whole 16-byte reads, then an 8- and/or 4-byte read for the rest
        unsigned w2[NV * 4];
        if constexpr (WIDE != 0) {
#pragma unroll
            for (int q = 0; q < (NP + 1) / 2; ++q) {
                // (an unsigned vector: `__builtin_bit_cast(unsigned, v.y)` on a float vector's element reads v.x, hipcc 7.0)
                const lm_u32x2_t v = *(lm_lds_u64_ptr)(row + 8 * q);
                w2[2 * q + 0] = v.x;
                w2[2 * q + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) {
                const uint4 v = *reinterpret_cast<const uint4 *>(row + 16 * q);
                w2[4 * q + 0] = v.x;
                w2[4 * q + 1] = v.y;
                w2[4 * q + 2] = v.z;
                w2[4 * q + 3] = v.w;
            }
            if (NP % 4 >= 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(row + 16 * (NP / 4));
                w2[4 * (NP / 4) + 0] = v.x;
                w2[4 * (NP / 4) + 1] = v.y;
            }
            if (NP % 2 == 1)
                w2[NP - 1] = *reinterpret_cast<const unsigned *>(row + 4 * (NP - 1));
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int j_lo = (k - 2 * i + 2 * MP) % MP;  // weight row of slot 2i at this step
            acc2[i] = pk_add_u16(acc2[i], w2[j_lo / 2]);
        }
        // slot (k+1) mod MP received its last weight: compare, then clear it for the
        // output that starts in it at the next step
        const int sc = (k + 1) % MP;
        if (STORE) {
            if (PHASE != PHASE_FIRST || k == MP - 1) {
                const unsigned sum = (sc & 1) ? (acc2[sc / 2] >> 16) : (acc2[sc / 2] & 0xffffu);
                const unsigned v = wrap_mask ? (sum & wrap_mask) : (sum < 255u ? sum : 255u);
                constexpr int KQ = MP / 4 * 4;  // steps covered by whole 4-row blocks
                if (PHASE == PHASE_FIRST || k >= KQ) {
                    op[(PHASE == PHASE_FIRST ? 0 : k) * 32] = (uint8_t)v;
                } else {
                    // Four completed rows are packed per lane (byte t = row k0 + t of the lane's
                    // column) and transposed inside each quad of lanes: lane q ends up with row
                    // k0 + q of the quad's four columns = ONE dword store, and a half-wave
                    // instruction covers 4 rows x 32 columns = 128 contiguous bytes instead of 32.
                    pack = (k % 4 == 0) ? v : pack | (v << (8 * (k % 4)));
                    if (k % 4 == 3) {
                        const unsigned p0 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0x00, 0xf, 0xf, true);
                        const unsigned p1 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0x55, 0xf, 0xf, true);
                        const unsigned p2 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0xaa, 0xf, 0xf, true);
                        const unsigned p3 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0xff, 0xf, 0xf, true);
                        const unsigned row = __builtin_amdgcn_perm(p1, p0, sel_lo) | __builtin_amdgcn_perm(p3, p2, sel_hi);
                        *reinterpret_cast<unsigned *>(oq + (k - 3) * 32) = row;
                    }
                }
            }
        } else if (PHASE != PHASE_FIRST || k == MP - 1) {
            mx = pk_max_u16(mx, acc2[sc / 2]);
        }
        acc2[sc / 2] &= (sc & 1) ? 0x0000ffffu : 0xffff0000u;
    }
}

// Same geometry as score_c32<M, MODE_THRESHOLD> with M replaced by MP: every stream
// sweeps T = q*MP + 1 outputs in (q+1) groups of MP steps.  `image` = the LDS image
// described above; `td` = discrete threshold.
template <int M, int PF = kScorePF, int WIDE = 0>
__global__ __launch_bounds__(kBlock, 6) void score_c32_prefilter(
    const uint8_t *__restrict__ seq, const unsigned *__restrict__ image, const int K,
    const unsigned long long row_begin, const unsigned long long row_end,
    const unsigned long long T, const unsigned long long nstreams, unsigned td,
    const FusedOut fo_in)
{
    FusedOut fo = fo_in;
    if (fo_in.batch) {  // multi-job launch: this block's job (wave-uniform)
        const BatchParams bp = fo_in.batch[blockIdx.y];
        image = static_cast<const unsigned *>(bp.table);
        td = bp.td;
        fo.job_key = bp.job_key;
    }
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
    const bool idle = stream >= nstreams;  // re-does the last stream, reports nothing
    if (idle)
        stream = nstreams - 1;
    unsigned long long o0 = row_begin + stream * T;
    if (o0 + T > row_end)
        o0 = row_end - T;

    // padded output o' = o - SHIFT covers input rows o' .. o'+MP-1; its first row carries
    // the all-zero weight row, so when o0 == 0 that (non-existent) row is never loaded
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
    static_assert(PFE >= 1, "the prefilter kernel needs a look-ahead of at least one step");
#pragma unroll
    for (int j = 0; j < PFE; ++j) {
        if (j < SHIFT) {
            if (o0 + j >= (unsigned long long)SHIFT)  // rows before the matrix do not exist; their weight rows are all zero anyway
                sym[j] = sp[j * 32];
        } else {
            sym[j] = sp[j * 32];
        }
    }

    // wave-uniform 32-bit bookkeeping (T <= 2^30, score_plan.hip)
    const unsigned ngroups = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)T + MP - 1u) / (unsigned)MP));  // exact: T = q*MP + 1, >= 2
    const unsigned G = (ngroups + 63u) / 64u;  // groups per note (= per bit of hit_groups)
    unsigned gleft = G, nnotes = 0, pending = 0;
    unsigned mx = 0;
    GroupNotes notes;
    auto note_group = [&]() {
        const bool flag = (mx & 0xffffu) >= td || (mx >> 16) >= td;
        pending |= flag ? 0x80000000u : 0u;
        mx = 0;
        if (--gleft == 0) {
            gleft = G;
            ++nnotes;
            notes.push(pending);
            pending = 0;
        }
    };

    prefilter_group<M, PFE, PHASE_FIRST, 0, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx);
    note_group();
    for (unsigned g = 1; g + 1 < ngroups; ++g) {
        sp += MP * 32;
        prefilter_group<M, PFE, PHASE_MAIN, 0, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx);
        note_group();
    }
    sp += MP * 32;
    prefilter_group<M, PFE, PHASE_LAST, 0, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx);
    note_group();
    if (gleft != G) {  // the last, partly filled note
        ++nnotes;
        notes.push(pending);
    }
    unsigned long long hit_groups = notes.finish(nnotes);

    // the flagged groups become candidates for exact re-scoring (outputs are counted from the stream's
    // first TRUE output row; group 0 completes output 0, group g >= 1 outputs
    // (g-1)*MP+1 .. g*MP).  Every cell is reported once: the shifted last stream skips
    // the rows the stream before it owns, idle half-waves report nothing.
    const long long first_row = (long long)(o0 - row_begin);
    const long long own_row = (long long)(stream * T);
    if (idle)
        hit_groups = 0;
    emit_candidates(hit_groups, col, fo, [=](int bit, long long &r0, long long &r1) {
        const unsigned long long g0 = (unsigned long long)bit * G;
        unsigned long long g1 = g0 + G;
        if (g1 > ngroups)
            g1 = ngroups;
        const long long i0 = g0 == 0 ? 0 : (long long)((g0 - 1) * MP + 1);
        long long i1 = (long long)((g1 - 1) * MP + 1);
        if (i1 > (long long)T)
            i1 = (long long)T;
        r0 = first_row + i0;
        if (r0 < own_row)
            r0 = own_row;
        r1 = first_row + i1;
    });
}

using PrefilterLauncher = hipError_t (*)(dim3 grid, size_t lds_bytes, hipStream_t stream,
                                         const uint8_t *seq, const unsigned *image, int K,
                                         unsigned long long row_begin, unsigned long long row_end,
                                         unsigned long long T, unsigned long long nstreams,
                                         unsigned td, FusedOut fo);

template <int M, int WIDE = 0>
hipError_t score_c32_prefilter_launch(dim3 grid, size_t lds_bytes, hipStream_t stream,
                                      const uint8_t *seq, const unsigned *image, int K,
                                      unsigned long long row_begin, unsigned long long row_end,
                                      unsigned long long T, unsigned long long nstreams,
                                      unsigned td, FusedOut fo)
{
    hipLaunchKernelGGL((score_c32_prefilter<M, kScorePF, WIDE>), grid, dim3(kBlock), lds_bytes, stream, seq, image,
                       K, row_begin, row_end, T, nstreams, td, fo);
    return hipGetLastError();
}

// `blocks`: the scan on 4-row symbol blocks (score_prefilter_blk.hpp: K = 21, 4-byte aligned matrix)
PrefilterLauncher score_c32_prefilter_lookup(int M, bool wide = false, bool blocks = false);

}  // namespace lm
