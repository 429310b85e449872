// score_prefilter_blk.hpp -- the one-symbol discrete prefilter scan of the PROTEIN alphabet (K = 21), round-6 form.
//
// Same arithmetic as score_prefilter.hpp (packed u16 rotating accumulators, the EVEN | ODD image of a matrix, candidates
// re-scored exactly by rescore_candidates: the hit set is bit-identical to the exact kernel's by construction); what
// changed is everything around the adds, which is where the round-5 kernel spent its time (0.095 ms per 200 Mres = 0.32
// of its LDS ceiling, DESIGN 4.9 -- 43 us of it in byte loads alone, tools/kbench/symload_bench `bytes`):
//
//   * symbols arrive in 4-row BLOCKS, one dword per lane, lane l of a half-wave on dword l of the block's 128 bytes (the
//     linear lane map of the pair scans: 173 instead of 213 us per Gpos for the loads alone).  The lane that holds row
//     q = l >> 3, columns 4b .. 4b + 3 (b = l & 7) of a block accumulates column 4b + q; the four lanes {b, 8 + b, 16 + b,
//     24 + b} transpose their 4 x 4 bytes with two exchanges -- DPP row_ror:8 (lane ^ 8) and v_permlane16_swap (lane ^ 16)
//     -- and two v_perm_b32: four operations per four steps, after which each lane holds its column's four symbols in
//     ONE register;
//   * the LDS address of a step is byte i of that register times the row size: one SDWA multiply.  The kernel has no
//     static LDS (lds_zero_based), so that product IS the address of the EVEN layout's row, and the ODD layout's rides in
//     the read's offset field;
//   * flags, not maxima (score_prefilter2.hpp: kFlagBits): the LDS copy of the image carries 0x8000 - td on the weights
//     of padded row 0, which every output receives exactly once, so "reached the threshold" is bit 15 of a completed half
//     and completed registers are OR-ed; the per-group record is GroupNotes (no compare, no 64-bit shift);
//   * motifs are padded to a multiple of 4 rows (prefilter_mp(m, wide)), so that blocks never straddle a group.
//
// Needs a 4-byte aligned sequence matrix (dword loads); other pointers take the byte-load scan of score_prefilter.hpp,
// which reads the same image.
#pragma once

#include "score_prefilter2.hpp"

namespace lm {

constexpr int kBlkKA = 21;  // the alphabet this kernel is built for (the ODD layout's offset is an immediate)

// (BlkTranspose, byte_times: score_kernels.hpp -- the store kernels' linear lane map shares them)

// The scan loop, software-pipelined by hand like the pair scans' (score_prefilter2.hpp: pair_items).  An ITEM is a pair of
// steps (2P, 2P + 1): the EVEN layout's row R0 of the first symbol and the ODD layout's row R1 of the second.  Accumulator
// register i receives dword m = (P - i) mod NP of BOTH rows, so
//     acc[i] = acc[i] + R0[m] + R1[m]                    one v_add3_u32, opaque to the optimiser
// except for the two registers a half of which completes inside the item:
//     m = 0       (register P):      its high half completes with R0[0] -> `fin0`, is cleared, then receives R1[0]
//     m = NP - 1  (register P + 1):  its low half completes with R1[NP - 1] -> `fin1`, then is cleared
// NP + 4 operations per item where one add per row and register takes 2 NP + 4.  The rows of the NEXT item are requested
// while the current ones are consumed, chunk by chunk (8 bytes of each row per chunk), scheduling barriers pinning that
// order: about 3 NP registers are live whatever the motif length.  Left to itself the compiler pairs terms of distant steps
// into v_add3_u32 and hoists the reads of several steps -- from M = 17 the rows in flight spill (M = 20: 55 scratch
// operations in the loop, 0.26 ms per 200 Mres where M = 16 took 0.07).
template <int NP>
struct BlkRows {
    unsigned r0[NP], r1[NP];
};

// chunk C (dwords 2C, 2C + 1) of both rows of an item: `row0` / `row1` = LDS addresses of the EVEN layout's row of the first
// symbol and of the ODD layout's row of the second (the ODD layout's offset rides in the read's offset field)
template <int M, int C>
__device__ __forceinline__ void blk_request_chunk(BlkRows<prefilter_mp(M, 1) / 2> &w, const unsigned row0, const unsigned row1)
{
    constexpr unsigned ODD = kBlkKA * prefilter_stride_dw(M, 1) * 4;  // byte offset of the ODD layout
    const lm_u32x2_t a = *(lm_lds_u64_ptr)(row0 + 8u * C);
    const lm_u32x2_t b = *(lm_lds_u64_ptr)(row1 + ODD + 8u * C);
    w.r0[2 * C + 0] = a.x;
    w.r0[2 * C + 1] = a.y;
    w.r1[2 * C + 0] = b.x;
    w.r1[2 * C + 1] = b.y;
}

template <int M, int P, int C>
__device__ __forceinline__ void blk_consume_chunk(unsigned (&acc2)[prefilter_mp(M, 1) / 2], const BlkRows<prefilter_mp(M, 1) / 2> &cur,
                                                  unsigned &fin0, unsigned &fin1)
{
    constexpr int NP = prefilter_mp(M, 1) / 2;
#pragma unroll
    for (int m = 2 * C; m < 2 * C + 2; ++m) {
        const int i = ((P - m) % NP + NP) % NP;
        if (m == 0) {
            const unsigned t = acc2[i] + cur.r0[0];
            fin0 = t;
            acc2[i] = (t & 0x0000ffffu) + cur.r1[0];
        } else if (m == NP - 1) {
            const unsigned t = add3_u32(acc2[i], cur.r0[m], cur.r1[m]);
            fin1 = t;
            acc2[i] = t & 0xffff0000u;
        } else {
            acc2[i] = add3_u32(acc2[i], cur.r0[m], cur.r1[m]);
        }
    }
}

template <int M, int P, int C>
__device__ __forceinline__ void blk_chunks(unsigned (&acc2)[prefilter_mp(M, 1) / 2], const BlkRows<prefilter_mp(M, 1) / 2> &cur,
                                           BlkRows<prefilter_mp(M, 1) / 2> &nxt, const unsigned row0, const unsigned row1, unsigned &fin0,
                                           unsigned &fin1, const bool has_next)
{
    if constexpr (2 * C < prefilter_mp(M, 1) / 2) {
        if (has_next)
            blk_request_chunk<M, C>(nxt, row0, row1);
        __builtin_amdgcn_sched_barrier(0);
        blk_consume_chunk<M, P, C>(acc2, cur, fin0, fin1);
        __builtin_amdgcn_sched_barrier(0);
        blk_chunks<M, P, C + 1>(acc2, cur, nxt, row0, row1, fin0, fin1, has_next);
    }
}

template <int M, int C>
__device__ __forceinline__ void blk_begin_rows(BlkRows<prefilter_mp(M, 1) / 2> &cur, const unsigned row0, const unsigned row1)
{
    if constexpr (2 * C < prefilter_mp(M, 1) / 2) {
        blk_request_chunk<M, C>(cur, row0, row1);
        blk_begin_rows<M, C + 1>(cur, row0, row1);
    }
}

// One group = NP items.  `blk` = the group's NB symbol blocks (requested one group ago); each is handed back to the loads --
// the same block of the NEXT group -- as soon as its symbols sit transposed in `sym4`.  On entry `cur` holds (or is about to
// receive) the rows of the group's item 0, `sym4` the symbols of its block 0 and blk[0] (the request of) block 0 of the next
// group; on exit the same for the next group, unless this was the stream's last.  `far`: see the kernel.
template <int M, int PHASE, int P>
__device__ __forceinline__ void blk_items(unsigned (&acc2)[prefilter_mp(M, 1) / 2], unsigned (&blk)[prefilter_mp(M, 1) / 4],
                                          BlkRows<prefilter_mp(M, 1) / 2> &cur, unsigned &sym4, const uint8_t *__restrict__ spq,
                                          unsigned &mx, const BlkTranspose &tr, const unsigned dsb, const unsigned far)
{
    constexpr int MP = prefilter_mp(M, 1);
    constexpr int NP = MP / 2;
    constexpr int NB = MP / 4;
    if constexpr (P < NP) {
        constexpr bool has_next = PHASE != PHASE_LAST || P + 1 < NP;
        constexpr int KN = (2 * P + 2) % MP;  // first step of the next item (step 0 of the next group after the last)
        unsigned row0 = 0, row1 = 0;
        if constexpr (has_next) {
            if constexpr (KN % 4 == 0) {
                // the next item opens a block: transposed now, and its register requests the same block one group on
                // (block 0 belongs to the NEXT group already and requests the one after it)
                sym4 = tr(blk[KN / 4]);
                if (PHASE != PHASE_LAST)
                    blk[KN / 4] = load_block(KN == 0 ? spq + far : spq + (NB + KN / 4) * 128);
            }
            row0 = byte_times<KN % 4>(sym4, dsb);
            row1 = byte_times<KN % 4 + 1>(sym4, dsb);
        }
        BlkRows<NP> nxt;
        unsigned fin0 = 0, fin1 = 0;
        blk_chunks<M, P, 0>(acc2, cur, nxt, row0, row1, fin0, fin1, has_next);
        if (PHASE != PHASE_FIRST)
            mx = or3_b32(mx, fin0, fin1);
        else if (P == NP - 1)  // the FIRST group completes the stream's output 0 only (its last step)
            mx = or_b32(mx, fin1);
        if constexpr (has_next) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                cur.r0[i] = nxt.r0[i];
                cur.r1[i] = nxt.r1[i];
            }
        }
        blk_items<M, PHASE, P + 1>(acc2, blk, cur, sym4, spq, mx, tr, dsb, far);
    }
}

// wavefronts per SIMD the register budget is cut for: 3 NP + NB + ~25 registers are live (accumulators, the rows of two
// steps, the block ring); a kernel with ANY scratch pays for it at every wavefront launch
constexpr int prefilter_blk_waves(int m)
{
    const int mp = prefilter_mp(m, 1);
    return mp <= 16 ? 8 : mp <= 20 ? 6 : mp <= 28 ? 5 : 4;
}

// Same stream geometry as score_c32_prefilter: T = q*MP + 1 outputs in q + 1 groups of MP steps.
template <int M>
__global__ __launch_bounds__(kBlock, prefilter_blk_waves(M)) void score_c32_prefilter_blk(
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
    (void)K;
    constexpr int MP = prefilter_mp(M, 1);
    constexpr int SHIFT = MP - M;
    constexpr int NP = MP / 2;
    constexpr int NB = MP / 4;
    constexpr int DSD = prefilter_stride_dw(M, 1);  // dwords per table row (2 * odd)
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    lds_zero_based(lds_raw);
    {
        // rows are whole 8-byte pieces; the first piece of a row holds the weights of padded row 0 -- low half in the
        // EVEN layout, high half in the ODD one (prefilter_pack_image) -- which take the bias
        uint2 *dst = reinterpret_cast<uint2 *>(lds_raw);
        const uint2 *src = reinterpret_cast<const uint2 *>(image);
        constexpr int n2 = prefilter_image_dw(M, kBlkKA) / 2, row2 = DSD / 2, even2 = kBlkKA * row2;
        const unsigned bias = td <= 0x8000u ? 0x8000u - td : 0u;  // td > 0x8000: out of reach of every sum, nothing is flagged
        for (int i = threadIdx.x; i < n2; i += kBlock) {
            uint2 v = src[i];
            if (i % row2 == 0)
                v.x += i < even2 ? bias : bias << 16;
            dst[i] = v;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int col = 4 * (lane & 7) + ((lane >> 3) & 3);  // the column this lane accumulates (see the file header)
    unsigned long long stream =
        ((unsigned long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool idle = stream >= nstreams;  // re-does the last stream, reports nothing
    if (idle)
        stream = nstreams - 1;
    unsigned long long o0 = row_begin + stream * T;
    if (o0 + T > row_end)
        o0 = row_end - T;

    // padded output l covers input rows (o0 - SHIFT) + l .. + MP - 1; the first SHIFT of them carry all-zero weight rows,
    // so rows before the matrix are never loaded (their "symbols" read as 0, a valid table row)
    const long long in0 = (long long)o0 - SHIFT;
    const int brow = (lane >> 3) & 3;                                       // this lane's row of a block
    const uint8_t *spq = seq + (in0 + brow) * 32 + (lane & 7) * 4;
    unsigned acc2[NP];
    unsigned blk[NB];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        acc2[i] = 0;
    blk[0] = 0;
    if (in0 + (long long)brow >= 0)
        blk[0] = load_block(spq);
#pragma unroll
    for (int j = 1; j < NB; ++j)
        blk[j] = load_block(spq + j * 128);

    const unsigned ngroups = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)T - 1u) / (unsigned)MP + 1u));  // exact: T = q*MP + 1
    const unsigned G = (ngroups + 63u) / 64u;  // groups per note (= per bit of hit_groups)
    unsigned gleft = G, nnotes = 0;
    unsigned mx = 0;
    GroupNotes notes;
    auto end_group = [&]() {
        if (--gleft == 0) {
            gleft = G;
            ++nnotes;
            notes.note(mx);
        }
    };
    const BlkTranspose tr;
    const unsigned dsb = DSD * 4u;
    // lead-in: block 0 transposed, the rows of item 0 requested, block 0 of group 1 requested in its place
    BlkRows<NP> cur;
    unsigned sym4 = tr(blk[0]);
    blk[0] = load_block(spq + NB * 128);
    blk_begin_rows<M, 0>(cur, byte_times<0>(sym4, dsb), byte_times<1>(sym4, dsb));
    // `far`: offset of block 0 of the group after next, which the last item of a group requests -- of the NEXT group's where
    // the stream has no such group (a re-read of a block that surely exists instead of 128 bytes past the stream's rows)
    constexpr unsigned FAR = 2u * NB * 128u, NEAR = NB * 128u;
    blk_items<M, PHASE_FIRST, 0>(acc2, blk, cur, sym4, spq, mx, tr, dsb, ngroups > 2 ? FAR : NEAR);
    end_group();
    for (unsigned g = 1; g + 1 < ngroups; ++g) {
        spq += MP * 32;
        blk_items<M, PHASE_MAIN, 0>(acc2, blk, cur, sym4, spq, mx, tr, dsb, g + 2 < ngroups ? FAR : NEAR);
        end_group();
    }
    spq += MP * 32;
    blk_items<M, PHASE_LAST, 0>(acc2, blk, cur, sym4, spq, mx, tr, dsb, 0u);
    end_group();
    if (gleft != G) {  // the last, partly filled note
        ++nnotes;
        notes.note(mx);
    }
    unsigned long long hit_groups = notes.finish(nnotes);

    // flagged groups -> candidate row ranges (group 0: output 0; group g >= 1: outputs (g-1)*MP + 1 .. g*MP, counted from
    // the stream's first output row)
    const long long first_row = (long long)(o0 - row_begin);
    const long long own_row = (long long)(stream * T);
    if (idle)
        hit_groups = 0;
    __syncthreads();  // the table is done with: its first bytes become emit_candidates' scratch
    emit_candidates<true>(hit_groups, col, fo, [=](int bit, long long &r0, long long &r1) {
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
    }, lds_raw);
}

template <int M>
hipError_t score_c32_prefilter_blk_launch(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq, const unsigned *image,
                                          int K, unsigned long long row_begin, unsigned long long row_end, unsigned long long T,
                                          unsigned long long nstreams, unsigned td, FusedOut fo)
{
    hipLaunchKernelGGL((score_c32_prefilter_blk<M>), grid, dim3(kBlock), lds_bytes, stream, seq, image, K, row_begin, row_end, T,
                       nstreams, td, fo);
    return hipGetLastError();
}

}  // namespace lm
