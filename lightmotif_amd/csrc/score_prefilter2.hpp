// score_prefilter2.hpp -- DNA prefilter scan over PAIRS of symbols.
//
// The discrete prefilter (score_prefilter.hpp) only has to over-estimate, and its weights
// are integers: their sum does not depend on the order of the additions.  That frees the
// scan from the one-symbol-per-lookup shape of the exact kernel.  Two consecutive input
// rows (symbols a, b) are looked up TOGETHER: the LDS table row of the pair holds, for
// every output they touch, the sum of the two weights they contribute,
//
//     E[e][(a, b)] = d[e-1][a] + d[e][b]        e = 0 .. M'   (out-of-range terms are 0)
//
// so one row of (M'+1) u16 entries advances all in-flight outputs by two input rows: half
// the LDS bytes and half the adds per position of the one-symbol prefilter, and a quarter
// of the exact f32 kernel's.  The table has 25 live rows for DNA (K = 5), placed so that the 16
// pairs without N occupy rows 0..15 -- distinct 16-byte LDS slots, conflict-free reads.
//
// Geometry.  The motif is padded to a length M' = 3 (mod 4) by leading all-zero rows, so
// M'+1 = RING is a multiple of 4 (see the symbol loads below) and the two outputs completed by a pair of rows always
// share one accumulator dword: outputs 2t, 2t+1 live in dword t mod NPAIR (NPAIR = RING/2,
// low half = even output).  At super-step u (input rows 2u, 2u+1 of the stream) dword t
// receives table dword m = u - t = (lo E[2m+1], hi E[2m]); after m = NPAIR-1 both outputs
// are complete, the dword joins the packed running maximum and restarts at zero.  One LDS
// layout suffices (no even/odd step parity).  A stream sweeps T = q*RING + 2 outputs in
// q+1 groups of NPAIR super-steps; group 0 completes local outputs 0 and 1, group g >= 1
// the outputs (g-1)*RING + 2 .. g*RING + 1.
//
// Everything downstream (group bitmask -> candidates -> exact re-scoring) is shared with
// score_prefilter.hpp, so the hit lists are bit-identical to the exact kernel's.
#pragma once

#include "score_prefilter.hpp"

namespace lm {

constexpr int kPairPFB = 3;  // 4-row symbol blocks requested ahead of use in the pair scans

// padded length M' = 3 (mod 4).  A table row of NPAIR = (M' + 1) / 2 dwords is read as whole 16-byte pieces plus,
// when NPAIR % 4 == 2 (M' = 11, 19, 27, 35), an 8-byte tail.  Left to itself the compiler fuses those tails (of
// neighbouring steps, or of the motifs of a multi-motif pass) into ds_read2_b64 / ds_read_b96 -- 8 LDS cycles each and
// conflict-prone (MI355X_MICROARCH.md, LDS table) -- which is where the "2-way bank conflicts of the 8-byte reads"
// of round 2 came from (a third of the LDS cycles of multi<8..11, 4>, profiles/r02_c3_record.md; ISA of round 3:
// 22 ds_read2_b64 + 4 ds_read_b96 in multi<16, 2>).  Round 2 worked around it by padding 8 <= M <= 11 to M' = 15
// (two whole 16-byte reads, +33 % adds: the 2 346-motif batch 37.2 -> 33.8 ms).  Round 3 reads the tail through a
// volatile LDS pointer instead -- a lone ds_read_b64 costs 2 cycles, as the protein counters showed -- and drops the
// padding: the JASPAR batch 22.5 -> 20.6 ms (mean of six interleaved runs; with the padding kept: 22.0;
// profiles/r03_pair_tail_ab.txt).
constexpr int prefilter2_mo(int m) { return m | 3; }
constexpr int prefilter2_ring(int m) { return prefilter2_mo(m) + 1; }   // input rows per group
constexpr int prefilter2_npair(int m) { return prefilter2_ring(m) / 2; }
// dwords per table row: 4 * odd >= NPAIR (rows 0..15 then sit in distinct 16-byte slots)
constexpr int prefilter2_stride_dw(int m) { return 4 * (((prefilter2_npair(m) + 3) / 4) | 1); }
// table rows = symbol pairs.  Protein (K = 21): 441 rows, a * 21 + b.  Protein rows cannot be conflict-free
// (441 rows, 16 slots): tools/kbench/lds_rows_bench measures 5.5 ns per wavefront read against 3.5 ns for the 21
// rows of the one-symbol prefilter -- but the pair scan needs half the reads.
// DNA (K = 5): row(a, b) = 4 a + b', b' = b for A C T G and 20 for N -- ADDITIVE in the two symbols, so that a lane
// can look both terms up in registers (dna_pair_offsets below).  The 16 pairs without N sit in rows 0..15 = distinct
// 16-byte LDS slots (conflict-free reads); (N, b) in 16..19, (a, N) in 20, 24, 28, 32, (N, N) in 36; the other rows
// of the 37 are never read.
constexpr int prefilter2_rows(int ka) { return ka == 5 ? 37 : ka * ka; }
constexpr int prefilter2_image_dw(int m, int ka = 5) { return prefilter2_rows(ka) * prefilter2_stride_dw(m); }

__host__ __device__ __forceinline__ unsigned dna_pair_row(unsigned a, unsigned b)
{
    return 4u * a + (b == 4u ? 20u : b);
}
template <int KA>
__host__ __device__ __forceinline__ unsigned pair_row(unsigned a, unsigned b)
{
    return KA == 5 ? dna_pair_row(a, b) : a * (unsigned)KA + b;
}

// DNA decode in registers.  The lanes of a quad hold a 4 x 4 block of symbol bytes (lane q: row r + q, columns
// 4i .. 4i+3).  Picking one symbol apart costs a DPP move + a bit-field extract, and the pair's row another five
// operations -- together more than half of the scan's VALU work once the adds were halved (round 5 counters:
// 18.4 VALU per super-step, 10 of them decode, LDS array busy 66 %).  Instead the quad TRANSPOSES its block (two DPP
// moves + two v_perm_b32: every lane then holds the four rows of its OWN column in one dword), two more v_perm_b32
// use that dword as a selector into byte tables -- 4 so a / so b' in units of 16 bytes, `so` = 16-byte slots per
// table row -- and one shifted add leaves row(a, b) * so of the block's two pairs in bytes 0 and 2.  Eight
// operations per four symbols, and the byte select rides on the shift that makes the LDS address (SDWA).
// Needs 20 so + 16 so < 256: table rows of up to 7 slots (M' <= 55); longer motifs keep the one-by-one decode.
constexpr bool prefilter2_lut_decode(int m, int ka) { return ka == 5 && prefilter2_stride_dw(m) / 4 <= 7; }

struct QuadTranspose {  // per-lane selectors of the two v_perm_b32 steps (set once per kernel)
    unsigned sel1, sel2;
    unsigned four;  // the shift count of byte_times_16 (an SDWA operand must be a register)
};
__device__ __forceinline__ QuadTranspose quad_transpose_setup()
{
    const unsigned lane = threadIdx.x;
    QuadTranspose qt;
    qt.sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    qt.sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    qt.four = 4u;
    return qt;
}
// lane q of a quad: dword = (row r + q; columns c0 .. c0 + 3)  ->  (rows r .. r + 3; column c0 + q)
__device__ __forceinline__ unsigned quad_transpose(unsigned d, const QuadTranspose &qt)
{
    const unsigned x = (unsigned)__builtin_amdgcn_mov_dpp((int)d, 0xb1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]
    const unsigned p = __builtin_amdgcn_perm(x, d, qt.sel1);
    const unsigned y = (unsigned)__builtin_amdgcn_mov_dpp((int)p, 0x4e, 0xf, 0xf, true);  // quad_perm [2, 3, 0, 1]
    return __builtin_amdgcn_perm(y, p, qt.sel2);
}
// bytes 0 / 2 = row(a, b) * SO of the pairs (rows 0, 1) / (rows 2, 3) of a transposed block; SO = 16-byte slots per row
template <int SO>
__device__ __forceinline__ unsigned dna_pair_offsets(unsigned t)
{
    constexpr unsigned A_LO = 0u | (4u * SO << 8) | (8u * SO << 16) | (12u * SO << 24), A_HI = 16u * SO;
    constexpr unsigned B_LO = 0u | (1u * SO << 8) | (2u * SO << 16) | (3u * SO << 24), B_HI = 20u * SO;
    const unsigned qa = __builtin_amdgcn_perm(A_HI, A_LO, t);  // byte k = 4 SO * symbol(row k)
    const unsigned pb = __builtin_amdgcn_perm(B_HI, B_LO, t);  // byte k = SO * b'(row k)
    return qa + (pb >> 8);  // no byte carries: every sum is below 256
}
// ((s >> 8 * BYTE) & 0xff) << 4 in ONE operation: the byte select rides on the shift (SDWA; hipcc emits and + shift)
template <int BYTE>
__device__ __forceinline__ unsigned byte_times_16(unsigned s, unsigned four)
{
    unsigned r;
    if constexpr (BYTE == 0)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(four), "v"(s));
    else
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(four), "v"(s));
    return r;
}

// Host side: the pair table from the unpadded discrete weights d[j * ka + s], j < m.
inline void prefilter2_pack_image(const unsigned *d, int m, unsigned *image2, int ka = 5)
{
    const int mo = prefilter2_mo(m), shift2 = mo - m, np2 = prefilter2_npair(m);
    const int dsd2 = prefilter2_stride_dw(m);
    auto dq = [&](int j, int s) -> unsigned {  // padded weight, 0 outside shift2 .. mo-1
        return (j < shift2 || j >= mo) ? 0u : d[(size_t)(j - shift2) * ka + s];
    };
    for (int i = 0; i < prefilter2_image_dw(m, ka); ++i)
        image2[i] = 0u;
    for (int a = 0; a < ka; ++a)
        for (int b = 0; b < ka; ++b) {
            unsigned *row = image2 + (size_t)(ka == 5 ? dna_pair_row((unsigned)a, (unsigned)b) : (unsigned)(a * ka + b)) * dsd2;
            auto entry = [&](int e) -> unsigned { return e > mo ? 0u : dq(e - 1, a) + dq(e, b); };
            for (int w = 0; w < np2; ++w)
                row[w] = entry(2 * w + 1) | (entry(2 * w) << 16);
        }
}

// Symbol loads.  A byte load per lane and row moves 64 bytes per wavefront instruction, and
// at > 3 Tpos/s the scan is bound by that request rate, not by LDS or VALU.  So the lanes
// of a quad (columns 4i..4i+3) fetch a 4 x 4 block of symbols with ONE dword load each --
// lane q reads row r+q, columns 4i..4i+3; a half-wave instruction covers 4 rows = 128
// contiguous bytes -- and every lane picks its column out of its neighbours' registers:
// symbol(row r+t, own column) = byte (lane & 3) of the dword held by quad lane t (one DPP
// quad broadcast + one bit-field extract).  RING is a multiple of 4, so blocks never
// straddle a group; `blk` is a ring of the RING/4 blocks of a group, prefetched PFB ahead.
//
// STORE = 1 (score_u8.hpp): the sums are a DiscreteMatrix's u8 scores.  A super-step completes
// two rows of the lane's column; two super-steps make a 4-row block that is transposed inside
// the quad of lanes and written with one dword store per lane (128 contiguous bytes per
// half-wave instruction).  `op` = the lane's cell of the group's first completed row; the
// FIRST group completes rows 0 and 1 only (byte stores).  `wrap_mask` as in prefilter_group.
template <int M, int PFB, int PHASE, int STORE = 0, int KA = 5>
__device__ __forceinline__ void prefilter2_group(unsigned (&acc)[prefilter2_npair(M)],
                                                 unsigned (&blk)[prefilter2_ring(M) / 4],
                                                 const uint8_t *__restrict__ spq, const unsigned shq,
                                                 const char *__restrict__ tab, unsigned &mx,
                                                 const QuadTranspose &qt,
                                                 uint8_t *__restrict__ op = nullptr,
                                                 const unsigned wrap_mask = 0)
{
    constexpr bool LUT = prefilter2_lut_decode(M, KA);
    unsigned pair_off = 0;  // LUT: dna_pair_offsets of the current block
    const unsigned q = STORE ? (threadIdx.x & 3u) : 0u;
    uint8_t *oq = STORE ? op - q + q * 32 : nullptr;  // lane q of a quad writes row +q, the quad's 4 columns
    const unsigned sel_lo = 0x0c0c0000u | q | ((4u + q) << 8);
    const unsigned sel_hi = 0x00000c0cu | (q << 16) | ((4u + q) << 24);
    unsigned pack = 0;
    constexpr int RING = prefilter2_ring(M);
    constexpr int NB = RING / 4;
    constexpr int NP = prefilter2_npair(M);
    constexpr int NV = (NP + 3) / 4;
    constexpr unsigned DSB = prefilter2_stride_dw(M) * 4;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        // this super-step's two symbols: rows 2k, 2k+1 of the group = block k/2, rows (2k)%4, +1
        const unsigned d = blk[k / 2];
        unsigned row_off;
        if constexpr (LUT) {
            if ((k & 1) == 0)
                pair_off = dna_pair_offsets<(int)(DSB / 16)>(quad_transpose(d, qt));
            row_off = (k & 1) ? byte_times_16<2>(pair_off, qt.four) : byte_times_16<0>(pair_off, qt.four);
        } else {
            const unsigned a = (k & 1) ? quad_symbol<2>(d, shq) : quad_symbol<0>(d, shq);
            const unsigned b = (k & 1) ? quad_symbol<3>(d, shq) : quad_symbol<1>(d, shq);
            row_off = __umul24(pair_row<KA>(a, b), DSB);
        }
        // a block is free once its second pair is taken: request the block PFB ahead
        if ((k & 1) && (PHASE != PHASE_LAST || k / 2 + PFB < NB))
            blk[(k / 2 + PFB) % NB] = *reinterpret_cast<const unsigned *>(spq + (k / 2 + PFB) * 128);
        const char *row = static_cast<const char *>(__builtin_assume_aligned(tab + row_off, 16));
        unsigned w[NV * 4];
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
            const uint4 v = *reinterpret_cast<const uint4 *>(row + 16 * q);
            w[4 * q + 0] = v.x;
            w[4 * q + 1] = v.y;
            w[4 * q + 2] = v.z;
            w[4 * q + 3] = v.w;
        }
        if (NP % 4 >= 2) {
            const lm_u32x2_t v = *(lm_lds_u64_ptr)(row + 16 * (NP / 4));
            w[4 * (NP / 4) + 0] = v.x;
            w[4 * (NP / 4) + 1] = v.y;
        }
        if (NP % 2 == 1)
            w[NP - 1] = *reinterpret_cast<const unsigned *>(row + 4 * (NP - 1));
#pragma unroll
        for (int m = 0; m < NP; ++m)
            acc[(k - m + NP) % NP] = pk_add_u16(acc[(k - m + NP) % NP], w[m]);
        // dword (k+1) mod NP received its last entry: both of its outputs are complete
        const int c = (k + 1) % NP;
        if (STORE) {
            if (PHASE != PHASE_FIRST || k == NP - 1) {
                // (lo, hi) = (even row, odd row): clamp or mask both halves at once
                const unsigned r = wrap_mask ? (acc[c] & 0x00ff00ffu) : pk_min_u16(acc[c], 0x00ff00ffu);
                if (PHASE == PHASE_FIRST) {
                    op[0] = (uint8_t)(r & 0xffu);
                    op[32] = (uint8_t)(r >> 16);
                } else if ((k & 1) == 0) {
                    pack = __builtin_amdgcn_perm(r, r, 0x0c0c0200u);            // bytes 0, 1 = rows 2k, 2k+1
                } else {
                    pack |= __builtin_amdgcn_perm(r, r, 0x02000c0cu);           // bytes 2, 3
                    const unsigned p0 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0x00, 0xf, 0xf, true);
                    const unsigned p1 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0x55, 0xf, 0xf, true);
                    const unsigned p2 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0xaa, 0xf, 0xf, true);
                    const unsigned p3 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0xff, 0xf, 0xf, true);
                    const unsigned row = __builtin_amdgcn_perm(p1, p0, sel_lo) | __builtin_amdgcn_perm(p3, p2, sel_hi);
                    *reinterpret_cast<unsigned *>(oq + 2 * (k - 1) * 32) = row;
                }
            }
        } else if (PHASE != PHASE_FIRST || k == NP - 1) {
            mx = pk_max_u16(mx, acc[c]);
        }
        acc[c] = 0;
    }
}

template <int M, int KA = 5>
__global__ __launch_bounds__(kBlock, (KA == 5 && M <= kMaxFastM) ? 6 : M <= 52 ? 4 : M <= 80 ? 3 : 2) void score_c32_prefilter2(
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
    constexpr int MO = prefilter2_mo(M);
    constexpr int SHIFT = MO - M;
    constexpr int RING = prefilter2_ring(M);
    constexpr int NP = prefilter2_npair(M);
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        const uint4 *src = reinterpret_cast<const uint4 *>(image);
        constexpr int n4 = prefilter2_image_dw(M, KA) / 4;
        for (int i = threadIdx.x; i < n4; i += kBlock)
            dst[i] = src[i];
    }
    __syncthreads();

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

    // padded output l covers input rows (o0 - SHIFT) + l .. + MO - 1; the first SHIFT of them
    // carry all-zero weight rows, so rows before the matrix are never loaded (their
    // "symbols" read as 0, a valid table row)
    const long long in0 = (long long)o0 - SHIFT;               // first input row of the stream
    const unsigned shq = 8u * (col & 3);
    const uint8_t *spq = seq + (in0 + (col & 3)) * 32 + (col >> 2) * 4;  // this lane's row of a block
    constexpr int NB = RING / 4;
    constexpr int PFB = NB > kPairPFB ? kPairPFB : NB;                       // blocks requested ahead of use (<= NB:
                                                               // a request reuses a slot only after its last read)
    unsigned acc[NP];
    unsigned blk[NB];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        acc[i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
        blk[j] = 0;
    // prologue: block 0 (rows in0 .. in0+3; a lane whose row lies before the matrix skips
    // it) and the next PFB - 1 blocks
    if (in0 + (long long)(col & 3) >= 0)
        blk[0] = *reinterpret_cast<const unsigned *>(spq);
#pragma unroll
    for (int j = 1; j < PFB; ++j)
        if (4 * j >= SHIFT || in0 + 4 * j + (long long)(col & 3) >= 0)  // SHIFT > 3: block 1 may start before the matrix too
            blk[j] = *reinterpret_cast<const unsigned *>(spq + j * 128);

    const unsigned long long ngroups = (T - 2) / RING + 1;  // exact: T = q*RING + 2
    unsigned long long hit_groups = 0;
    const unsigned long long G = (ngroups + 63) / 64;  // groups per bit
    unsigned long long gbit = 1, gleft = G;
    unsigned mx = 0;
    auto note_group = [&]() {
        const bool flag = (mx & 0xffffu) >= td || (mx >> 16) >= td;
        hit_groups |= flag ? gbit : 0ull;
        mx = 0;
        if (--gleft == 0) {
            gleft = G;
            gbit <<= 1;
        }
    };

    const QuadTranspose qt = quad_transpose_setup();
    prefilter2_group<M, PFB, PHASE_FIRST, 0, KA>(acc, blk, spq, shq, lds_raw, mx, qt);
    note_group();
    for (unsigned long long g = 1; g + 1 < ngroups; ++g) {
        spq += RING * 32;
        prefilter2_group<M, PFB, PHASE_MAIN, 0, KA>(acc, blk, spq, shq, lds_raw, mx, qt);
        note_group();
    }
    if (ngroups > 1) {
        spq += RING * 32;
        prefilter2_group<M, PFB, PHASE_LAST, 0, KA>(acc, blk, spq, shq, lds_raw, mx, qt);
        note_group();
    }

    // flagged groups -> candidate row ranges (group 0: outputs 0, 1; group g >= 1: outputs
    // (g-1)*RING + 2 .. g*RING + 1, counted from the stream's first output row)
    const long long first_row = (long long)(o0 - row_begin);
    const long long own_row = (long long)(stream * T);
    if (idle)
        hit_groups = 0;
    emit_candidates(hit_groups, col, fo, [=](int bit, long long &r0, long long &r1) {
        const unsigned long long g0 = (unsigned long long)bit * G;
        unsigned long long g1 = g0 + G;
        if (g1 > ngroups)
            g1 = ngroups;
        const long long i0 = g0 == 0 ? 0 : (long long)((g0 - 1) * RING + 2);
        long long i1 = (long long)((g1 - 1) * RING + 2);
        if (i1 > (long long)T)
            i1 = (long long)T;
        r0 = first_row + i0;
        if (r0 < own_row)
            r0 = own_row;
        r1 = first_row + i1;
    });
}

// ---- several motifs of one length per pass ------------------------------------------------------
//
// On many-motif batches the sequence is cache-resident and the scan is bound by instruction
// issue; about half of a super-step's instructions fetch and decode the two symbols and
// compute the table row -- work that does not depend on the motif.  This kernel advances NM
// motifs of the same length per pass: one decode, NM table reads / adds / maxima.  Launched
// with grid.y = ceil(jobs / NM) on the `batch` array (padded to a multiple of NM with entries
// whose td = 0xffffffff flags nothing).
constexpr int prefilter2_multi(int m)  // motifs per pass: bounded by the accumulator registers
{
    return prefilter2_npair(m) <= 8 ? 4 : prefilter2_npair(m) <= 16 ? 2 : 1;
}

template <int M, int NM, int PFB, int PHASE>
__device__ __forceinline__ void prefilter2_group_multi(unsigned (&acc)[NM][prefilter2_npair(M)],
                                                       unsigned (&blk)[prefilter2_ring(M) / 4],
                                                       const uint8_t *__restrict__ spq, const unsigned shq,
                                                       const char *__restrict__ tab, unsigned (&mx)[NM],
                                                       const QuadTranspose &qt)
{
    constexpr bool LUT = prefilter2_lut_decode(M, 5);
    unsigned pair_off = 0;
    constexpr int RING = prefilter2_ring(M);
    constexpr int NB = RING / 4;
    constexpr int NP = prefilter2_npair(M);
    constexpr int NV = (NP + 3) / 4;
    constexpr unsigned DSB = prefilter2_stride_dw(M) * 4;
    constexpr unsigned IMG = prefilter2_image_dw(M) * 4;  // bytes per motif's table
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const unsigned d = blk[k / 2];
        unsigned row_off;
        if constexpr (LUT) {
            if ((k & 1) == 0)
                pair_off = dna_pair_offsets<(int)(DSB / 16)>(quad_transpose(d, qt));
            row_off = (k & 1) ? byte_times_16<2>(pair_off, qt.four) : byte_times_16<0>(pair_off, qt.four);
        } else {
            const unsigned a = (k & 1) ? quad_symbol<2>(d, shq) : quad_symbol<0>(d, shq);
            const unsigned b = (k & 1) ? quad_symbol<3>(d, shq) : quad_symbol<1>(d, shq);
            row_off = __umul24(dna_pair_row(a, b), DSB);
        }
        if ((k & 1) && (PHASE != PHASE_LAST || k / 2 + PFB < NB))
            blk[(k / 2 + PFB) % NB] = *reinterpret_cast<const unsigned *>(spq + (k / 2 + PFB) * 128);
        const char *row0 = tab + row_off;
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) {
            const char *row = static_cast<const char *>(__builtin_assume_aligned(row0 + mi * IMG, 16));
            unsigned w[NV * 4];
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) {
                const uint4 v = *reinterpret_cast<const uint4 *>(row + 16 * q);
                w[4 * q + 0] = v.x;
                w[4 * q + 1] = v.y;
                w[4 * q + 2] = v.z;
                w[4 * q + 3] = v.w;
            }
            if (NP % 4 >= 2) {
                const lm_u32x2_t v = *(lm_lds_u64_ptr)(row + 16 * (NP / 4));
                w[4 * (NP / 4) + 0] = v.x;
                w[4 * (NP / 4) + 1] = v.y;
            }
            if (NP % 2 == 1)
                w[NP - 1] = *reinterpret_cast<const unsigned *>(row + 4 * (NP - 1));
#pragma unroll
            for (int m = 0; m < NP; ++m)
                acc[mi][(k - m + NP) % NP] = pk_add_u16(acc[mi][(k - m + NP) % NP], w[m]);
            const int c = (k + 1) % NP;
            if (PHASE != PHASE_FIRST || k == NP - 1)
                mx[mi] = pk_max_u16(mx[mi], acc[mi][c]);
            acc[mi][c] = 0;
        }
    }
}

// (register budget: at least 4 workgroups per CU; 3 and 5 measure the same, 6 spills: 32 vs 18 ms on the JASPAR argmax batch)
template <int M, int NM>
__global__ __launch_bounds__(kBlock, 4) void score_c32_prefilter2_multi(
    const uint8_t *__restrict__ seq, const unsigned long long row_begin, const unsigned long long row_end,
    const unsigned long long T, const unsigned long long nstreams, const FusedOut fo_in)
{
    static_assert(prefilter2_image_dw(M) % 4 == 0, "tables are copied 16 bytes at a time");
    constexpr int MO = prefilter2_mo(M);
    constexpr int SHIFT = MO - M;
    constexpr int RING = prefilter2_ring(M);
    constexpr int NP = prefilter2_npair(M);
    constexpr int IMG_DW = prefilter2_image_dw(M);
    const BatchParams *bps = fo_in.batch + (size_t)blockIdx.y * NM;  // this workgroup's NM jobs
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw) + mi * (IMG_DW / 4);
        const uint4 *src = static_cast<const uint4 *>(bps[mi].table);
        for (int i = threadIdx.x; i < IMG_DW / 4; i += kBlock)
            dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int col = lane & 31;
    unsigned long long stream =
        ((unsigned long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool idle = stream >= nstreams;
    if (idle)
        stream = nstreams - 1;
    unsigned long long o0 = row_begin + stream * T;
    if (o0 + T > row_end)
        o0 = row_end - T;
    const long long in0 = (long long)o0 - SHIFT;
    const unsigned shq = 8u * (col & 3);
    const uint8_t *spq = seq + (in0 + (col & 3)) * 32 + (col >> 2) * 4;
    constexpr int NB = RING / 4;
    constexpr int PFB = NB > kPairPFB ? kPairPFB : NB;
    unsigned acc[NM][NP];
    unsigned blk[NB];
#pragma unroll
    for (int mi = 0; mi < NM; ++mi)
#pragma unroll
        for (int i = 0; i < NP; ++i)
            acc[mi][i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
        blk[j] = 0;
    if (in0 + (long long)(col & 3) >= 0)
        blk[0] = *reinterpret_cast<const unsigned *>(spq);
#pragma unroll
    for (int j = 1; j < PFB; ++j)
        if (4 * j >= SHIFT || in0 + 4 * j + (long long)(col & 3) >= 0)  // SHIFT > 3: block 1 may start before the matrix too
            blk[j] = *reinterpret_cast<const unsigned *>(spq + j * 128);

    const unsigned long long ngroups = (T - 2) / RING + 1;
    const unsigned long long G = (ngroups + 63) / 64;
    unsigned long long gbit = 1, gleft = G;
    unsigned long long hit_groups[NM];
    unsigned mx[NM], td[NM];
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
        hit_groups[mi] = 0;
        mx[mi] = 0;
        td[mi] = bps[mi].td;
    }
    auto note_group = [&]() {
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) {
            const bool flag = (mx[mi] & 0xffffu) >= td[mi] || (mx[mi] >> 16) >= td[mi];
            hit_groups[mi] |= flag ? gbit : 0ull;
            mx[mi] = 0;
        }
        if (--gleft == 0) {
            gleft = G;
            gbit <<= 1;
        }
    };
    const QuadTranspose qt = quad_transpose_setup();
    prefilter2_group_multi<M, NM, PFB, PHASE_FIRST>(acc, blk, spq, shq, lds_raw, mx, qt);
    note_group();
    for (unsigned long long g = 1; g + 1 < ngroups; ++g) {
        spq += RING * 32;
        prefilter2_group_multi<M, NM, PFB, PHASE_MAIN>(acc, blk, spq, shq, lds_raw, mx, qt);
        note_group();
    }
    if (ngroups > 1) {
        spq += RING * 32;
        prefilter2_group_multi<M, NM, PFB, PHASE_LAST>(acc, blk, spq, shq, lds_raw, mx, qt);
        note_group();
    }

    const long long first_row = (long long)(o0 - row_begin);
    const long long own_row = (long long)(stream * T);
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
        FusedOut fo = fo_in;
        fo.job_key = bps[mi].job_key;
        emit_candidates(idle ? 0ull : hit_groups[mi], col, fo, [=](int bit, long long &r0, long long &r1) {
            const unsigned long long g0 = (unsigned long long)bit * G;
            unsigned long long g1 = g0 + G;
            if (g1 > ngroups)
                g1 = ngroups;
            const long long i0 = g0 == 0 ? 0 : (long long)((g0 - 1) * RING + 2);
            long long i1 = (long long)((g1 - 1) * RING + 2);
            if (i1 > (long long)T)
                i1 = (long long)T;
            r0 = first_row + i0;
            if (r0 < own_row)
                r0 = own_row;
            r1 = first_row + i1;
        });
        __syncthreads();  // emit_candidates' shared scratch is reused by the next motif
    }
}

// grid.y = groups of NM jobs; `fo.batch` = the (padded) job table of the launch
using PrefilterMultiLauncher = hipError_t (*)(dim3 grid, hipStream_t stream, const uint8_t *seq,
                                              unsigned long long row_begin, unsigned long long row_end,
                                              unsigned long long T, unsigned long long nstreams, FusedOut fo);

template <int M>
hipError_t score_c32_prefilter2_multi_launch(dim3 grid, hipStream_t stream, const uint8_t *seq,
                                             unsigned long long row_begin, unsigned long long row_end,
                                             unsigned long long T, unsigned long long nstreams, FusedOut fo)
{
    constexpr int NM = prefilter2_multi(M);
    hipLaunchKernelGGL((score_c32_prefilter2_multi<M, NM>), grid, dim3(kBlock),
                       (size_t)NM * prefilter2_image_dw(M) * 4, stream, seq, row_begin, row_end, T, nstreams, fo);
    return hipGetLastError();
}

template <int M, int KA = 5>
hipError_t score_c32_prefilter2_launch(dim3 grid, size_t lds_bytes, hipStream_t stream,
                                       const uint8_t *seq, const unsigned *image, int K,
                                       unsigned long long row_begin, unsigned long long row_end,
                                       unsigned long long T, unsigned long long nstreams,
                                       unsigned td, FusedOut fo)
{
    hipLaunchKernelGGL((score_c32_prefilter2<M, KA>), grid, dim3(kBlock), lds_bytes, stream, seq, image,
                       K, row_begin, row_end, T, nstreams, td, fo);
    return hipGetLastError();
}

PrefilterLauncher score_c32_prefilter2_lookup(int M, int K = 5);

}  // namespace lm
