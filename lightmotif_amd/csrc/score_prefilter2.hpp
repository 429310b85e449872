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

// 4-row symbol blocks requested ahead of their decode in the pair scans, at most the ring of a group (RING / 4 registers,
// held anyway).  Three were enough while the scan was VALU-bound; at 0.24 ms per Gbp the single-motif scan reads 4.2 TB/s
// with ~6 000 resident wavefronts, and 3 x 256 B in flight per wavefront (4.7 MB) is less than bandwidth x latency
// (round 5: VALU 59 %, LDS 69 % busy after the hand-pipelining -- neither pipe saturated: the loads themselves were the
// bound, in the lane order they were issued in; see LIN below.  With the linear order: VALU 74 %, LDS 75 %).
constexpr int kPairPFB = 8;

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
// Two layouts of a DNA pair table (see prefilter2_stride_dw).  Which one a kernel reads travels in its alphabet argument:
// KA = 5 -- 16-byte slots; KA = kDnaSlot8 -- 8-byte slots.  The multi-motif passes of a batch (LDS-bound, the layout
// whose speed does not depend on the input) take the second unless built with -DLM_PAIR_SLOT8=0 (A/B builds); the
// single-motif scans and the u8 pair store, which are bound by their HBM traffic up to M = 20, keep the first.
#ifndef LM_PAIR_SLOT8
#define LM_PAIR_SLOT8 1
#endif
constexpr int kDnaSlot8 = 5 | 256;
constexpr int kDnaMulti = LM_PAIR_SLOT8 != 0 ? kDnaSlot8 : 5;   // what score_c32_prefilter2_multi reads (lm_hip_pssm::d_image2_multi)
constexpr int pair_syms(int ka) { return ka & 255; }
constexpr bool prefilter2_slot8(int ka) { return (ka & 256) != 0; }
// dwords per table row.
//   16-byte slots (protein; DNA until round 5): 4 * odd >= NPAIR, read 16 bytes at a time -- a 16-lane service group of a
//     ds_read_b128 covers the 64 banks exactly when its lanes sit in 16 DISTINCT slots, which rows 0..15 do.  DNA has 25 pair
//     rows: the nine with an N alias a row without (mod 16), and a lane on an N row costs its group a second cycle whenever
//     another lane reads the aliased row -- 0.4 % of the LDS cycles on ACGT-only input, **25-31 %** with 4.6 % N in the
//     sequence (profiles/r06_stalls_c3_realistic.txt): the whole of the JASPAR batch's 1.3 x on non-i.i.d. input.
//   8-byte slots (the batch's DNA tables, round 6): 2 * odd >= NPAIR, read 8 bytes at a time -- a ds_read_b64 serves 32
//     lanes per cycle from 32 slots of 8 bytes, the same 256 B per clock, and all 25 rows (5 a + b) sit in distinct slots
//     whatever the symbols.  Twice the LDS instructions per byte: on ACGT-only input 1-2 % slower in the batch, 1-6 % in
//     single scans up to M = 20 and 12-19 % in the long ones, which is why those keep the 16-byte slots
//     (profiles/r06_pair_slot_ab.txt, r06_single_scan_slot_ab.txt).
constexpr int prefilter2_stride_dw(int m, int ka = 5)
{
    return prefilter2_slot8(ka) ? 2 * ((prefilter2_npair(m) / 2) | 1) : 4 * (((prefilter2_npair(m) + 3) / 4) | 1);
}
// ... in slots (what the register decode's byte tables count in), and bytes per slot as a shift
constexpr int prefilter2_so(int m, int ka = 5) { return prefilter2_stride_dw(m, ka) / (prefilter2_slot8(ka) ? 2 : 4); }
// table rows = symbol pairs.  Protein (K = 21): 441 rows, a * 21 + b.  Protein rows cannot be conflict-free
// (441 rows, 16 slots): tools/kbench/lds_rows_bench measures 5.5 ns per wavefront read against 3.5 ns for the 21
// rows of the one-symbol prefilter -- but the pair scan needs half the reads.
// DNA (K = 5): row(a, b) is ADDITIVE in the two symbols, so that a lane can look both terms up in registers
// (dna_pair_offsets below).  8-byte slots: 5 a + b, 25 rows.  (16-byte slots: 4 a + b', b' = b for A C T G and 20 for N --
// the 16 pairs without N in rows 0..15, (N, b) in 16..19, (a, N) in 20, 24, 28, 32, (N, N) in 36; the other rows of the 37
// never read.)
constexpr int prefilter2_rows(int ka) { return pair_syms(ka) == 5 ? (prefilter2_slot8(ka) ? 25 : 37) : pair_syms(ka) * pair_syms(ka); }
// (a multiple of 4 dwords: tables are copied 16 bytes at a time, and the tables of a multi-motif pass follow each other)
constexpr int prefilter2_image_dw(int m, int ka = 5) { return (prefilter2_rows(ka) * prefilter2_stride_dw(m, ka) + 3) / 4 * 4; }

__host__ __device__ __forceinline__ unsigned dna_pair_row(unsigned a, unsigned b, bool slot8 = false)
{
    return slot8 ? 5u * a + b : 4u * a + (b == 4u ? 20u : b);
}
template <int KA>
__host__ __device__ __forceinline__ unsigned pair_row(unsigned a, unsigned b)
{
    return pair_syms(KA) == 5 ? dna_pair_row(a, b, prefilter2_slot8(KA)) : a * (unsigned)pair_syms(KA) + b;
}

// DNA decode in registers.  The lanes of a quad hold a 4 x 4 block of symbol bytes (lane q: row r + q, columns
// 4i .. 4i+3).  Picking one symbol apart costs a DPP move + a bit-field extract, and the pair's row another five
// operations -- together more than half of the scan's VALU work once the adds were halved (round 5 counters of
// score_c32_prefilter2<20, 5>: 18.4 VALU per super-step, 10 of them decode; VALU 98 % busy, LDS array 66 %,
// profiles/r05_stalls_prefilter2.txt).  row(a, b) is additive, so the look-up happens BEFORE the bytes change lanes:
//   c = v_perm_b32(table_q, d)     lane q even: byte j = 4 so a(row r + q, column 4i + j), lane q odd: so b'(...)
//   s = c + c[quad lane q ^ 1]     bytes = so row(a, b) of rows (r, r + 1) in lanes 0, 1; of (r + 2, r + 3) in lanes 2, 3
//   y = s[quad lane q ^ 2]         the other pair of rows
//   p = v_perm_b32(y, s, sel_q)    byte 0 / byte 2 = own column's offset of the block's first / second pair
// with `so` = 16-byte slots per table row; four operations per block of four symbols (the transposition that
// preceded the look-up in the first register form cost two more), and the byte select rides on the shift that
// makes the LDS address (SDWA).  Needs the largest row offset below 256 slots: 36 so (16-byte slots: rows of up to 7,
// M' <= 55) or 24 so (8-byte slots: rows of up to 10, M' <= 35 -- M = 36 keeps the one-by-one decode, like longer motifs).
constexpr bool prefilter2_lut_decode(int m, int ka)
{
    return pair_syms(ka) == 5 && (prefilter2_slot8(ka) ? 24 * prefilter2_so(m, ka) <= 255 : prefilter2_so(m, ka) <= 7);
}

struct PairDecode {  // per-lane constants (set once per kernel)
    unsigned tab_lo, tab_hi;  // byte tables of the lane's row parity: 4 so a (even rows of a pair) or so b' (odd rows)
    unsigned sel;             // byte selector of the last step
    unsigned four;            // the shift count of byte_times_16: log2 of the slot size (an SDWA operand must be a register)
};
//
// LIN: which lanes form a "quad".  With four ADJACENT lanes on the four rows of a block, adjacent lanes read addresses 32
// bytes apart, and the symbol loads alone -- nothing else in the kernel -- take 208 us per Gbp (4.8 TB/s), which is what
// every single scan of M <= 20 took whatever its table (tools/kbench/symload_bench: the same loads with adjacent lanes on
// adjacent dwords 173 us, 5.8 TB/s).  LIN = lane l of a half-wave reads dword l of the block's 128 bytes: its row is
// q = l >> 3, its four columns 4 (l & 7) .., and it accumulates column 4 (l & 7) + q; the "quad" of a 4 x 4 tile is the lanes
// {b, 8 + b, 16 + b, 24 + b}.  The exchanges become row_ror:8 (lane ^ 8, a DPP modifier as before) and
// v_permlane16_swap (lane ^ 16: one operation for both directions, where the quad form took a DPP move) -- 7 operations
// per block instead of 6.
template <int SO, bool LIN = false, bool S8 = false>
__device__ __forceinline__ PairDecode pair_decode_setup()
{
    constexpr unsigned RA = S8 ? 5u : 4u, RN = S8 ? 4u : 20u;  // dna_pair_row
    constexpr unsigned A_LO = 0u | (RA * SO << 8) | (2u * RA * SO << 16) | (3u * RA * SO << 24), A_HI = 4u * RA * SO;
    constexpr unsigned B_LO = 0u | (1u * SO << 8) | (2u * SO << 16) | (3u * SO << 24), B_HI = RN * SO;
    // ((4 RA + RN) SO > 255: the kernel decodes one symbol at a time and never reads these -- prefilter2_lut_decode)
    const unsigned q = LIN ? (threadIdx.x >> 3) & 3u : threadIdx.x & 3u;
    PairDecode pd;
    pd.tab_lo = (q & 1u) ? B_LO : A_LO;
    pd.tab_hi = (q & 1u) ? B_HI : A_HI;
    // quad form: lanes 0, 1 hold the first pair in `s` and receive the second in `y`; lanes 2, 3 the other way round.
    // LIN: after the swap the first pair is in one register and the second in the other for every lane.
    pd.sel = (!LIN && (q & 2u)) ? (0x0c000c00u | (4u + q) | (q << 16)) : (0x0c000c00u | q | ((4u + q) << 16));
    pd.four = S8 ? 3u : 4u;
    return pd;
}
// bytes 0 / 2 = row(a, b) * SO of the lane's column for the pairs (rows 0, 1) / (rows 2, 3) of the quad's block `d`
template <bool LIN = false>
__device__ __forceinline__ unsigned dna_pair_offsets(unsigned d, const PairDecode &pd)
{
    const unsigned c = __builtin_amdgcn_perm(pd.tab_hi, pd.tab_lo, d);
    if constexpr (LIN) {
        const unsigned s = c + (unsigned)__builtin_amdgcn_mov_dpp((int)c, 0x128, 0xf, 0xf, true);  // row_ror:8 = lane ^ 8; no byte carries
        // rows of 16 lanes: r[0] = (s.row0, s.row0, s.row2, s.row2) -- the first pair of every lane's tile,
        //                   r[1] = (s.row1, s.row1, s.row3, s.row3) -- the second
        const auto r = __builtin_amdgcn_permlane16_swap(s, s, false, false);
        return __builtin_amdgcn_perm(r[1], r[0], pd.sel);
    } else {
        const unsigned s = c + (unsigned)__builtin_amdgcn_mov_dpp((int)c, 0xb1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]; no byte carries
        const unsigned y = (unsigned)__builtin_amdgcn_mov_dpp((int)s, 0x4e, 0xf, 0xf, true);      // quad_perm [2, 3, 0, 1]
        return __builtin_amdgcn_perm(y, s, pd.sel);
    }
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

// Flags instead of maxima.  The scans only ask whether a completed sum reached the discrete threshold td.  The
// table copy in LDS carries  bias = 0x8000 - td  in both halves of dword 0 of every row -- every output receives
// dword 0 exactly once -- so a sum >= td shows as bit 15 of its half (sums stay below 0x8000 - 2 M': kPrefilterTop,
// no carry into the neighbour), and the completed dwords are OR-ed together: v_or3_b32 folds two per operation where
// the packed maximum took one each.  td > 0x8000 is out of reach of every sum: bias 0, nothing flagged.  The
// other half of a completed dword is a partial sum of an output in flight, as before (score_prefilter.hpp).
constexpr unsigned kFlagBits = 0x80008000u;
static_assert(kPrefilterTop + 2u * (unsigned)kMaxPairM < 0x8000u, "biased sums must stay within their 16-bit half");
__device__ __forceinline__ unsigned prefilter2_bias(unsigned td) { return td <= 0x8000u ? (0x8000u - td) * 0x10001u : 0u; }
// 16 bytes (dwords 4 i .. 4 i + 3) of a table image on their way into LDS: dword 0 of every row of STRIDE dwords gets the bias
template <int STRIDE>
__device__ __forceinline__ uint4 prefilter2_biased(uint4 v, const int i, const unsigned bias)
{
    if constexpr (STRIDE % 4 == 0) {
        if (i % (STRIDE / 4) == 0)
            v.x += bias;
    } else {  // rows of 2 * odd dwords start on every second 8-byte boundary
        const int d = 4 * i;
        if (d % STRIDE == 0)
            v.x += bias;
        if ((d + 2) % STRIDE == 0)
            v.z += bias;
    }
    return v;
}

// (GroupNotes, the per-stream record of flagged groups, lives in score_prefilter.hpp)

// Opaque to the optimiser on purpose: LLVM re-associates chains of integer adds and ORs into trees, which keeps table
// rows and completed sums alive across many super-steps (round 5: hundreds of spilled registers in the unrolled scans).
__device__ __forceinline__ unsigned add3_u32(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned or3_b32(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned or_b32(unsigned a, unsigned b)
{
    unsigned d;
    asm("v_or_b32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Host side: the pair table from the unpadded discrete weights d[j * ka + s], j < m.
// (`ka_layout`: the alphabet size, or kDnaSlot8 for the 8-byte-slot form of a DNA table)
inline void prefilter2_pack_image(const unsigned *d, int m, unsigned *image2, int ka_layout = 5)
{
    const int ka = pair_syms(ka_layout);
    const int mo = prefilter2_mo(m), shift2 = mo - m, np2 = prefilter2_npair(m);
    const int dsd2 = prefilter2_stride_dw(m, ka_layout);
    auto dq = [&](int j, int s) -> unsigned {  // padded weight, 0 outside shift2 .. mo-1
        return (j < shift2 || j >= mo) ? 0u : d[(size_t)(j - shift2) * ka + s];
    };
    for (int i = 0; i < prefilter2_image_dw(m, ka_layout); ++i)
        image2[i] = 0u;
    for (int a = 0; a < ka; ++a)
        for (int b = 0; b < ka; ++b) {
            unsigned *row = image2 + (size_t)(ka == 5 ? dna_pair_row((unsigned)a, (unsigned)b, prefilter2_slot8(ka_layout)) : (unsigned)(a * ka + b)) * dsd2;
            auto entry = [&](int e) -> unsigned { return e > mo ? 0u : dq(e - 1, a) + dq(e, b); };
            for (int w = 0; w < np2; ++w)
                row[w] = entry(2 * w + 1) | (entry(2 * w) << 16);
        }
}

// ---- the scan loop, software-pipelined by hand ------------------------------------------------------------
//
// Symbol loads.  A byte load per lane and row moves 64 bytes per wavefront instruction, and at > 3 Tpos/s the scan
// is bound by that request rate, not by LDS or VALU.  So the four lanes of a "quad" fetch a 4 x 4 block of symbols
// (rows r .. r+3, columns 4i .. 4i+3) with ONE dword load each; a half-wave instruction covers 4 rows = 128 contiguous
// bytes.  Which lanes form a quad matters (pair_decode_setup, LIN): four adjacent lanes put neighbours 32 bytes apart,
// and those loads alone cost 208 us per Gbp; lanes 8 apart -- lane l on dword l -- 173 us (profiles/r05_symload_bench.txt).  RING is a multiple of 4, so blocks never straddle a group; `blk` is a ring of the RING/4
// blocks of a group, requested PFB blocks ahead of their decode.
//
// The schedule is spelled out, not left to the compiler.  Integer adds and ORs may be re-associated, and LLVM pairs an
// accumulator's terms into v_add3_u32 across distant super-steps, which keeps table rows alive for several steps
// (round 5: 36 scratch operations in the unrolled loop of <20, 5> once its decode was cheap; thousands of spilled
// registers in the long kernels).  A PAIR of super-steps (one 4-row block of symbols, table rows R0 and R1) advances
// every accumulator by two entries at once:
//     slot (2p - m) mod NP  +=  R0[m] + R1[m + 1]        m = 1 .. NP - 2     one v_add3_u32 (opaque to the optimiser)
//     slot (2p) mod NP       =  R0[0] + R1[1]            (it completed, and was taken, at the end of the pair before)
//     slot (2p + 1) mod NP  +=  R0[NP - 1]  -> complete (`fin0`); restarts as R1[0]
//     slot (2p + 2) mod NP  completes with m = NP - 2
// NP + 1 operations per pair where one add per entry takes 2 NP.  The rows of the NEXT item -- the next motif of a
// multi-motif pass, else the next pair -- are requested WHILE the current ones are consumed, chunk by chunk (16 bytes
// of each row per chunk): about one item's rows are in flight per wavefront, at most a chunk's worth of extra
// registers is live, and scheduling barriers pin that order.  All LDS addresses are integers: the kernels that use
// this have no static LDS (lds_zero_based).

template <int NP>
struct PairRows {  // the two table rows of a pair of super-steps
    unsigned r0[(NP + 3) / 4 * 4], r1[(NP + 3) / 4 * 4];
};

// chunk C (dwords 4C .. 4C + 3) of the table row at LDS address `off`: a whole 16-byte piece or, when NP % 4 == 2, the
// 8-byte tail read on its own (see prefilter2_mo)
template <int NP, int C, bool S8 = false>
__device__ __forceinline__ void read_row_chunk(unsigned (&w)[(NP + 3) / 4 * 4], const unsigned off)
{
    if constexpr (S8) {  // 8-byte slots: rows start on 8-byte boundaries, every read is a lone ds_read_b64 (volatile: not fused)
        const lm_u32x2_t v = *(lm_lds_u64_ptr)(off + 16u * C);
        w[4 * C + 0] = v.x;
        w[4 * C + 1] = v.y;
        if constexpr (4 * C + 2 < NP) {
            const lm_u32x2_t u = *(lm_lds_u64_ptr)(off + 16u * C + 8u);
            w[4 * C + 2] = u.x;
            w[4 * C + 3] = u.y;
        }
    } else if constexpr (4 * C + 4 <= NP) {
        const lm_u32x4_t v = *(lm_lds_u128_ptr)(off + 16u * C);
        w[4 * C + 0] = v.x;
        w[4 * C + 1] = v.y;
        w[4 * C + 2] = v.z;
        w[4 * C + 3] = v.w;
    } else {
        const lm_u32x2_t v = *(lm_lds_u64_ptr)(off + 16u * C);
        w[4 * C + 0] = v.x;
        w[4 * C + 1] = v.y;
    }
}

template <int NP, int P, int C>
__device__ __forceinline__ void consume_chunk(unsigned (&acc)[NP], const PairRows<NP> &cur, unsigned &fin0)
{
#pragma unroll
    for (int m = 4 * C; m < 4 * C + 4 && m < NP; ++m) {
        const int j = ((2 * P - m) % NP + NP) % NP;
        if (m == NP - 1) {
            fin0 = acc[j] + cur.r0[NP - 1];
            acc[j] = cur.r1[0];
        } else if (m == 0) {
            acc[j] = cur.r0[0] + cur.r1[1];
        } else {
            acc[j] = add3_u32(acc[j], cur.r0[m], cur.r1[m + 1]);
        }
    }
}

template <int NP, int P, int C, bool S8 = false>
__device__ __forceinline__ void pair_chunks(unsigned (&acc)[NP], const PairRows<NP> &cur, PairRows<NP> &nxt, const unsigned off0,
                                            const unsigned off1, unsigned &fin0, const bool has_next)
{
    if constexpr (C < (NP + 3) / 4) {
        if (has_next) {
            read_row_chunk<NP, C, S8>(nxt.r0, off0);
            read_row_chunk<NP, C, S8>(nxt.r1, off1);
        }
        __builtin_amdgcn_sched_barrier(0);
        consume_chunk<NP, P, C>(acc, cur, fin0);
        __builtin_amdgcn_sched_barrier(0);
        pair_chunks<NP, P, C + 1, S8>(acc, cur, nxt, off0, off1, fin0, has_next);
    }
}

// A 4-row block of symbols: one dword per lane.  (Non-temporal loads measured for the single-motif scans, which read the
// sequence once: 1-3 %, profiles/r05_pair_scan_ab.txt -- not worth a second instantiation of every kernel.)
__device__ __forceinline__ unsigned load_block(const uint8_t *__restrict__ p)
{
    return *reinterpret_cast<const unsigned *>(p);
}

// LDS byte offsets of the table rows of a block's two pairs of symbols
template <int M, int KA, bool LIN = false>
__device__ __forceinline__ void decode_block(const unsigned d, const unsigned shq, const PairDecode &pd, unsigned &off0, unsigned &off1)
{
    static_assert(!LIN || prefilter2_lut_decode(M, KA), "the linear lane map exists for the register decode only");
    if constexpr (prefilter2_lut_decode(M, KA)) {
        const unsigned pair_off = dna_pair_offsets<LIN>(d, pd);
        off0 = byte_times_16<0>(pair_off, pd.four);
        off1 = byte_times_16<2>(pair_off, pd.four);
    } else {  // one symbol at a time: own column's byte of a quad neighbour's dword (DPP move + bit-field extract)
        constexpr unsigned DSB = prefilter2_stride_dw(M, KA) * 4;
        off0 = __umul24(pair_row<KA>(quad_symbol<0>(d, shq), quad_symbol<1>(d, shq)), DSB);
        off1 = __umul24(pair_row<KA>(quad_symbol<2>(d, shq), quad_symbol<3>(d, shq)), DSB);
    }
}

// What becomes of the two dwords a pair completes (`fin0`: outputs of super-step 2P, `fin1`: of 2P + 1).
//
// Scans: flags of motif MI (kFlagBits).  The FIRST group completes the stream's outputs 0 and 1 only (its last dword).
template <int NM, bool LIN = false>
struct FlagSink {
    static constexpr bool kLinear = LIN;  // the lane map of the symbol blocks (pair_decode_setup)
    unsigned (&mx)[NM];
    template <int PHASE, int P, int NB, int MI>
    __device__ __forceinline__ void complete(const unsigned fin0, const unsigned fin1)
    {
        if (PHASE != PHASE_FIRST)
            mx[MI] = or3_b32(mx[MI], fin0, fin1);
        else if (P == NB - 1)
            mx[MI] = or_b32(mx[MI], fin1);
    }
};
// Score<u8> (score_u8.hpp): the sums are a DiscreteMatrix's u8 scores.  A pair completes four rows of the lane's
// column, which are transposed inside the quad of lanes and written with one dword store per lane (128 contiguous bytes
// per half-wave instruction; LIN: the quad of a tile is the four lanes 8 apart and lane l writes dword l of the 128 bytes,
// see pair_decode_setup).  `op` = the lane's cell of the group's first completed row; the FIRST group completes
// rows 0 and 1 only (byte stores).  `wrap_mask`: 0 = clamp at 255 (avx2.rs:336), 0xff = mod 256 (Generic's +=).
template <bool LIN = false>
struct StoreSink {
    static constexpr bool kLinear = LIN;  // the lane map of the symbol blocks and of the stored tiles (pair_decode_setup)
    uint8_t *op, *oq;
    unsigned wrap_mask, sel_lo, sel_hi;
    // `o`: the lane's cell (its column `col`) of the group's first completed row
    __device__ __forceinline__ StoreSink(uint8_t *o, unsigned wm, int col) : op(o), wrap_mask(wm)
    {
        const unsigned q = (unsigned)col & 3u;  // lane q of a tile's quad writes row +q, the tile's 4 columns
        oq = op - col + q * 32 + ((unsigned)col >> 2) * 4;
        if constexpr (!LIN) {
            sel_lo = 0x0c0c0000u | q | ((4u + q) << 8);
            sel_hi = 0x00000c0cu | (q << 16) | ((4u + q) << 24);
        } else {
            // the four packs of the quad arrive as (own, lane ^ 16) in one pair of registers and (lane ^ 8, lane ^ 24) in
            // another, which of a pair is which depending on the lane's row of 16 (v_permlane16_swap): selectors per lane
            const bool even = q < 2;
            sel_lo = sel_hi = 0;
            for (unsigned j = 0; j < 4; ++j) {  // output byte j = byte q of the pack of quad lane j = lane ^ (8 (j ^ q))
                const unsigned k = j ^ q;
                const unsigned first = q, second = 4u + q;  // v_perm: 0..3 = bytes of the second source, 4..7 of the first
                unsigned vp = 0x0cu, vx = 0x0cu;
                if (k == 0) vp = even ? first : second;
                if (k == 2) vp = even ? second : first;
                if (k == 1) vx = even ? first : second;
                if (k == 3) vx = even ? second : first;
                sel_lo |= vp << (8 * j);
                sel_hi |= vx << (8 * j);
            }
        }
    }
    __device__ __forceinline__ void advance(const size_t bytes)
    {
        op += bytes;
        oq += bytes;
    }
    __device__ __forceinline__ unsigned narrow(const unsigned a) const
    {   // (lo, hi) = (even row, odd row): clamp or mask both halves at once
        return wrap_mask ? (a & 0x00ff00ffu) : pk_min_u16(a, 0x00ff00ffu);
    }
    template <int PHASE, int P, int NB, int MI>
    __device__ __forceinline__ void complete(const unsigned fin0, const unsigned fin1)
    {
        if (PHASE == PHASE_FIRST) {
            if (P == NB - 1) {
                const unsigned r = narrow(fin1);
                op[0] = (uint8_t)(r & 0xffu);
                op[32] = (uint8_t)(r >> 16);
            }
        } else {
            const unsigned r0 = narrow(fin0), r1 = narrow(fin1);
            const unsigned pack = __builtin_amdgcn_perm(r0, r0, 0x0c0c0200u) | __builtin_amdgcn_perm(r1, r1, 0x02000c0cu);  // rows 4P .. 4P + 3
            unsigned row;
            if constexpr (!LIN) {
                const unsigned p0 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0x00, 0xf, 0xf, true);
                const unsigned p1 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0x55, 0xf, 0xf, true);
                const unsigned p2 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0xaa, 0xf, 0xf, true);
                const unsigned p3 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0xff, 0xf, 0xf, true);
                row = __builtin_amdgcn_perm(p1, p0, sel_lo) | __builtin_amdgcn_perm(p3, p2, sel_hi);
            } else {
                const unsigned x8 = (unsigned)__builtin_amdgcn_mov_dpp((int)pack, 0x128, 0xf, 0xf, true);  // row_ror:8 = lane ^ 8
                const auto rp = __builtin_amdgcn_permlane16_swap(pack, pack, false, false);
                const auto rx = __builtin_amdgcn_permlane16_swap(x8, x8, false, false);
                row = __builtin_amdgcn_perm(rp[1], rp[0], sel_lo) | __builtin_amdgcn_perm(rx[1], rx[0], sel_hi);
            }
            *reinterpret_cast<unsigned *>(oq + 4 * P * 32) = row;
        }
    }
};

// One group = NB pairs x NM motifs, item IDX = P * NM + MI.  On entry `cur` holds (or is about to receive) the rows of
// item IDX, (off0, off1) are the LDS offsets of pair P's rows in motif 0's table, and blk[] holds the PFB blocks after
// block P; on exit `cur` is the next item's (requested), unless this was the stream's last.
// `far`: byte offset, from `spq`, of the group TWO groups ahead -- 2 * RING * 32 -- or, where the stream has no such group (the
// group before the LAST one), of the NEXT group, RING * 32: with PFB == NB the request of the group's last pair reaches block 0
// of the group after next, which for the final stream of a matrix would lie past its wrap rows (a matrix ending on a page
// boundary: a memory fault).  The fallback re-reads a block that surely exists (block 0 of the current group may lie BEFORE
// the matrix: the first stream's padding rows); its value is never used.
template <int M, int KA, int NM, int PFB, int PHASE, int IDX, class Sink>
__device__ __forceinline__ void pair_items(unsigned (&acc)[NM][prefilter2_npair(M)], unsigned (&blk)[prefilter2_ring(M) / 4],
                                           PairRows<prefilter2_npair(M)> &cur, unsigned &off0, unsigned &off1,
                                           const uint8_t *__restrict__ spq, const unsigned shq, const PairDecode &pd, Sink &sink,
                                           const unsigned far)
{
    constexpr int NP = prefilter2_npair(M);
    constexpr int NB = prefilter2_ring(M) / 4;  // = NP / 2 pairs per group
    constexpr unsigned IMG = prefilter2_image_dw(M, KA) * 4;  // bytes per motif's table
    if constexpr (IDX < NB * NM) {
        constexpr int P = IDX / NM, MI = IDX % NM;
        constexpr bool has_next = PHASE != PHASE_LAST || IDX + 1 < NB * NM;
        constexpr unsigned next_table = MI + 1 < NM ? (MI + 1) * IMG : 0u;  // folded into the reads' offset field
        if constexpr (has_next && MI + 1 == NM) {
            // the next item starts pair P + 1 (the next group's pair 0 after the last): decode its block, whose register
            // then takes block P + 1 + PFB
            decode_block<M, KA, Sink::kLinear>(blk[(P + 1) % NB], shq, pd, off0, off1);
            if (PHASE != PHASE_LAST || P + 1 + PFB < NB) {
                if constexpr (P + 1 + PFB >= 2 * NB)
                    blk[(P + 1 + PFB) % NB] = load_block(spq + far + (P + 1 + PFB - 2 * NB) * 128);
                else
                    blk[(P + 1 + PFB) % NB] = load_block(spq + (P + 1 + PFB) * 128);
            }
        }
        PairRows<NP> nxt;
        unsigned fin0 = 0;
        pair_chunks<NP, P, 0, prefilter2_slot8(KA)>(acc[MI], cur, nxt, off0 + next_table, off1 + next_table, fin0, has_next);
        sink.template complete<PHASE, P, NB, MI>(fin0, acc[MI][(2 * P + 2) % NP]);
        if constexpr (has_next) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                cur.r0[i] = nxt.r0[i];
                cur.r1[i] = nxt.r1[i];
            }
        }
        pair_items<M, KA, NM, PFB, PHASE, IDX + 1>(acc, blk, cur, off0, off1, spq, shq, pd, sink, far);
    }
}

template <int NP, int C, bool S8 = false>
__device__ __forceinline__ void pair_begin_rows(PairRows<NP> &cur, const unsigned off0, const unsigned off1)
{
    if constexpr (C < (NP + 3) / 4) {
        read_row_chunk<NP, C, S8>(cur.r0, off0);
        read_row_chunk<NP, C, S8>(cur.r1, off1);
        pair_begin_rows<NP, C + 1, S8>(cur, off0, off1);
    }
}

// The FIRST group's lead-in: block 0 is decoded, item 0's rows are requested, block PFB follows the PFB blocks the
// kernel's prologue requested.
template <int M, int KA, int PFB, bool LIN = false>
__device__ __forceinline__ void pair_begin(unsigned (&blk)[prefilter2_ring(M) / 4], PairRows<prefilter2_npair(M)> &cur,
                                           unsigned &off0, unsigned &off1, const uint8_t *__restrict__ spq, const unsigned shq,
                                           const PairDecode &pd)
{
    constexpr int NP = prefilter2_npair(M);
    constexpr int NB = prefilter2_ring(M) / 4;
    decode_block<M, KA, LIN>(blk[0], shq, pd, off0, off1);
    blk[PFB % NB] = load_block(spq + PFB * 128);
    pair_begin_rows<NP, 0, prefilter2_slot8(KA)>(cur, off0, off1);
}

// wavefronts per SIMD the register budget is cut for.  The hand-pipelined DNA scan keeps 3 NP + ~40 registers live
// (accumulators, one pair of rows, a chunk of the next), the unrolled form about as many with its rows in flight.
constexpr int prefilter2_waves(int m, int ka)
{
    const int np = prefilter2_npair(m);
    if (pair_syms(ka) != 5)
        return m <= 52 ? 4 : m <= 80 ? 3 : 2;
    // (one step below what the registers in use would allow: at the tighter bound the allocator spills a few registers
    // around the loop, and a kernel with ANY scratch pays for it at every wavefront launch -- M = 20: 80 VGPRs either way,
    // 0.302 -> 0.283 ms per Gbp without the 12 bytes of scratch; profiles/r05_pair_scan_ab.txt)
    return np <= 14 ? 5 : np <= 22 ? 4 : np <= 30 ? 3 : 2;
}

template <int M, int KA = 5>
__global__ __launch_bounds__(kBlock, prefilter2_waves(M, KA)) void score_c32_prefilter2(
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
    lds_zero_based(lds_raw);
    {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        const uint4 *src = reinterpret_cast<const uint4 *>(image);
        constexpr int n4 = prefilter2_image_dw(M, KA) / 4;
        const unsigned bias = prefilter2_bias(td);
        for (int i = threadIdx.x; i < n4; i += kBlock)
            dst[i] = prefilter2_biased<prefilter2_stride_dw(M, KA)>(src[i], i, bias);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    // the lane's row of a 4-row symbol block is col & 3 and its four columns 4 (col >> 2) ..; LIN (pair_decode_setup):
    // lane l of a half-wave reads dword l of the block, i.e. row l >> 3
    constexpr bool LIN = prefilter2_lut_decode(M, KA);
    const int col = LIN ? 4 * (lane & 7) + ((lane >> 3) & 3) : lane & 31;
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
    unsigned acc[1][NP];
    unsigned blk[NB];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        acc[0][i] = 0;
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

    // wave-uniform 32-bit bookkeeping (T <= 2^30, score_plan.hip): in 64 bits the division lands on the VALU and
    // drags the group counters into vector registers
    const unsigned ngroups = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)T - 2u) / (unsigned)RING + 1u));  // exact: T = q*RING + 2
    const unsigned G = (ngroups + 63u) / 64u;  // groups per note (= per bit of hit_groups)
    unsigned gleft = G, nnotes = 0;
    unsigned mx[1] = {0};
    GroupNotes notes;
    auto end_group = [&]() {
        if (--gleft == 0) {
            gleft = G;
            ++nnotes;
            notes.note(mx[0]);
        }
    };

    const PairDecode pd = pair_decode_setup<prefilter2_so(M, KA), LIN, prefilter2_slot8(KA)>();
    PairRows<NP> cur;
    unsigned off0, off1;
    FlagSink<1, LIN> sink{mx};
    pair_begin<M, KA, PFB, LIN>(blk, cur, off0, off1, spq, shq, pd);
    constexpr unsigned FAR = 2u * RING * 32u, NEAR = RING * 32u;  // (pair_items: the group two groups ahead, where the stream has one)
    pair_items<M, KA, 1, PFB, PHASE_FIRST, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, ngroups > 2 ? FAR : NEAR);
    end_group();
    for (unsigned g = 1; g + 1 < ngroups; ++g) {
        spq += RING * 32;
        pair_items<M, KA, 1, PFB, PHASE_MAIN, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, g + 2 < ngroups ? FAR : NEAR);
        end_group();
    }
    if (ngroups > 1) {
        spq += RING * 32;
        pair_items<M, KA, 1, PFB, PHASE_LAST, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, 0u);
        end_group();
    }
    if (gleft != G) {  // the last, partly filled note
        ++nnotes;
        notes.note(mx[0]);
    }
    unsigned long long hit_groups = notes.finish(nnotes);

    // flagged groups -> candidate row ranges (group 0: outputs 0, 1; group g >= 1: outputs
    // (g-1)*RING + 2 .. g*RING + 1, counted from the stream's first output row)
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
        const long long i0 = g0 == 0 ? 0 : (long long)((g0 - 1) * RING + 2);
        long long i1 = (long long)((g1 - 1) * RING + 2);
        if (i1 > (long long)T)
            i1 = (long long)T;
        r0 = first_row + i0;
        if (r0 < own_row)
            r0 = own_row;
        r1 = first_row + i1;
    }, lds_raw);
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

// (register budget: at least 4 workgroups per CU; 3 and 5 measure the same, 6 spills: 32 vs 18 ms on the JASPAR argmax batch)
template <int M, int NM>
__global__ __launch_bounds__(kBlock, prefilter2_npair(M) <= 14 ? 4 : 3) void score_c32_prefilter2_multi(
    const uint8_t *__restrict__ seq, const unsigned long long row_begin, const unsigned long long row_end,
    const unsigned long long T, const unsigned long long nstreams, const FusedOut fo_in)
{
    constexpr int KM = kDnaMulti;  // the tables' layout (lm_hip_pssm::d_image2_multi)
    static_assert(prefilter2_image_dw(M, KM) % 4 == 0, "tables are copied 16 bytes at a time");
    constexpr int MO = prefilter2_mo(M);
    constexpr int SHIFT = MO - M;
    constexpr int RING = prefilter2_ring(M);
    constexpr int NP = prefilter2_npair(M);
    constexpr int IMG_DW = prefilter2_image_dw(M, KM);
    const BatchParams *bps = fo_in.batch + (size_t)blockIdx.y * NM;  // this workgroup's NM jobs
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    lds_zero_based(lds_raw);
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw) + mi * (IMG_DW / 4);
        const uint4 *src = static_cast<const uint4 *>(bps[mi].table);
        const unsigned bias = prefilter2_bias(bps[mi].td);
        for (int i = threadIdx.x; i < IMG_DW / 4; i += kBlock)
            dst[i] = prefilter2_biased<prefilter2_stride_dw(M, KM)>(src[i], i, bias);
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

    const unsigned ngroups = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)T - 2u) / (unsigned)RING + 1u));
    const unsigned G = (ngroups + 63u) / 64u;
    unsigned gleft = G, nnotes = 0;
    unsigned mx[NM];
    GroupNotes notes[NM];
#pragma unroll
    for (int mi = 0; mi < NM; ++mi)
        mx[mi] = 0;
    auto end_group = [&]() {
        if (--gleft == 0) {
            gleft = G;
            ++nnotes;
#pragma unroll
            for (int mi = 0; mi < NM; ++mi)
                notes[mi].note(mx[mi]);
        }
    };
    const PairDecode pd = pair_decode_setup<prefilter2_so(M, KM), false, prefilter2_slot8(KM)>();
    PairRows<NP> cur;
    unsigned off0, off1;
    FlagSink<NM> sink{mx};
    pair_begin<M, KM, PFB>(blk, cur, off0, off1, spq, shq, pd);
    constexpr unsigned FAR = 2u * RING * 32u, NEAR = RING * 32u;  // (pair_items: the group two groups ahead, where the stream has one)
    pair_items<M, KM, NM, PFB, PHASE_FIRST, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, ngroups > 2 ? FAR : NEAR);
    end_group();
    for (unsigned g = 1; g + 1 < ngroups; ++g) {
        spq += RING * 32;
        pair_items<M, KM, NM, PFB, PHASE_MAIN, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, g + 2 < ngroups ? FAR : NEAR);
        end_group();
    }
    if (ngroups > 1) {
        spq += RING * 32;
        pair_items<M, KM, NM, PFB, PHASE_LAST, 0>(acc, blk, cur, off0, off1, spq, shq, pd, sink, 0u);
        end_group();
    }
    if (gleft != G) {
        ++nnotes;
#pragma unroll
        for (int mi = 0; mi < NM; ++mi)
            notes[mi].note(mx[mi]);
    }
    unsigned long long hit_groups[NM];
#pragma unroll
    for (int mi = 0; mi < NM; ++mi)
        hit_groups[mi] = notes[mi].finish(nnotes);

    const long long first_row = (long long)(o0 - row_begin);
    const long long own_row = (long long)(stream * T);
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
        FusedOut fo = fo_in;
        fo.job_key = bps[mi].job_key;
        __syncthreads();  // the tables are done with (first motif) / the scratch is reused (the others)
        emit_candidates<true>(idle ? 0ull : hit_groups[mi], col, fo, [=](int bit, long long &r0, long long &r1) {
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
        }, lds_raw);
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
                       (size_t)NM * prefilter2_image_dw(M, kDnaMulti) * 4, stream, seq, row_begin, row_end, T, nstreams, fo);
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
