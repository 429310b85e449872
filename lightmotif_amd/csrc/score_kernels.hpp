// score_kernels.hpp -- device code of the PSSM scoring kernels (gfx950 only).
//
// What is computed (reference: lightmotif/src/pli/mod.rs:96-105, the Generic
// body every SIMD back-end must agree with):
//
//     out[r][c] = (((0.0f + P[0][s(r,c)]) + P[1][s(r+1,c)]) + ... ) + P[M-1][s(r+M-1,c)]
//
// with s(r,c) the symbol byte at row r, column c of the striped sequence.  The
// sum is M *sequential* IEEE f32 adds in motif order -- no re-association, no
// FMA, no pre-summed k-mer tables -- so results are bit-identical to Generic.
//
// C = 32 kernel (`score_c32`): "input-stationary rotating accumulators".
//   * one lane owns ONE column of the striped matrix and sweeps a run of T
//     consecutive rows ("stream"); a wavefront = 32 columns x 2 streams, so every
//     load is a contiguous 32-byte row and every store a contiguous 128-byte row;
//   * the PSSM is staged in LDS *transposed*: tab[s][j] = P[j][s], one padded
//     row of TS floats per symbol.  For the symbol s read at sequence row r the
//     lane fetches the whole column P[0..M-1][s] with ceil(M/4) ds_read_b128
//     (row stride TS = 4*odd floats: the <=16 distinct rows a 16-lane b128 group
//     can touch fall in distinct 4-bank slots, equal rows broadcast -> no bank
//     conflicts for K <= 16);
//   * P[j][s] belongs to output row r-j, so the lane keeps M accumulators in
//     registers, one per in-flight output row; the slot of output row o is
//     (o - o0) mod M, made a compile-time register index by unrolling M steps.
//     Each accumulator therefore receives P[0],P[1],... in order as r advances:
//     exactly the reference's add order;
//   * at step t the slot started at t-M+1 completes and is emitted (stored /
//     compared), and is restarted by the next step.
//   Every symbol byte is read from memory once per stream and every weight comes
//   from LDS; HBM traffic is the algorithmic 1 B in + 4 B out per position.
//
// Generic kernel (`score_generic`): one thread per cell, any C / stride / M / K.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "lm_internal.hpp"

namespace lm {

constexpr int kBlock = 256;                      // threads per workgroup (4 wavefronts)
constexpr int kStreamsPerBlock = kBlock / 32;    // 2 per wavefront
constexpr int kMaxFastM = 36;        // largest motif the whole kernel family (every length, prefilters, u8) is built for
// 36 < M <= 64: the exact f32 kernels only, for the padded lengths M' = 40, 44 ... 64 (leading zero rows, see
// lm_hip_pssm::Part::lead): one pass over the sequence like the reference's AVX2 loop takes for any length
// (avx2.rs:146-193).  M' = 64 needs 163 VGPRs, three wavefronts per SIMD, no scratch (score_long_inst.hip).
constexpr int kMaxLongM = 64;
// ... and the plain STORE kernel alone goes on in one pass to this length (score_xlong_inst.hip: lengths padded to a
// multiple of 8, two wavefronts per SIMD); the other exact modes of such motifs take the pair scan or the chunked route
constexpr int kMaxStoreM = 88;
// The DNA pair-symbol prefilter scan (score_prefilter2.hpp) is length-generic and cheap in registers (M / 2 packed
// accumulators): it is instantiated for every length up to 128, so the FUSED threshold / argmax scans have no cliff
// where the exact kernels end (candidates are re-scored exactly by rescore_candidates, any length).
constexpr int kMaxPairM = 128;

// Floats per symbol row of the transposed LDS table used by score_c32<M>:
// 4 * (smallest odd number >= ceil(M/4)).
// WIDE alphabets (K > 16: protein, K = 21).  A 16-byte LDS read is served in four 16-lane groups over 16
// four-bank slots: 21 symbol rows cannot all sit in different slots, and 44 % of the LDS cycles of the K = 21
// kernels were bank-conflict cycles (7.2 LDS cycles per read instead of 4; profiles/r02_stalls_protein.txt).  An
// 8-byte read is served in two 32-lane groups over 32 two-bank slots (MI355X_MICROARCH.md, LDS table: ds_read_b64
// 2 cycles, 256 B/clk like b128): with rows of 2 * odd dwords up to 32 symbols are conflict-free.  So the WIDE
// kernels read the same column with twice as many single `ds_read_b64` (volatile: two of them fused into a
// ds_read2_b64 cost 8 cycles, which is what round 1's attempt measured and round 2 misread as "4 cycles per
// read whatever its width").  Counters, round 3 (profiles/r03_protein_b64.txt): conflicts 44 % -> 0, 2.0 LDS cycles
// per instruction, store kernel 0.214 -> 0.191 ms per 200 Mres, u16 prefilter scan 0.131 -> 0.111 ms.  DNA keeps
// the 16-byte reads: the 8-byte form costs it 3-10 % (twice the LDS instructions for the same LDS cycles).
constexpr bool lds_wide(int k) { return k > 16; }
constexpr int table_stride(int m, int wide = 0) { return wide ? 4 * ((m + 3) / 4) + 2 : 4 * (((m + 3) / 4) | 1); }
typedef float lm_f32x2_t __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) lm_f32x2_t *lm_lds_b64_ptr;  // volatile: never fused into ds_read2_b64
typedef unsigned lm_u32x2_t __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) lm_u32x2_t *lm_lds_u64_ptr;
// LDS reads at an address computed as an INTEGER (kernels whose dynamic LDS starts at address 0, see lds_zero_based)
typedef unsigned lm_u32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) lm_u32x4_t *lm_lds_u128_ptr;
typedef const __attribute__((address_space(3))) unsigned *lm_lds_u32_ptr;
// A kernel without static LDS has its dynamic LDS at address 0, but the compiler learns that only after instruction
// selection: `table + offset` stays a v_add_u32 with a zero operand in front of every LDS read (one per super-step of
// the pair scans, which are bound by VALU issue).  Such kernels form their LDS addresses from the offset alone and
// check the premise once: a trap (the launch fails loudly) if a static LDS object ever slips in.
__device__ __forceinline__ void lds_zero_based(const void *dynamic_lds)
{
    if ((unsigned)(size_t)(const __attribute__((address_space(3))) char *)dynamic_lds != 0u)
        __builtin_trap();
}

// MODE_CONTINUE: like MODE_STORE, but every output starts from the partial sum already stored in
// its cell instead of +0.0 -- the later passes of a motif longer than kMaxFastM, which is scored
// in slices of <= kMaxFastM rows with the SAME sequential add order (pass 1 stores
// ((0 + P[0]) + ... + P[M1-1]), pass 2 continues with + P[M1] ...), hence bit-identical.
// MODE_STORE_TRACK: MODE_STORE that also tracks (value, cell) of the best score per lane like MODE_ARGMAX -- the
// small-input form of the reference's `score_into` + `argmax` pair (lightmotif-bench dna.rs:104-107): ONE launch,
// the last workgroup to finish folds the workgroup records (FusedOut::final_out).  MODE_STORE_ARGMAX (value only,
// cell located afterwards) stays the large-input form: index tracking costs the store kernel 15 %.
enum : int { MODE_STORE = 0, MODE_ARGMAX = 1, MODE_THRESHOLD = 2, MODE_STORE_ARGMAX = 3, MODE_CONTINUE = 4,
             MODE_STORE_TRACK = 5 };
constexpr bool mode_stores(int mode)
{
    return mode == MODE_STORE || mode == MODE_STORE_ARGMAX || mode == MODE_CONTINUE || mode == MODE_STORE_TRACK;
}
constexpr bool mode_tracks_best(int mode)
{
    return mode == MODE_ARGMAX || mode == MODE_STORE_ARGMAX || mode == MODE_STORE_TRACK;
}
constexpr bool mode_tracks_cell(int mode) { return mode == MODE_ARGMAX || mode == MODE_STORE_TRACK; }

// One above-threshold cell: key = (job << 40) | flat index (row * cols + col).  Flat
// indices stay below 2^40 for anything that fits in 288 GB of HBM.
struct __attribute__((aligned(16))) HitRecord {
    unsigned long long key;
    float value;
    unsigned pad;
};

// One row range of one column that may hold hits: output rows [r0, r0 + nrows) of
// column `col`, r0 relative to the job's row_begin; key = (job << 40) | r0, nrows <= 32.
struct __attribute__((aligned(16))) Candidate {
    unsigned long long key;
    unsigned nrows;
    unsigned col;
};

// Per-job parameters of a multi-job launch (grid.y = jobs of one motif length over the
// same sequence rows): block (x, y) works for job y.
struct BatchParams {
    const void *table;        // transposed f32 table, or the prefilter's LDS image
    ArgmaxRecord *block_best; // MODE_ARGMAX: gridDim.x records of this job
    float threshold;          // MODE_THRESHOLD (exact kernel)
    unsigned td;              // prefilter kernel: discrete threshold
    unsigned long long job_key;
};

struct FusedOut {
    // MODE_ARGMAX: one record per block
    ArgmaxRecord *block_best;
    // MODE_THRESHOLD
    float threshold;
    unsigned long long *hit_count;  // device counter
    HitRecord *hits;                // hit_capacity entries
    unsigned long long job_key;     // batch scans: (job index << 40), OR-ed into the key
    unsigned long long hit_capacity;
    unsigned long long key_rows;    // 0: key = row * cols + col; else key = col * key_rows + row
    // MODE_THRESHOLD of the C = 32 kernels: candidate ranges for rescore_candidates
    unsigned long long *cand_count;
    Candidate *cands;
    unsigned long long cand_capacity;
    // non-null: a multi-job launch, the fields above that differ per job come from batch[blockIdx.y]
    const BatchParams *batch;
    // store kernels: leading all-zero motif rows the table was padded with (0..3) so that M is a
    // multiple of 4 (dword symbol loads): step t then reads sequence row o0 - lead_rows + t
    unsigned lead_rows;
    // MODE_ARGMAX / MODE_STORE_TRACK, single job: non-null = the last workgroup to finish (ticket counter, left
    // at zero again) folds the workgroup records and writes the job's result -- no second launch.  final_host
    // (optional) is a device-visible pinned copy the host reads after synchronising the stream: no copy command
    unsigned *ticket;
    ArgmaxRecord *final_out;
    ArgmaxRecord *final_host;
    int first_cell_rule;
    // written to the word behind *final_host AFTER the record (system-scope release): a host that polls it
    // sees the result a PCIe write after the fold, without waiting for the kernel's completion signal
    unsigned generation;
    // MODE_STORE_TRACK / MODE_ARGMAX of one job, small inputs: non-null = NO fold on the device at all.  Every wavefront writes its
    // (value, cell) with ONE 16-byte store into this pinned array -- {generation, value bits | generation, cell} --
    // the lane that owns the matrix's first cell adds scores[0][0] in the slot behind the last wavefront's, and
    // the HOST folds the few hundred records (handles.hip: host_fold).  The three dependent L2 round trips of the
    // device fold (publish, ticket, read) were 3.8 of the kernel's 10.3 us at the reference's bench size.
    uint4 *host_records;
};

// Ordering used by every argmax reduction: larger value wins; equal values ->
// larger flat index (= later in the reference's row-major scan, pli/mod.rs:144-151
// with `>=`); NaN is never a candidate; index -1 = "no candidate yet".
__device__ __forceinline__ void best_merge(float &v, long long &i, float ov, long long oi)
{
    if (oi >= 0 && (i < 0 || ov > v || (ov == v && oi > i))) {
        v = ov;
        i = oi;
    }
}

__device__ __forceinline__ void best_wave_reduce(float &v, long long &i)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const long long oi = __shfl_xor(i, off);
        best_merge(v, i, ov, oi);
    }
}

// Maximum of a float over the wavefront, NaN operands ignored (NaN only if every lane holds NaN),
// with DPP moves only: `__shfl_xor` is a ds_bpermute, i.e. an LDS-pipeline instruction, and the store
// kernel's epilogue is paid by every short-lived wavefront while the LDS pipeline is its bottleneck
// (18 bpermutes per ~400 table reads).  Result in every lane.
__device__ __forceinline__ float wave_max_dpp(float v)
{
    auto step = [](float x, auto ctrl) {
        const int moved = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x),
                                                      decltype(ctrl)::value, 0xf, 0xf, false);
        return __builtin_fmaxf(x, __builtin_bit_cast(float, moved));
    };
    v = step(v, std::integral_constant<int, 0xb1>{});   // quad_perm [1, 0, 3, 2]
    v = step(v, std::integral_constant<int, 0x4e>{});   // quad_perm [2, 3, 0, 1]
    v = step(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});  // row_mirror: every lane holds its row-of-16 maximum
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return __builtin_fmaxf(__builtin_fmaxf(r0, r1), __builtin_fmaxf(r2, r3));
}

// Block-level reduce of (v, i); result valid in thread 0.  `sm` has room for
// kBlock/64 entries of each type.
template <int BLK = kBlock>
__device__ __forceinline__ void best_block_reduce(float &v, long long &i, float *sm_v,
                                                  long long *sm_i)
{
    best_wave_reduce(v, i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sm_v[wave] = v;
        sm_i[wave] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLK / 64; ++w)
            best_merge(v, i, sm_v[w], sm_i[w]);
    }
}

// Appends one above-threshold cell (row, col) to the (unordered) device hit list.
// key = row-major flat index row * cols + col, or -- for Scanner-style output
// (fo.key_rows != 0) -- the sequence position col * key_rows + row (scores.rs:155-157).
// The lanes of a wavefront that reach this point together share ONE atomicAdd.
__device__ __forceinline__ void record_hit(const FusedOut &fo, unsigned long long row,
                                           unsigned col, unsigned cols, float score)
{
    const unsigned long long active = __ballot(1);
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)active) - 1;
    unsigned long long base = 0;
    if (lane == leader)
        base = atomicAdd(fo.hit_count, (unsigned long long)__popcll(active));
    base = __shfl(base, leader);
    const unsigned long long slot_i = base + __popcll(active & ((1ull << lane) - 1ull));
    if (slot_i < fo.hit_capacity) {
        HitRecord r;
        r.key = fo.job_key | (fo.key_rows ? col * fo.key_rows + row : row * cols + col);
        r.value = score;
        r.pad = 0;
        fo.hits[slot_i] = r;  // one 16-byte store
    }
}

// Tuning constants of score_c32 (every alternative was measured and its losing side removed; HISTORY 4.7-4.8 and the
// profiles/r0*_*_ab.txt files keep the numbers):
//   kScorePF  global prefetch distance in steps of the byte-load form: the symbol byte of step k + PF is requested
//             while step k is processed.  (Issuing the LDS reads of step k + 1 before the adds of step k, batched row
//             stores, non-temporal symbol loads, plain stores and an XCD-aware workgroup remap all measured equal or
//             slower, bit-identical: r01_kbench*, r02_store_batch_ab, r02_store_policy, r03_nt_symbol_loads_ab.)
constexpr int kScorePF = 12;
// Minimum wavefronts per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument).  Without
// it the fused-threshold variant is allocated 228 VGPRs (2 waves/SIMD) although ~75 suffice.
constexpr int score_min_waves_base(int m) { return m <= 20 ? 6 : m <= 40 ? 4 : 3; }

// The long family (M' = 44 ... 64) is compiled against a budget of two wavefronts per SIMD (256 VGPRs).  Most of
// its kernels need 130-170 registers and still run three wavefronts; where the tighter bound made the
// scheduler serialise reads and adds it cost more than the wavefront (plain store, 1 Gbp: M' = 44 1.85 -> 1.50 ms,
// M' = 64 2.24 -> 2.04 ms, the other lengths unchanged; profiles/r03_long_variants_ab.txt), and the fused forms
// from M' = 56 on would spill at the bound of three (15-38 VGPRs at M' = 64).
// The fused forms, measured per length (fused threshold / argmax call, 1 Gbp, bound 3 vs 2: M' = 44 1.66 / 1.88 vs
// 1.41 / 1.28 ms, 48 1.95 / 1.42 vs 1.55 / 1.40, 52 1.86 / 1.56 vs 2.15 / 1.56; profiles/r03_long_variants_ab.txt):
// two everywhere but at M' = 52.
// (short family: the exact fused threshold kernel of M = 29 ... 36 schedules better against a bound of three -- M = 32
//  1.11 -> 0.99 ms, M = 33 1.48 -> 1.04 per Gbp, the other modes and lengths unchanged; tools/fused_ab.sh, round 3)
// (the tracked store of M <= 20 takes 74 VGPRs = six wavefronts where the plain store's 68 allow seven; compiled against a
//  bound of seven it fits 72 + 2 spills and runs 2 % SLOWER at M = 20, equal at 12 / 16: profiles/r03_track_ab.txt)
constexpr int score_min_waves(int m, int mode)
{
    return m <= 40 ? ((mode == 2 /* MODE_THRESHOLD */ && m > 28 && m <= 36) ? 3
                      : (mode == 3 /* MODE_STORE_ARGMAX */ && m <= 20) ? 6 : score_min_waves_base(m))
                   : mode == 0 /* MODE_STORE */ ? 2 : m == 52 ? 3 : 2;
}

// MODE_CONTINUE: how many steps ahead a row's partial sum is requested (see score_group)
constexpr int kContinueAhead = 4;
// (M' = 52, the one continuation kernel on a bound of three wavefronts per SIMD, gets SLOWER with the ring: 2.28 -> ~3.1 ms
//  per Gbp; it keeps the step-ahead form.  profiles/r03_continue_ahead_ab.txt)
constexpr int continue_ahead(int m) { return (m % kContinueAhead == 0 && m != 52) ? kContinueAhead : 1; }
// Upper bound of the wavefronts per SIMD (second argument of amdgpu_waves_per_eu; 0 = none, i.e. plain
// __launch_bounds__).  hipcc schedules two tracked store kernels of the long family badly when it is free to aim at
// a higher occupancy than the bound asks for: M' = 56 with 1 424 `s_waitcnt` instead of the plain store's 823 (LDS
// reads next to their uses) and M' = 40 alike -- 2.33 / 1.60 ms per Gbp against 1.87 / 1.41 for the plain store.
// Capped at their lower bound the register allocator stops trading the schedule for a wavefront it does not get:
// 484 waits, 2.00 / 1.43 ms.  The same cap on every other store kernel is a disaster (M' = 44 1.52 -> 3.11 ms,
// M = 24 1.04 -> 1.88; profiles/r03_maxw_ab.txt), hence the two lengths by name.
constexpr int score_max_waves(int m, int mode, int minw)
{
    return (mode == 3 /* MODE_STORE_ARGMAX */ && (m == 40 || m == 56)) ? minw : 0;
}

template <int M, int WIDE = 0>
__device__ __forceinline__ void lds_fetch_column(float (&w)[4 * ((M + 3) / 4)],
                                                 const char *__restrict__ tab, const unsigned s)
{
    constexpr unsigned TSB = table_stride(M, WIDE) * 4;  // bytes per symbol row
    if constexpr (WIDE != 0) {
        const char *row8 = static_cast<const char *>(__builtin_assume_aligned(tab + __umul24(s, TSB), 8));
#pragma unroll
        for (int q = 0; q < (M + 1) / 2; ++q) {
            const lm_f32x2_t v = *(lm_lds_b64_ptr)(row8 + 8 * q);
            w[2 * q + 0] = v.x;
            w[2 * q + 1] = v.y;
        }
        return;
    }
    // M floats: whole 16-byte reads, then an 8- and/or 4-byte read for the rest
    const char *row =
        static_cast<const char *>(__builtin_assume_aligned(tab + __umul24(s, TSB), 16));
#pragma unroll
    for (int q = 0; q < M / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(row + 16 * q);
        w[4 * q + 0] = v.x;
        w[4 * q + 1] = v.y;
        w[4 * q + 2] = v.z;
        w[4 * q + 3] = v.w;
    }
    if (M % 4 >= 2) {
        const float2 v = *reinterpret_cast<const float2 *>(row + 16 * (M / 4));
        w[4 * (M / 4) + 0] = v.x;
        w[4 * (M / 4) + 1] = v.y;
    }
    if (M % 2 == 1)
        w[M - 1] = *reinterpret_cast<const float *>(row + 4 * (M - 1));
}

// Edge groups (FIRST / LAST) need only part of the column: floats [4*C0, 4*C1).  Left to itself the
// compiler narrows those reads to the exact floats (ds_read_b32 / b64: 85 LDS instructions in the
// LAST group of M = 20 instead of 55), and the pieces cost the LDS as much as the read they replace while fetching
// less (a b64 + a b32 = 2 + 2 LDS cycles, a b128 = 4: MI355X_MICROARCH.md's LDS table, confirmed by round 3's
// counters on the WIDE kernels).  `volatile`
// keeps them whole: 9 % fewer LDS cycles.  Interleaved A/B per motif length on one box
// (profiles/r02_edge_ab.txt, 1 Gbp, best stream length each): M = 12 0.872 vs 0.884 ms, M = 16
// 0.915 vs 0.927, M = 24 1.001 vs 1.004, M = 28 1.077 vs 1.089, M = 33 1.223 vs 1.253, M = 36
// 1.296 vs 1.306 -- but M = 20 0.958 vs 0.940 at every stream length (the volatile reads also
// pin the schedule of the edge groups; at five chunks per column that costs more than the
// narrow reads do), hence the hole in the middle.
constexpr bool edge_reads_whole(int m) { return m <= 16 || m >= 24; }
template <int M, int WIDE = 0>
__device__ __forceinline__ void lds_fetch_chunks(float (&w)[4 * ((M + 3) / 4)], const char *__restrict__ tab,
                                                 const unsigned s, const int c0, const int c1)
{
    constexpr unsigned TSB = table_stride(M, WIDE) * 4;
    const char *row = static_cast<const char *>(__builtin_assume_aligned(tab + __umul24(s, TSB), WIDE ? 8 : 16));
#pragma unroll
    for (int q = 0; q < (M + 3) / 4; ++q) {
        if (q < c0 || q >= c1)  // folds: the bounds are constants of the unrolled step
            continue;
        if constexpr (WIDE != 0) {
            const lm_f32x2_t a = *(lm_lds_b64_ptr)(row + 16 * q), b = *(lm_lds_b64_ptr)(row + 16 * q + 8);
            w[4 * q + 0] = a.x;
            w[4 * q + 1] = a.y;
            w[4 * q + 2] = b.x;
            w[4 * q + 3] = b.y;
            continue;
        }
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        typedef const volatile __attribute__((address_space(3))) f32x4_t *lds_ptr;  // explicit LDS space:
        const f32x4_t v = *(lds_ptr)(row + 16 * q);  // a volatile generic pointer would become flat_load
        w[4 * q + 0] = v.x;
        w[4 * q + 1] = v.y;
        w[4 * q + 2] = v.z;
        w[4 * q + 3] = v.w;
    }
}

// Quad-gathered symbol loads: the lanes of a quad (columns 4i..4i+3) fetch a 4 x 4 block of
// symbols with ONE dword load each (lane q: row r+q, columns 4i..4i+3) and every lane picks
// its own column out of its neighbours' registers: symbol(row r+t) = byte (lane & 3) of the
// dword held by quad lane t.  `shift` = 8 * (lane & 3).
template <int T4>
__device__ __forceinline__ unsigned quad_symbol(unsigned block_dword, unsigned shift)
{
    const unsigned x = (unsigned)__builtin_amdgcn_mov_dpp((int)block_dword, T4 * 0x55, 0xf, 0xf, true);
    return (x >> shift) & 0xffu;
}

// One bit per NOTE (= G groups of a stream), up to 64 per stream, collected without a compare AND without a 64-bit shift by
// a register (score_prefilter.hpp, prefilter_lookahead: such a shift by the last allocated VGPR is what broke the protein scans): the field shifts right one bit per
// note and takes the note's flag in at bit 63, so after n notes they sit in its top n bits, oldest lowest.
struct GroupNotes {
    unsigned lo = 0, hi = 0;
    // `mx`: bit 15 of a half = that half reached the threshold (the pair scans' biased sums, score_prefilter2.hpp: kFlagBits)
    __device__ __forceinline__ void note(unsigned &mx)
    {
        const unsigned either = mx | (mx << 16);  // bit 31: one of the two halves reached the threshold
        push(either);
        mx = 0;
    }
    __device__ __forceinline__ void push(const unsigned flag31)  // bit 31 of `flag31` = the note's flag
    {
        lo = __builtin_amdgcn_alignbit(hi, lo, 1);  // (hi:lo) >> 1
        hi = (hi >> 1) | (flag31 & 0x80000000u);
    }
    __device__ __forceinline__ unsigned long long finish(unsigned n) const  // n = notes taken, 1 ... 64 (wave-uniform)
    {
        // (hi:lo) >> (64 - n) with 32-bit operations only
        const unsigned sh = 64u - n;
        const unsigned a = sh >= 32u ? hi : lo, b = sh >= 32u ? 0u : hi, r = sh & 31u;
        const unsigned out_lo = r ? (a >> r) | (b << (32u - r)) : a;
        const unsigned out_hi = r ? (b >> r) : b;
        return ((unsigned long long)out_hi << 32) | out_lo;
    }
};

// ---- the linear lane map of 4-row symbol blocks (round 6; score_prefilter_blk.hpp) -----------------------------------
// Lane l of a half-wave reads dword l of a block's 128 bytes (row q = l >> 3, columns 4b .. 4b + 3, b = l & 7) and accumulates
// column 4b + q.  The four lanes {b, 8 + b, 16 + b, 24 + b} transpose their 4 x 4 bytes with two exchanges -- DPP row_ror:8
// (lane ^ 8) and v_permlane16_swap (lane ^ 16) -- and two v_perm_b32, after which a lane holds its column's four symbols in ONE
// register; the LDS address of a step is then byte i of that register times the row size, one SDWA multiply (kernels without
// static LDS: the product IS the address, lds_zero_based).  The protein block scan is built on it.  The f32 store kernel was
// tried on it too (2.25 decode operations per step instead of 3, 451 instead of 466 VALU per group at M = 20, bit-identical)
// and ran SLOWER at M = 20 ... 28 (0.947 against 0.921 ms per Gbp at M = 20; profiles/r06_store_linear_map_ab.json): it keeps
// the quad map of quad_symbol above.
struct BlkTranspose {
    unsigned sel1, sel2;
    __device__ __forceinline__ BlkTranspose()
    {
        const unsigned q = (threadIdx.x >> 3) & 3u;
        // v_perm_b32(S0, S1, sel): selector bytes 0..3 take from S1, 4..7 from S0
        // step 1, S0 = partner row (q ^ 1), S1 = own: lanes of an even row keep columns (0, 2) of rows (q, q + 1), of an
        // odd row columns (1, 3) of rows (q - 1, q)
        sel1 = (q & 1u) ? 0x03070105u : 0x06020400u;
        // step 2, S0 = the pair of rows (2, 3), S1 = the pair (0, 1) (v_permlane16_swap hands both to every lane): the
        // lane's column is the first of its pair for q < 2, the second for q >= 2
        sel2 = (q & 2u) ? 0x07060302u : 0x05040100u;
    }
    __device__ __forceinline__ unsigned operator()(const unsigned d) const
    {
        const unsigned x = (unsigned)__builtin_amdgcn_mov_dpp((int)d, 0x128, 0xf, 0xf, true);  // row_ror:8 = lane ^ 8
        const unsigned u = __builtin_amdgcn_perm(x, d, sel1);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // r[0]: rows (0, 1) of the tile, r[1]: rows (2, 3)
        return __builtin_amdgcn_perm(r[1], r[0], sel2);
    }
};

// byte BYTE of `s` times `mult` (a register: SDWA takes no constants) in one operation
template <int BYTE>
__device__ __forceinline__ unsigned byte_times(const unsigned s, const unsigned mult)
{
    unsigned r;
    if constexpr (BYTE == 0)
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(s), "v"(mult));
    else if constexpr (BYTE == 1)
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(s), "v"(mult));
    else if constexpr (BYTE == 2)
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(s), "v"(mult));
    else
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(s), "v"(mult));
    return r;
}

enum : int { PHASE_FIRST = 0, PHASE_MAIN = 1, PHASE_LAST = 2 };

// One group of M consecutive steps of one lane.  `sp` -> symbol byte of the
// group's first step; `tbase` = index of that step within the stream (wave-uniform;
// the output row completed by step t is o0 + t - (M-1)); `orow` = output row completed by
// the group's first step (in the FIRST group only step M-1 completes a row).
// `sym` is a ring of symbol bytes indexed by step mod M (only ~PF are live).
template <int M, int MODE, int PF, int PHASE, int QL = 0, int OC = 32, int WIDE = 0>
__device__ __forceinline__ void score_group(float (&acc)[M], unsigned (&sym)[M],
                                            const uint8_t *__restrict__ sp,
                                            const char *__restrict__ tab,
                                            float *__restrict__ op, const unsigned tbase,
                                            const int col, float &best_v, unsigned &best_t,
                                            const FusedOut &fo, const unsigned shq, float &init_next,
                                            float (&initq)[kContinueAhead], const bool prelast)
{
    constexpr int NW = 4 * ((M + 3) / 4);
    constexpr int NB = M / 4;                  // QL: 4-row symbol blocks per group (M % 4 == 0)
    constexpr int PFB = NB > 3 ? 3 : NB;       // QL: blocks requested ahead of use
#pragma unroll
    for (int k = 0; k < M; ++k) {
        // (1) request the symbol byte PF steps ahead (stays inside the stream's
        //     T+M-1 input rows: the LAST group does not look past its end)
        unsigned s_now = 0;
        if (QL) {
            // sym[] holds dword blocks (see quad_symbol); `sp` = this lane's row of block 0
            const unsigned d = sym[k / 4];
            s_now = (k % 4 == 0)   ? quad_symbol<0>(d, shq)
                    : (k % 4 == 1) ? quad_symbol<1>(d, shq)
                    : (k % 4 == 2) ? quad_symbol<2>(d, shq)
                                   : quad_symbol<3>(d, shq);
            if (k % 4 == 3 && (PHASE != PHASE_LAST || k / 4 + PFB < NB))
                sym[(k / 4 + PFB) % NB] = *reinterpret_cast<const unsigned *>(sp + (k / 4 + PFB) * 128);
        } else if (PF > 0) {
            if (PHASE != PHASE_LAST || k + PF < M)
                sym[(k + PF) % M] = sp[(k + PF) * 32];
        } else {
            sym[k] = sp[k * 32];
        }
        // (2) the PSSM column of this step's symbol
        float w[NW];
        if (edge_reads_whole(M) && PHASE != PHASE_MAIN) {
            // FIRST: outputs started at steps 0..k -> weights 0..k.  LAST: the stream's last output
            // starts at the group's step 0, so step k still needs weights k..M-1
            const unsigned sy = QL ? s_now : sym[k];
#pragma unroll
            for (int i = 0; i < NW; ++i)
                w[i] = 0.0f;
            if (PHASE == PHASE_FIRST)
                lds_fetch_chunks<M, WIDE>(w, tab, sy, 0, k / 4 + 1);
            else
                lds_fetch_chunks<M, WIDE>(w, tab, sy, k / 4, (M + 3) / 4);
        } else if (QL) {
            lds_fetch_column<M, WIDE>(w, tab, s_now);
        } else {
            lds_fetch_column<M, WIDE>(w, tab, sym[k]);
        }
        // MODE_CONTINUE: the output started at this step resumes from the partial sum in its cell
        // (requested one step ago); request the next step's.  The row started at step k is stored
        // at op + (k + M - 1) * 32.  Only rows of the stream are ever requested: in the LAST group
        // step 0 starts the stream's last row, later starts are dead.
        float init = 0.0f;
        if (MODE == MODE_CONTINUE) {
            // a ring of CD partial sums in flight: the one requested CD steps ago is used, the row started CD steps
            // from now is requested (M % CD == 0 keeps the ring's phase from group to group).  One step ahead the
            // kernel waited out most of an HBM read latency per step at two or three wavefronts per SIMD:
            // <48,4> 2.24 ms per Gbp against 1.59 for <48,0>, <64,4> 3.21 against 1.96.  Starts beyond step 0 of the
            // LAST group are dead rows (possibly past the matrix): in the group before it those requests go to a
            // row of this group instead (`prelast`, wave-uniform: two v_cndmask on CD - 1 steps per group).
            constexpr int CD = continue_ahead(M);  // 1 (the step-ahead form) for lengths that are no multiple of the ring
            init = initq[k % CD];
            if (PHASE != PHASE_LAST) {
                const float *src = op + (k + CD + M - 1) * OC;
                if (k + CD > M)
                    src = prelast ? op + (M - 1) * OC : src;
                initq[k % CD] = *src;
            }
        }
        // (3) P[j][s] goes to the output row started j steps ago: slot (k - j) mod M.
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int slot = (k - j + M) % M;
            if (j == 0)
                acc[slot] = init + w[0];  // T::default() + P[0][s]   (pli/mod.rs:98,101), or the partial sum
            else
                acc[slot] = acc[slot] + w[j];
        }
        // (4) the slot started at step k-(M-1) is complete.
        if (PHASE != PHASE_FIRST || k == M - 1) {
            const float score = acc[(k + 1) % M];
            if (mode_tracks_cell(MODE) && PHASE == PHASE_FIRST)
                init_next = score;  // (MODE_CONTINUE's carrier is free in these modes) the stream's first output
            if (mode_stores(MODE))
                __builtin_nontemporal_store(score, op + k * OC);  // streaming: the matrix is never read back by this kernel
            if (MODE == MODE_STORE_ARGMAX) {
                // value only (one v_max_f32; NaN operands are ignored, the NaN start value
                // survives only if every score was NaN): the cell is located afterwards in
                // the stored matrix.  Tracking the index too cost the store kernel 15 %.
                best_v = __builtin_fmaxf(best_v, score);
            } else if (mode_tracks_cell(MODE)) {
                if (score >= best_v) {  // same `>=` as pli/mod.rs:146, NaN never passes
                    best_v = score;
                    best_t = tbase + k;  // scalar + constant: no per-lane arithmetic
                }
            } else if (MODE == MODE_THRESHOLD) {
                // Hits are rare (a p = 1e-5 tail).  The hot loop only tracks the
                // lane's "saw a hit" flag (v_cmp + v_cndmask); a group with a
                // flagged lane is re-scored out of line by rescan_rows.  Recording
                // hits inline at every unrolled step cost 230 VGPRs and a third of
                // the throughput.
                if (score >= fo.threshold)  // pli/mod.rs:215
                    best_t = 1;
            }
        }
    }
}

// Tail of the fused threshold kernels.  After its stream a lane only knows WHICH row
// ranges of its column may hold a hit (one bit per range in `hit_groups`); it appends
// them, cut into pieces of at most 32 rows, to the candidate list.  The pieces are
// re-scored exactly by `rescore_candidates` (score_threshold.hip) -- like the reference Scanner,
// which collects candidate positions from the discrete scores and re-scores them with
// `score_position` (scan.rs:179-190) -- with one half-wave per piece, so that the
// symbol loads of neighbouring rows share cache lines.  `range(bit, r0, r1)` maps a
// bit to output rows [r0, r1) relative to row_begin.//
// The whole workgroup reserves its slots with ONE global atomicAdd (atomics on the
// single list counter serialise in L2 at ~6 ns each; a counter bump per wavefront and
// loop trip cost 0.4 ms per launch at 1e6 candidates): count, workgroup prefix sum,
// reserve, write.  Every thread of the workgroup must call this.
//
// OVERLAY: the 72 bytes of workgroup scratch live at `overlay` (16-byte aligned dynamic LDS the caller no longer
// needs -- the pair scans hand in their table, behind a barrier) instead of in static LDS.  A kernel without
// static LDS has its dynamic LDS at address 0, and the table-row address of every LDS read of its hot loop is then
// the row offset itself: one v_add_u32 per super-step less in scans that are bound by VALU issue.
template <bool OVERLAY = false, typename Range>
__device__ __forceinline__ void emit_candidates(const unsigned long long hit_groups, const int col,
                                                const FusedOut &fo, Range range, char *overlay = nullptr)
{
    unsigned *wave_total;
    unsigned long long *block_base_p;
    if constexpr (OVERLAY) {
        block_base_p = reinterpret_cast<unsigned long long *>(overlay);
        wave_total = reinterpret_cast<unsigned *>(overlay + 8);
    } else {
        __shared__ unsigned wave_total_s[16];
        __shared__ unsigned long long block_base_s;
        wave_total = wave_total_s;
        block_base_p = &block_base_s;
    }
    unsigned long long &block_base = *block_base_p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    unsigned mine = 0;
    for (unsigned long long m = hit_groups; m; m &= m - 1) {
        long long r0, r1;
        range(__ffsll((long long)m) - 1, r0, r1);
        if (r1 > r0)
            mine += (unsigned)((r1 - r0 + 31) / 32);
    }
    // Hits are rare (p ~ 1e-5 per cell): almost every wavefront has nothing to report, and the prefix
    // sum below is six ds_bpermute -- LDS-pipeline instructions, which the scans that end here are bound
    // by (one epilogue per ~64-row stream and motif: ~8 % of the pair scan's LDS instructions).
    unsigned incl = mine;
    if (__ballot(mine != 0)) {  // wavefront-uniform
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(incl, off);
            if (lane >= off)
                incl += y;
        }
    }
    if (lane == 63)
        wave_total[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < nwaves; ++w)
            tot += wave_total[w];
        block_base = tot ? atomicAdd(fo.cand_count, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    unsigned long long slot = block_base + incl - mine;
    for (int w = 0; w < wave; ++w)
        slot += wave_total[w];
    for (unsigned long long m = hit_groups; m; m &= m - 1) {
        long long r0, r1;
        range(__ffsll((long long)m) - 1, r0, r1);
        for (; r0 < r1; r0 += 32, ++slot) {
            if (slot < fo.cand_capacity) {
                Candidate c;
                c.key = fo.job_key | (unsigned long long)r0;
                c.nrows = (unsigned)(r1 - r0 < 32 ? r1 - r0 : 32);
                c.col = (unsigned)col;
                fo.cands[slot] = c;  // one 16-byte store
            }
        }
    }
}

// C = 32, seq stride 32 B, out stride 32 floats.  Grid: ceil(nstreams / 8) blocks.
// Every stream sweeps exactly T = q*M + 1 output rows, q >= 1, i.e. T + M - 1 =
// (q+1)*M steps = one FIRST group, q-1 MAIN groups and one LAST group.  The last
// stream is shifted back so that it ends at row_end; idle half-waves re-do the
// last stream (identical values -> benign duplicates).
template <int M, int MODE, int QLREQ = 0, int OC = 32, int WIDE = 0>
__global__ __attribute__((amdgpu_flat_work_group_size(1, kBlock),
                          amdgpu_waves_per_eu(score_min_waves(M, MODE),
                                              score_max_waves(M, MODE, score_min_waves(M, MODE))))) void score_c32(
    const uint8_t *__restrict__ seq, const float *__restrict__ table, const int K,
    const unsigned long long row_begin, const unsigned long long row_end,
    const unsigned long long T, const unsigned long long nstreams, float *__restrict__ out,
    const FusedOut fo_in)
{
    constexpr int BLK = kBlock, PF = kScorePF;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    FusedOut fo = fo_in;
    if (!mode_stores(MODE) && fo_in.batch) {  // multi-job launch: this block's job (wave-uniform)
        const BatchParams bp = fo_in.batch[blockIdx.y];
        table = static_cast<const float *>(bp.table);
        fo.block_best = bp.block_best;
        fo.threshold = bp.threshold;
        fo.job_key = bp.job_key;
    }
    {
        // (WIDE rows are 2 * odd dwords: the table is no multiple of 16 bytes)
        typedef std::conditional_t<WIDE != 0, float2, float4> vec_t;
        vec_t *dst = reinterpret_cast<vec_t *>(lds_raw);
        const vec_t *src = reinterpret_cast<const vec_t *>(table);
        const int n4 = K * table_stride(M, WIDE) / (WIDE ? 2 : 4);
        for (int i = threadIdx.x; i < n4; i += BLK)
            dst[i] = src[i];
    }
    __syncthreads();

    // OC = 16 (the 16-lane back-ends' column count; plain store only): a wavefront carries four streams
    // of 16 columns, score rows are 16 floats; the sequence rows keep their 32-byte stride (dense.rs:43-48)
    static_assert(OC == 32 || (OC == 16 && MODE == MODE_STORE), "C = 16 is built for the plain store kernel");
    // quad-gathered symbol loads need whole 4-row blocks per group
    constexpr int QL = (QLREQ && M % 4 == 0) ? 1 : 0;
    const int lane = threadIdx.x & 63;
    const int col = lane & (OC - 1);
    // (plain dispatch order = one compact window of rows in flight; an XCD-aware remap -- each XCD one contiguous
    //  eighth of the rows -- measured slower on this write pattern: HISTORY 4.7)
    const unsigned long long bid = blockIdx.x;
    unsigned long long stream = (bid * (BLK / 64) + (threadIdx.x >> 6)) * (64 / OC) + lane / OC;
    const bool idle = stream >= nstreams;  // re-does the last stream, reports nothing
    if (MODE == MODE_CONTINUE && idle)
        return;  // in-place continuation: every cell may be read and rewritten exactly once
    if (idle)
        stream = nstreams - 1;
    unsigned long long o0 = row_begin + stream * T;
    if (o0 + T > row_end)
        o0 = row_end - T;

    const unsigned shq = 8u * (col & 3);
    // (padded motifs: the first `lead` rows of a window carry zero weights and may lie before the matrix)
    // (the fused modes of the short family are never launched on padded tables; the long family always is)
    const long long lead = (mode_stores(MODE) || M > kMaxFastM) ? (long long)fo_in.lead_rows : 0;
    const long long in0 = (long long)o0 - lead;
    const uint8_t *sp = QL ? seq + (in0 + (col & 3)) * 32 + (col >> 2) * 4 : seq + in0 * 32 + col;
    // output row completed by step t is o0 + t - (M-1); `op` tracks step 0 of the group
    const long long orow = (long long)(o0 - row_begin) - (M - 1);
    float *op = mode_stores(MODE) ? out + orow * OC + col : nullptr;

    constexpr int PFE = PF < M ? PF : M - 1;  // look-ahead stays inside one group
    float acc[M];
    unsigned sym[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        acc[j] = 0.0f;
        sym[j] = 0;
    }
    if (QL) {
        constexpr int NB = M / 4, PFB = NB > 3 ? 3 : NB;
#pragma unroll
        for (int j = 0; j < PFB; ++j)
            if (M <= kMaxLongM ? (j > 0 || in0 + (col & 3) >= 0)  // lead <= 3: only block 0 can start before row 0; its rows
                               : in0 + 4 * j + (long long)(col & 3) >= 0)  // (beyond kMaxLongM: lead <= 7, two blocks can)
                sym[j] = *reinterpret_cast<const unsigned *>(sp + j * 128);  // meet zero weights whatever they hold
    } else {
#pragma unroll
        for (int j = 0; j < PFE; ++j)
            sym[j] = sp[j * 32];
    }
    float best_v = (MODE == MODE_STORE_ARGMAX) ? __builtin_nanf("") : -INFINITY;
    // argmax mode: step index of the lane's best score (0xffffffff = none);
    // threshold mode: "this group saw a hit" flag
    unsigned best_t = (MODE == MODE_THRESHOLD) ? 0u : 0xffffffffu;
    unsigned tbase = 0;
    // MODE_CONTINUE: partial sum of the row started at step 0 (= the stream's first row)
    float init_next = 0.0f;
    float initq[kContinueAhead];
#pragma unroll
    for (int d = 0; d < kContinueAhead; ++d)  // rows started at steps 0 .. CD-1 of the stream (T >= M + 1 rows: they exist)
        initq[d] = (MODE == MODE_CONTINUE && d < continue_ahead(M)) ? op[(d + M - 1) * OC] : 0.0f;

    const unsigned long long ngroups = (T + M - 1) / M;  // exact: T = q*M + 1, >= 2

    // Fused threshold: the hot loop must stay branch-free (any `if` inside the
    // unrolled groups wrecks the schedule: 230 VGPRs / spills), so a lane only
    // remembers WHICH groups saw a score >= t -- one bit per G groups (GroupNotes) -- and
    // re-scores those rows after its stream.  Hits are rare (p ~ 1e-5).
    const unsigned G = (unsigned)((ngroups + 63) / 64);  // groups per note (T <= 2^30: 32 bits hold every count here)
    unsigned gleft = G, nnotes = 0, pending = 0;
    GroupNotes notes;
    auto note_group = [&]() {
        if (MODE == MODE_THRESHOLD) {
            pending |= best_t ? 0x80000000u : 0u;
            best_t = 0;
            if (--gleft == 0) {
                gleft = G;
                ++nnotes;
                notes.push(pending);
                pending = 0;
            }
        }
    };

    score_group<M, MODE, PFE, PHASE_FIRST, QL, OC, WIDE>(acc, sym, sp, lds_raw, op, tbase, col, best_v,
                                                best_t, fo, shq, init_next, initq, ngroups == 2);
    const float first_out = init_next;  // MODE_STORE_TRACK: the stream's first output (scores[0][0] for stream 0, column 0)
    note_group();
    for (unsigned long long g = 1; g + 1 < ngroups; ++g) {
        sp += M * 32;
        tbase += M;
        if (mode_stores(MODE))
            op += M * OC;
        score_group<M, MODE, PFE, PHASE_MAIN, QL, OC, WIDE>(acc, sym, sp, lds_raw, op, tbase, col,
                                                   best_v, best_t, fo, shq, init_next, initq, g + 2 == ngroups);
        note_group();
    }
    sp += M * 32;
    tbase += M;
    if (mode_stores(MODE))
        op += M * OC;
    score_group<M, MODE, PFE, PHASE_LAST, QL, OC, WIDE>(acc, sym, sp, lds_raw, op, tbase, col, best_v,
                                               best_t, fo, shq, init_next, initq, false);
    note_group();
    if (MODE == MODE_THRESHOLD && gleft != G) {  // the last, partly filled note
        ++nnotes;
        notes.push(pending);
    }
    const unsigned long long hit_groups = MODE == MODE_THRESHOLD ? notes.finish(nnotes ? nnotes : 1u) : 0ull;

    if (MODE == MODE_THRESHOLD) {
        const long long first_row = (long long)(o0 - row_begin);  // = orow0 + M - 1
        // every cell is reported once: the shifted last stream skips the rows the
        // stream before it owns, idle half-waves report nothing
        const long long own_row = (long long)(stream * T);
        // bit b <-> groups [b*G, (b+1)*G): group g completes output rows
        // orow0 + g*M .. orow0 + g*M + M-1, of which group 0 only the stream's first row
        emit_candidates(idle ? 0ull : hit_groups, col, fo, [=](int bit, long long &r0, long long &r1) {
            const unsigned long long g0 = (unsigned long long)bit * G;
            unsigned long long g1 = g0 + G;
            if (g1 > ngroups)
                g1 = ngroups;
            r0 = first_row - (M - 1) + (long long)(g0 * M);
            if (r0 < own_row)
                r0 = own_row;  // >= first_row
            r1 = first_row - (M - 1) + (long long)(g1 * M);
        });
    }

    if (MODE == MODE_STORE_ARGMAX) {
        // one record per WAVEFRONT (no workgroup barrier at the end of a short-lived
        // workgroup: with ~80 steps per stream the barrier + LDS reduce cost 15 %);
        // "index" = the workgroup, ties go to the later rows
        // (every lane of the wavefront would report the same workgroup: only the value needs reducing)
        best_v = wave_max_dpp(best_v);
        const long long idx = best_v != best_v ? -1 : (long long)bid;
        if ((threadIdx.x & 63) == 0) {
            ArgmaxRecord *rec = fo.block_best + (size_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
            rec->value = best_v;
            rec->index = idx;
            rec->found = idx >= 0;
        }
    } else if (mode_tracks_cell(MODE) && fo.host_records) {
        // no device-side fold: one 16-byte record per wavefront straight into pinned host memory
        long long idx = best_t != 0xffffffffu
                            ? ((long long)(o0 - row_begin) + (long long)best_t - (M - 1)) * 32 + col
                            : -1;
        best_wave_reduce(best_v, idx);
        const unsigned gen = fo.generation;
        if ((threadIdx.x & 63) == 0)
            fo.host_records[(size_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6)] =
                make_uint4(gen, __builtin_bit_cast(unsigned, best_v), gen, idx >= 0 ? (unsigned)idx : 0xffffffffu);
        if (!idle && stream == 0 && col == 0 && o0 == row_begin)
            fo.host_records[(size_t)gridDim.x * (BLK / 64)] = make_uint4(gen, __builtin_bit_cast(unsigned, first_out), gen, 0u);
    } else if (mode_tracks_best(MODE)) {
        // the table is dead: reuse the dynamic LDS (>= 64 B) as reduction scratch
        __syncthreads();
        long long *sm_i = reinterpret_cast<long long *>(lds_raw);
        float *sm_v = reinterpret_cast<float *>(lds_raw + 32);
        // step t completes output row o0 + t - (M-1)
        long long idx = best_t != 0xffffffffu
                            ? ((long long)(o0 - row_begin) + (long long)best_t - (M - 1)) * 32 + col
                            : -1;
        best_block_reduce<BLK>(best_v, idx, sm_v, sm_i);
        if (threadIdx.x == 0) {
            fo.block_best[blockIdx.x].value = best_v;
            fo.block_best[blockIdx.x].index = idx;
            fo.block_best[blockIdx.x].found = idx >= 0;
        }
        if (fo.final_out) {
            // The last workgroup to finish folds the records (Generic rule: best_merge) and applies the
            // first-cell rule (pli/mod.rs:142-146) -- what argmax_finalize does in a launch of its own.
            __shared__ int is_last;
            if (threadIdx.x == 0) {
                __threadfence();  // this workgroup's record (and score rows) before its ticket
                is_last = atomicAdd(fo.ticket, 1u) == gridDim.x - 1;
            }
            __syncthreads();
            if (is_last) {  // workgroup-uniform
                __threadfence();
                float v = -INFINITY;
                long long i = -1;
                // (requested before the records so that the two round trips overlap)
                unsigned first_bits = 0;
                if (mode_stores(MODE) && fo.first_cell_rule && threadIdx.x == 0)
                    first_bits = __hip_atomic_load(reinterpret_cast<const unsigned *>(out), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned b = threadIdx.x; b < gridDim.x; b += BLK) {
                    const unsigned long long vf = __hip_atomic_load(
                        reinterpret_cast<const unsigned long long *>(&fo.block_best[b]), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT);  // {value, found}
                    const long long bi = __hip_atomic_load(&fo.block_best[b].index, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(vf >> 32))
                        best_merge(v, i, __builtin_bit_cast(float, (unsigned)vf), bi);
                }
                __syncthreads();  // (sm_v / sm_i are reused)
                best_block_reduce<BLK>(v, i, sm_v, sm_i);
                if (threadIdx.x == 0) {
                    if (fo.first_cell_rule) {
                        float first;
                        if (mode_stores(MODE)) {  // the stored cell itself (written by whichever workgroup owns row 0)
                            first = __builtin_bit_cast(float, first_bits);
                        } else {  // recomputed in the reference's add order from the transposed table
                            first = 0.0f;
                            const uint8_t *s00 = seq + row_begin * 32;
                            for (int j = (int)lead; j < M; ++j)
                                first = first + table[s00[(j - (int)lead) * 32] * table_stride(M, WIDE) + j];
                        }
                        if (first != first) {  // NaN: nothing ever compares >= it
                            v = first;
                            i = 0;
                        }
                    }
                    ArgmaxRecord o;
                    o.value = v;
                    o.index = i;
                    o.found = i >= 0;
                    *fo.final_out = o;
                    *fo.ticket = 0u;
                    if (fo.final_host) {
                        *fo.final_host = o;
                        __hip_atomic_store(reinterpret_cast<unsigned *>(fo.final_host + 1), fo.generation, __ATOMIC_RELEASE,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
    }
}

// Any column count / stride / motif length / alphabet: one thread per cell.
// `pssm` is the dense M x K matrix; it is staged in LDS when `use_lds`.
template <int MODE>
__global__ __launch_bounds__(kBlock) void score_generic(
    const uint8_t *__restrict__ seq, const unsigned long long seq_stride, const int cols,
    const float *__restrict__ pssm, const int M, const int K, const int use_lds,
    const unsigned long long row_begin, const unsigned long long row_end, float *__restrict__ out,
    const unsigned long long out_stride, const FusedOut fo)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const float *tab = pssm;
    if (use_lds) {
        float *dst = reinterpret_cast<float *>(lds_raw);
        for (int i = threadIdx.x; i < M * K; i += kBlock)
            dst[i] = pssm[i];
        __syncthreads();
        tab = dst;
    }
    const unsigned long long ncells = (row_end - row_begin) * (unsigned long long)cols;
    float best_v = -INFINITY;
    long long best_i = -1;
    for (unsigned long long cell = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
         cell < ncells; cell += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long r = cell / cols;
        const int c = (int)(cell - r * cols);
        const uint8_t *sp = seq + (row_begin + r) * seq_stride + c;
        float score = 0.0f;                       // pli/mod.rs:98
        for (int j = 0; j < M; ++j)               // pli/mod.rs:99-102
            score = score + tab[j * K + sp[j * seq_stride]];
        if (MODE == MODE_STORE) {
            out[r * out_stride + c] = score;      // pli/mod.rs:103
        } else if (MODE == MODE_ARGMAX) {
            if (score >= best_v) {
                best_v = score;
                best_i = (long long)cell;
            }
        } else {
            if (score >= fo.threshold)
                record_hit(fo, r, (unsigned)c, (unsigned)cols, score);
        }
    }
    if (MODE == MODE_ARGMAX) {
        __syncthreads();
        long long *sm_i = reinterpret_cast<long long *>(lds_raw);
        float *sm_v = reinterpret_cast<float *>(lds_raw + 32);
        best_block_reduce(best_v, best_i, sm_v, sm_i);
        if (threadIdx.x == 0) {
            fo.block_best[blockIdx.x].value = best_v;
            fo.block_best[blockIdx.x].index = best_i;
            fo.block_best[blockIdx.x].found = best_i >= 0;
        }
    }
}

// Any column count / stride / motif length / alphabet, materialising, tiled: the fallback that
// is more than a correctness path (C = 16 geometries of the reference's 16-lane back-ends,
// M > 64, unaligned matrices).  A workgroup stages the symbols of TR + M - 1 rows x `cols` live
// columns (padding bytes skipped) and the dense PSSM in LDS; a thread then owns a strip of
// kTiledStrip vertically adjacent outputs of one column and slides down its M + strip - 1
// input rows: ONE symbol read feeds up to `strip` outputs (weights P[rho - i][s] for the
// outputs i = rho - j), so the LDS traffic is ~1.1 reads per add instead of the 2 global byte
// loads + 1 table read per add of score_generic.  Each output still receives P[0], P[1], ...
// in order from +0.0: the reference's add sequence (pli/mod.rs:98-102), bit-identical.
constexpr int kTiledStrip = 8;

template <int STRIP, int KT>  // KT: the alphabet size at compile time (5, 21), 0 = read it at run time
__global__ __launch_bounds__(kBlock) void score_tiled(
    const uint8_t *__restrict__ seq, const unsigned long long seq_stride, const int cols,
    const float *__restrict__ pssm, const int M, const int K_rt, const unsigned long long row_begin,
    const unsigned long long row_end, const int TR, float *__restrict__ out,
    const unsigned long long out_stride, uint4 *__restrict__ host_records, const unsigned generation)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int K = KT ? KT : K_rt;
    // host_records (small inputs, see FusedOut::host_records): every wavefront also leaves its best (value, cell)
    // in pinned memory -- Generic rule: greater value, ties to the greater row-major cell, NaN never
    float best_v = -INFINITY;
    long long best_c = -1;
    float *tab = reinterpret_cast<float *>(lds_raw);                       // M x K
    uint8_t *tile = reinterpret_cast<uint8_t *>(lds_raw) + (((size_t)M * K * 4 + 15) / 16) * 16;
    for (int i = threadIdx.x; i < M * K; i += kBlock)
        tab[i] = pssm[i];
    const unsigned long long r0 = row_begin + (unsigned long long)blockIdx.x * TR;
    const unsigned long long nout = row_end - r0 < (unsigned long long)TR ? row_end - r0 : (unsigned long long)TR;
    const unsigned long long nin = nout + M - 1;  // input rows r0 .. r0 + nin - 1 (wrap rows exist)
    if (seq_stride == (unsigned long long)cols && (reinterpret_cast<uintptr_t>(seq + r0 * seq_stride) & 3) == 0) {
        // contiguous tile: dword copies
        const unsigned *src = reinterpret_cast<const unsigned *>(seq + r0 * seq_stride);
        unsigned *dst = reinterpret_cast<unsigned *>(tile);
        const unsigned long long nbytes = nin * cols, ndw = nbytes / 4;
        for (unsigned long long i = threadIdx.x; i < ndw; i += kBlock)
            dst[i] = src[i];
        for (unsigned long long i = ndw * 4 + threadIdx.x; i < nbytes; i += kBlock)
            tile[i] = seq[r0 * seq_stride + i];
    } else {
        const unsigned long long n = nin * cols;
        for (unsigned long long i = threadIdx.x; i < n; i += kBlock) {
            const unsigned long long r = i / cols;
            tile[i] = seq[(r0 + r) * seq_stride + (i - r * cols)];
        }
    }
    __syncthreads();
    // strips: strip index -> (strip row block, column); consecutive threads take consecutive
    // columns of one strip row block, so their tile reads are consecutive bytes
    const unsigned long long nsr = (nout + STRIP - 1) / STRIP;
    for (unsigned long long sidx = threadIdx.x; sidx < nsr * cols; sidx += kBlock) {
        const unsigned long long sr = sidx / cols;
        const int c = (int)(sidx - sr * cols);
        const unsigned long long o = sr * STRIP;  // first output row of the strip (tile-relative)
        const int n = (int)(nout - o < (unsigned long long)STRIP ? nout - o : (unsigned long long)STRIP);
        float acc[STRIP];
#pragma unroll
        for (int i = 0; i < STRIP; ++i)
            acc[i] = 0.0f;  // T::default() (pli/mod.rs:98)
        const uint8_t *tp = tile + o * cols + c;
        // input row o + rho feeds the outputs i with j = rho - i in [0, M): weight tab[(rho - i) * K + s]
        if (n == STRIP && M >= STRIP) {
            // full strip: ramp-up (outputs 0..rho), steady state (all STRIP outputs, no tests), ramp-down
#pragma unroll
            for (int rho = 0; rho < STRIP - 1; ++rho) {
                const float *base = tab + tp[(size_t)rho * cols] + rho * K;
#pragma unroll
                for (int i = 0; i <= rho; ++i)
                    acc[i] = acc[i] + base[-i * K];
            }
            for (int rho = STRIP - 1; rho < M; ++rho) {
                const float *base = tab + tp[(size_t)rho * cols] + rho * K;
#pragma unroll
                for (int i = 0; i < STRIP; ++i)
                    acc[i] = acc[i] + base[-i * K];
            }
#pragma unroll
            for (int d = 1; d < STRIP; ++d) {
                const float *base = tab + tp[(size_t)(M - 1 + d) * cols] + (M - 1 + d) * K;
#pragma unroll
                for (int i = d; i < STRIP; ++i)
                    acc[i] = acc[i] + base[-i * K];
            }
        } else {
            for (int rho = 0; rho < M + n - 1; ++rho) {
                const float *trow = tab + tp[(size_t)rho * cols];
#pragma unroll
                for (int i = 0; i < STRIP; ++i) {
                    const int j = rho - i;
                    if (i < n && j >= 0 && j < M)
                        acc[i] = acc[i] + trow[j * K];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < STRIP; ++i)
            if (i < n) {
                out[(r0 - row_begin + o + i) * out_stride + c] = acc[i];
                if (host_records && acc[i] == acc[i])
                    best_merge(best_v, best_c, acc[i], (long long)((r0 - row_begin + o + i) * cols + c));
            }
        if (host_records && blockIdx.x == 0 && sidx == 0)  // scores[0][0], for the first-cell rule
            host_records[gridDim.x] = make_uint4(generation, __builtin_bit_cast(unsigned, acc[0]), generation, 0u);
    }
    if (host_records) {  // ONE record per workgroup: the host reads every record's cache line from memory
        __shared__ float sm_v[kBlock / 64];
        __shared__ long long sm_i[kBlock / 64];
        best_block_reduce(best_v, best_c, sm_v, sm_i);
        if (threadIdx.x == 0)
            host_records[blockIdx.x] =
                make_uint4(generation, __builtin_bit_cast(unsigned, best_v), generation, best_c >= 0 ? (unsigned)best_c : 0xffffffffu);
    }
}

// The few rows a continuation pass (MODE_CONTINUE) cannot cover with whole streams: one thread
// per cell, `out` holds the partial sum, `pssm` the slice's rows (M x K), `seq` row 0 = the
// slice's first input row of output row 0.
template <int UNUSED>
__global__ __launch_bounds__(kBlock) void score_continue_cells(
    const uint8_t *__restrict__ seq, const unsigned long long seq_stride, const int cols,
    const float *__restrict__ pssm, const int M, const int K, const unsigned long long row_begin,
    const unsigned long long row_end, float *__restrict__ out, const unsigned long long out_stride)
{
    const unsigned long long ncells = (row_end - row_begin) * (unsigned long long)cols;
    for (unsigned long long cell = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; cell < ncells;
         cell += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long r = cell / cols;
        const int c = (int)(cell - r * cols);
        const uint8_t *sp = seq + (row_begin + r) * seq_stride + c;
        float score = out[r * out_stride + c];
        for (int j = 0; j < M; ++j)
            score = score + pssm[j * K + sp[j * seq_stride]];
        out[r * out_stride + c] = score;
    }
}

// Host-side launch shim, one per (M, MODE), defined in score_inst_*.hip.
using ScoreC32Launcher = hipError_t (*)(dim3 grid, size_t lds_bytes, hipStream_t stream,
                                        const uint8_t *seq, const float *table, int K,
                                        unsigned long long row_begin, unsigned long long row_end,
                                        unsigned long long T, unsigned long long nstreams,
                                        float *out, FusedOut fo);

template <int M, int MODE, int QL = 0, int OC = 32, int WIDE = 0>
hipError_t score_c32_launch(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq,
                            const float *table, int K, unsigned long long row_begin,
                            unsigned long long row_end, unsigned long long T,
                            unsigned long long nstreams, float *out, FusedOut fo)
{
    hipLaunchKernelGGL((score_c32<M, MODE, QL, OC, WIDE>), grid, dim3(kBlock), lds_bytes, stream, seq, table, K,
                       row_begin, row_end, T, nstreams, out, fo);
    return hipGetLastError();
}

// Filled by the score_inst_*.hip translation units; [M][MODE], nullptr if absent.
// Registry row: [0..2] = modes, [3..6] = unused, [7] = store kernel with quad-gathered symbol
// loads (M % 4 == 0), [8] = store + running maximum (score_into on handles).
constexpr int kRegistrySlots = 12;  // [9] = MODE_CONTINUE (later passes of motifs longer than kMaxFastM),
                                    // [10] = store kernel for C = 16 (M % 4 == 0, quad loads),
                                    // [11] = MODE_STORE_TRACK (M % 4 == 0, quad loads): small score_into + argmax
// `wide`: the kernels for alphabets of more than 16 symbols (lds_wide(K): 8-byte LDS reads)
ScoreC32Launcher score_c32_lookup(int M, int mode, bool wide = false);
ScoreC32Launcher score_c32_lookup_ql(int M, bool wide = false);
ScoreC32Launcher score_c32_lookup_store_argmax(int M, bool wide = false);
ScoreC32Launcher score_c32_lookup_continue(int M, bool wide = false);
ScoreC32Launcher score_c32_lookup_c16(int M, bool wide = false);
ScoreC32Launcher score_c32_lookup_store_track(int M, bool wide = false);
const char *score_c32_name(int M, int mode);

}  // namespace lm
