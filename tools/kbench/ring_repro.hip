// ring_repro -- reproducer of the round-5 parity fault of the one-symbol prefilter scans.
//
// The shipped scan (`score_c32_prefilter<M, PF, WIDE>`, score_prefilter.hpp) keeps a ring of MP symbol registers
// refilled PFE steps ahead of their use.  With PFE = MP - 1 the protein kernels of M = 7, 8 (MP = 8) lost a
// run-to-run varying quarter of their candidates on MI355X.  This program runs THE SHIPPED step function
// (`lm::prefilter_group`) inside a kernel with the scan's stream geometry, once per look-ahead:
//     PFE = MP - 1   the faulty form
//     PFE = MP - 2   the shipped form (prefilter_lookahead)
//     PFE = 0        no ring at all: every step loads its own symbol (the referee)
// and counts the lanes whose 64-bit flag word differs from the referee's, per launch.
//
//   ./ring_repro [--hsaco FILE] [residues = 50e6] [launches = 8] [td as a fraction of the largest sum = 0.644: about half of the groups flagged] [q = 63: streams of q MP + 1 rows]
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I../../include -I../../lightmotif_amd/csrc ring_repro.hip -o ring_repro
//   (-DRING_M=8 -DRING_WIDE=1 by default; tools/ring_isa.sh dumps the ISA of the three kernels)
#include "score_prefilter.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#ifndef RING_M
#define RING_M 8
#endif
#ifndef RING_WIDE
#define RING_WIDE 1
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

using namespace lm;

// The body of score_c32_prefilter (score_prefilter.hpp:217-328) with the look-ahead as a template argument and the
// candidate list replaced by one flag word per lane.
template <int M, int PFE, int WIDE>
__global__ __launch_bounds__(kBlock, 6) void ring_scan(const uint8_t *__restrict__ seq, const unsigned *__restrict__ image,
                                                       const int K, const unsigned long long rows, const unsigned long long T,
                                                       const unsigned long long nstreams, const unsigned td,
                                                       unsigned long long *__restrict__ flags)
{
    constexpr int MP = prefilter_mp(M);
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
    unsigned long long stream = ((unsigned long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    if (stream >= nstreams)
        stream = nstreams - 1;
    unsigned long long o0 = stream * T;
    if (o0 + T > rows)
        o0 = rows - T;
    const uint8_t *sp = seq + (long long)(o0 - SHIFT) * 32 + col;
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
        if (j == 0 && SHIFT) {
            if (o0 > 0)
                sym[0] = sp[0];
        } else {
            sym[j] = sp[j * 32];
        }
    }
    const unsigned long long ngroups = (T + MP - 1) / MP;
    unsigned long long hit_groups = 0;
    const unsigned long long G = (ngroups + 63) / 64;
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
    prefilter_group<M, PFE, PHASE_FIRST, 0, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx);
    note_group();
    for (unsigned long long g = 1; g + 1 < ngroups; ++g) {
        sp += MP * 32;
        prefilter_group<M, PFE, PHASE_MAIN, 0, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx);
        note_group();
    }
    sp += MP * 32;
    prefilter_group<M, PFE, PHASE_LAST, 0, WIDE>(acc2, sym, sp, tab_even, tab_odd, mx);
    note_group();
    flags[stream * 32 + col] = hit_groups;
}

static hipModule_t g_module = nullptr;  // --hsaco: the kernels come out of this code object (tools/ring_isa.py)

template <int PFE>
static void launch(const uint8_t *seq, const unsigned *image, int K, unsigned long long rows, unsigned long long T,
                   unsigned long long nstreams, unsigned td, unsigned long long *flags)
{
    const size_t lds = (size_t)prefilter_image_dw(RING_M, K) * 4;
    const unsigned blocks = (unsigned)((nstreams + 7) / 8);
    if (g_module) {
        char name[96];
        snprintf(name, sizeof name, "_Z9ring_scanILi%dELi%dELi%dEEvPKhPKjiyyyjPy", RING_M, PFE, RING_WIDE);
        hipFunction_t f;
        CK(hipModuleGetFunction(&f, g_module, name));
        void *args[] = {&seq, &image, &K, &rows, &T, &nstreams, &td, &flags};
        CK(hipModuleLaunchKernel(f, blocks, 1, 1, kBlock, 1, 1, (unsigned)lds, 0, args, nullptr));
        return;
    }
    hipLaunchKernelGGL((ring_scan<RING_M, PFE, RING_WIDE>), dim3(blocks), dim3(kBlock), lds, 0, seq, image, K, rows, T,
                       nstreams, td, flags);
    CK(hipGetLastError());
}

// ---- part 2: the SHIPPED kernel lm::score_c32_prefilter<M, PF, WIDE> itself (needs -DLM_RING_LOOKAHEAD_RAW, so that PF passes
// through prefilter_lookahead unchanged), candidate list and all
template <int PF>
static void launch_shipped(const uint8_t *seq, const unsigned *image, int K, unsigned long long rows, unsigned long long T,
                           unsigned long long nstreams, unsigned td, FusedOut fo)
{
    const size_t lds = (size_t)prefilter_image_dw(RING_M, K) * 4;
    const unsigned blocks = (unsigned)((nstreams + 7) / 8);
    unsigned long long row_begin = 0;
    if (g_module) {
        char name[128];
        snprintf(name, sizeof name, "_ZN2lm19score_c32_prefilterILi%dELi%dELi%dEEEvPKhPKjiyyyyjNS_8FusedOutE", RING_M, PF, RING_WIDE);
        hipFunction_t f;
        CK(hipModuleGetFunction(&f, g_module, name));
        void *args[] = {&seq, &image, &K, &row_begin, &rows, &T, &nstreams, &td, &fo};
        CK(hipModuleLaunchKernel(f, blocks, 1, 1, kBlock, 1, 1, (unsigned)lds, 0, args, nullptr));
        return;
    }
    hipLaunchKernelGGL((score_c32_prefilter<RING_M, PF, RING_WIDE>), dim3(blocks), dim3(kBlock), lds, 0, seq, image, K, row_begin, rows, T,
                       nstreams, td, fo);
    CK(hipGetLastError());
}

static std::vector<unsigned long long> candidate_keys(const FusedOut &fo, unsigned long long *count_out)
{
    unsigned long long n = 0;
    CK(hipMemcpy(&n, fo.cand_count, 8, hipMemcpyDeviceToHost));
    *count_out = n;
    std::vector<Candidate> c(std::min<unsigned long long>(n, fo.cand_capacity));
    CK(hipMemcpy(c.data(), fo.cands, c.size() * sizeof(Candidate), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> keys;
    for (const Candidate &x : c)
        keys.push_back((x.key << 14) | ((unsigned long long)x.col << 8) | x.nrows);
    std::sort(keys.begin(), keys.end());
    return keys;
}

int main(int argc, char **argv)
{
    if (argc > 2 && std::string(argv[1]) == "--hsaco") {
        CK(hipModuleLoad(&g_module, argv[2]));
        printf("kernels from %s\n", argv[2]);
        argv += 2;
        argc -= 2;
    }
    const unsigned long long residues = argc > 1 ? (unsigned long long)atof(argv[1]) : 50000000ull;
    const int launches = argc > 2 ? atoi(argv[2]) : 8;
    const double tdfrac = argc > 3 ? atof(argv[3]) : 0.644;
    constexpr int M = RING_M, MP = prefilter_mp(M), K = RING_WIDE ? 21 : 5;
    const unsigned long long q = argc > 4 ? strtoull(argv[4], nullptr, 10) : 63ull;  // groups per stream - 1 (<= 63: one flag bit per group)
    const unsigned long long T = q * MP + 1;
    const unsigned long long nstreams = residues / 32 / T;
    const unsigned long long rows = nstreams * T;
    std::mt19937_64 rng(0x5EED0005);
    std::vector<uint8_t> h_seq((size_t)(rows + MP) * 32 + 4096);
    for (auto &b : h_seq)
        b = (uint8_t)(rng() % (K - 1));
    // discrete weights d[j * K + s]: row 0 is the zero padding row of an odd M
    std::vector<unsigned> d((size_t)MP * K, 0u);
    const unsigned wmax = kPrefilterTop / MP;
    for (int j = MP - M; j < MP; ++j)
        for (int s = 0; s < K; ++s)
            d[(size_t)j * K + s] = (unsigned)(rng() % wmax);
    std::vector<unsigned> h_image(prefilter_image_dw(M, K));
    prefilter_pack_image(d.data(), M, K, h_image.data());
    const unsigned td = (unsigned)(tdfrac * wmax * M);

    uint8_t *seq;
    unsigned *image;
    unsigned long long *f_ref, *f_a, *f_b;
    const size_t nflags = (size_t)nstreams * 32;
    CK(hipMalloc(&seq, h_seq.size()));
    CK(hipMalloc(&image, h_image.size() * 4));
    CK(hipMalloc(&f_ref, nflags * 8));
    CK(hipMalloc(&f_a, nflags * 8));
    CK(hipMalloc(&f_b, nflags * 8));
    CK(hipMemcpy(seq, h_seq.data(), h_seq.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(image, h_image.data(), h_image.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned long long> ref(nflags), a(nflags), b(nflags);
    printf("ring_repro: M = %d (MP = %d) WIDE = %d K = %d, %llu rows x 32, T = %llu, %llu streams, td = %u\n", M, MP, RING_WIDE, K, rows, T,
           nstreams, td);
    unsigned long long bad_a = 0, bad_b = 0, flagged = 0;
    for (int it = 0; it < launches; ++it) {
        CK(hipMemset(f_ref, 0xff, nflags * 8));
        CK(hipMemset(f_a, 0xff, nflags * 8));
        CK(hipMemset(f_b, 0xff, nflags * 8));
        launch<0>(seq + 32, image, K, rows, T, nstreams, td, f_ref);
        launch<MP - 1>(seq + 32, image, K, rows, T, nstreams, td, f_a);
        launch<MP - 2>(seq + 32, image, K, rows, T, nstreams, td, f_b);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), f_ref, nflags * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(a.data(), f_a, nflags * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), f_b, nflags * 8, hipMemcpyDeviceToHost));
        unsigned long long wa = 0, wb = 0, lost_a = 0, extra_a = 0;
        for (size_t i = 0; i < nflags; ++i) {
            wa += a[i] != ref[i];
            wb += b[i] != ref[i];
            lost_a += __builtin_popcountll(ref[i] & ~a[i]);
            extra_a += __builtin_popcountll(a[i] & ~ref[i]);
            if (it == 0)
                flagged += __builtin_popcountll(ref[i]);
        }
        printf("launch %d: look-ahead MP-1: %llu wrong lanes of %zu (%llu flags lost, %llu extra); look-ahead MP-2: %llu wrong lanes\n", it, wa,
               nflags, lost_a, extra_a, wb);
        bad_a += wa;
        bad_b += wb;
    }
#ifdef LM_RING_LOOKAHEAD_RAW
    {
        static_assert(prefilter_lookahead(MP - 1, MP) == MP - 1, "built without -DLM_RING_LOOKAHEAD_RAW");
        const unsigned td2 = (unsigned)((argc > 5 ? atof(argv[5]) : 0.86) * wmax * M);  // a ~1e-4 tail of the sums
        FusedOut fo{};
        const unsigned long long cap = 1ull << 22;
        CK(hipMalloc(&fo.cand_count, 8));
        CK(hipMalloc(&fo.cands, cap * sizeof(Candidate)));
        fo.cand_capacity = cap;
        unsigned long long shipped_bad = 0, shipped_bad_cured = 0;
        for (int it = 0; it < launches; ++it) {
            unsigned long long n_ref, n_a, n_b;
            CK(hipMemset(fo.cand_count, 0, 8));
            launch_shipped<1>(seq + 32, image, K, rows, T, nstreams, td2, fo);
            CK(hipDeviceSynchronize());
            const auto k_ref = candidate_keys(fo, &n_ref);
            CK(hipMemset(fo.cand_count, 0, 8));
            launch_shipped<MP - 1>(seq + 32, image, K, rows, T, nstreams, td2, fo);
            CK(hipDeviceSynchronize());
            const auto k_a = candidate_keys(fo, &n_a);
            CK(hipMemset(fo.cand_count, 0, 8));
            launch_shipped<MP - 2>(seq + 32, image, K, rows, T, nstreams, td2, fo);
            CK(hipDeviceSynchronize());
            const auto k_b = candidate_keys(fo, &n_b);
            printf("shipped kernel, launch %d: candidates referee (look-ahead 1) %llu, look-ahead MP-1 %llu (%s), look-ahead MP-2 %llu (%s)\n", it,
                   n_ref, n_a, k_a == k_ref ? "same" : "DIFFERENT", n_b, k_b == k_ref ? "same" : "DIFFERENT");
            shipped_bad += k_a != k_ref;
            shipped_bad_cured += k_b != k_ref;
        }
        printf("RESULT shipped_kernel_lookahead_mp_minus_1_wrong_launches=%llu of %d, mp_minus_2_wrong_launches=%llu\n", shipped_bad, launches,
               shipped_bad_cured);
    }
#endif
    printf("referee flags per launch: %llu of %zu groups\n", flagged, nflags * 64);
    printf("RESULT lookahead_mp_minus_1_wrong_lanes=%llu lookahead_mp_minus_2_wrong_lanes=%llu\n", bad_a, bad_b);
    return bad_b ? 1 : 0;
}
