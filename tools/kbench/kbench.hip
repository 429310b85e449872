// kbench -- standalone sweep of score_c32 variants on one GPU (development tool).
// Build: see tools/kbench/build.sh.  Not part of the product library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <cstring>
#include <vector>

#include "score_kernels.hpp"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

#ifndef KB_M
#define KB_M 20
#endif

using namespace lm;

__device__ __forceinline__ unsigned long long splitmix(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// rows x 32 random symbols in [0, nsym), then wrap rows (seq.rs:373-378)
__global__ void fill_seq(uint8_t *d, unsigned long long rows, unsigned long long wrap, int nsym,
                         int defsym)
{
    const unsigned long long n = (rows + wrap) * 32;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long r = i / 32, c = i % 32;
        if (r >= rows) {  // wrap row: data[rows+k][c] = data[k][c+1]
            r -= rows;
            c += 1;
        }
        d[i] = (c >= 32) ? defsym : (uint8_t)(splitmix(r * 32 + c) % nsym);
    }
}

__global__ void compare_bits(const unsigned *a, const unsigned *b, unsigned long long n,
                             unsigned long long *bad)
{
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        local += a[i] != b[i];
    if (local)
        atomicAdd(bad, local);
}

// stream copy with the score kernel's traffic mix: read n bytes, write 4n bytes
__global__ void mix_copy(const uint4 *in, float4 *out, unsigned long long n16)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 o;
            o.x = (float)(w[q] & 0xff);
            o.y = (float)((w[q] >> 8) & 0xff);
            o.z = (float)((w[q] >> 16) & 0xff);
            o.w = (float)(w[q] >> 24);
            out[i * 4 + q] = o;
        }
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void mix_copy_nt(const uint4 *in, f32x4 *out, unsigned long long n16)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o = {(float)(w[q] & 0xff), (float)((w[q] >> 8) & 0xff),
                       (float)((w[q] >> 16) & 0xff), (float)(w[q] >> 24)};
            __builtin_nontemporal_store(o, &out[i * 4 + q]);
        }
    }
}

// each thread converts ONE input dword -> one float4 (fully coalesced 16 B/lane stores)
__global__ void mix_copy_dw(const unsigned *in, f32x4 *out, unsigned long long n4, int nt)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned w = in[i];
        f32x4 o = {(float)(w & 0xff), (float)((w >> 8) & 0xff), (float)((w >> 16) & 0xff),
                   (float)(w >> 24)};
        if (nt)
            __builtin_nontemporal_store(o, &out[i]);
        else
            out[i] = o;
    }
}

__global__ void fill_nt(f32x4 *out, unsigned long long n16, float v)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        f32x4 o = {v, v, v, v};
        __builtin_nontemporal_store(o, &out[i]);
    }
}

__global__ void copy_f4(const float4 *in, float4 *out, unsigned long long n16)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * blockDim.x)
        out[i] = in[i];
}

__global__ void fill_f4(float4 *out, unsigned long long n16, float v)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * blockDim.x)
        out[i] = make_float4(v, v, v, v);
}

// VALU throughput probes: NI dependent-free adds per lane
template <int PK>
__global__ void valu_probe(float *out, int iters)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        a[i] = threadIdx.x * 0.001f + i;
    const float inc = out[0];
    for (int it = 0; it < iters; ++it) {
        if (PK) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                typedef float v2 __attribute__((ext_vector_type(2)));
                v2 x = {a[i], a[i + 1]};
                v2 y = {inc, inc};
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                a[i] = x.x;
                a[i + 1] = x.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(inc));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        s += a[i];
    if (s == 12345.678f)
        out[1] = s;
}

struct Variant {
    const char *name;
    ScoreC32Launcher fn;
};

template <int PF, int LP>
hipError_t launch_v(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq,
                    const float *table, int K, unsigned long long row_begin,
                    unsigned long long row_end, unsigned long long T,
                    unsigned long long nstreams, float *out, FusedOut fo)
{
    hipLaunchKernelGGL((score_c32<KB_M, MODE_STORE, PF, LP>), grid, dim3(kBlock), lds_bytes, stream,
                       seq, table, K, row_begin, row_end, T, nstreams, out, fo);
    return hipGetLastError();
}

template <int PF, int LP>
hipError_t launch_am(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq,
                     const float *table, int K, unsigned long long row_begin,
                     unsigned long long row_end, unsigned long long T,
                     unsigned long long nstreams, float *out, FusedOut fo)
{
    hipLaunchKernelGGL((score_c32<KB_M, MODE_ARGMAX, PF, LP>), grid, dim3(kBlock), lds_bytes,
                       stream, seq, table, K, row_begin, row_end, T, nstreams, out, fo);
    return hipGetLastError();
}

template <int PF, int BLK, int XCD, int MINW>
hipError_t launch_cfg(dim3 grid, size_t lds_bytes, hipStream_t stream, const uint8_t *seq,
                      const float *table, int K, unsigned long long row_begin,
                      unsigned long long row_end, unsigned long long T,
                      unsigned long long nstreams, float *out, FusedOut fo)
{
    hipLaunchKernelGGL((score_c32<KB_M, MODE_STORE, PF, 0, BLK, XCD, MINW>), grid, dim3(BLK),
                       lds_bytes, stream, seq, table, K, row_begin, row_end, T, nstreams, out, fo);
    return hipGetLastError();
}

struct Config {
    std::string name;
    ScoreC32Launcher fn;
    int blk;
    int q;
};

static double median(std::vector<float> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv)
{
    const unsigned long long L = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const int K = argc > 2 ? atoi(argv[2]) : 5;
    const int reps = argc > 3 ? atoi(argv[3]) : 15;
    const char *tag = argc > 4 ? argv[4] : "";
    const bool quick = argc > 5 && !strcmp(argv[5], "quick");
    constexpr int M = KB_M;
    const unsigned long long rows = (L + 31) / 32, wrap = M - 1;

    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s CUs=%d clock=%d MHz  L=%llu rows=%llu M=%d K=%d %s\n", prop.name,
           prop.multiProcessorCount, prop.clockRate / 1000, L, rows, M, K, tag);

    uint8_t *d_seq;
    float *d_ref, *d_out, *d_table, *d_dense;
    unsigned long long *d_bad;
    CK(hipMalloc(&d_seq, (rows + wrap) * 32 + 4096));
    CK(hipMalloc(&d_ref, rows * 32 * 4));
    CK(hipMalloc(&d_out, rows * 32 * 4));
    CK(hipMalloc(&d_bad, 8));
    hipLaunchKernelGGL(fill_seq, dim3(4096), dim3(256), 0, 0, d_seq, rows, wrap, K - 1, K - 1);

    // random log-odds-like PSSM with -inf in the last column
    std::vector<float> pssm(M * K), table(K * table_stride(M), 0.0f);
    srand(12345);
    for (int j = 0; j < M; ++j)
        for (int s = 0; s < K; ++s)
            pssm[j * K + s] = (s == K - 1) ? -INFINITY : (float)(rand() % 100000) / 7919.0f - 6.0f;
    for (int s = 0; s < K; ++s)
        for (int j = 0; j < M; ++j)
            table[s * table_stride(M) + j] = pssm[j * K + s];
    CK(hipMalloc(&d_table, table.size() * 4));
    CK(hipMalloc(&d_dense, pssm.size() * 4));
    CK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dense, pssm.data(), pssm.size() * 4, hipMemcpyHostToDevice));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    FusedOut fo{};

    // reference: generic kernel
    {
        const size_t lds = std::max<size_t>(M * K * 4, 64);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((score_generic<MODE_STORE>), dim3(8192), dim3(kBlock), lds, 0, d_seq,
                           32ull, 32, d_dense, M, K, 1, 0ull, rows, d_ref, 32ull, fo);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("generic            ms=%8.3f  Gpos/s=%8.1f\n", ms, rows * 32 / ms * 1e-6);
    }

    // HBM calibration with the same traffic mix
    if (!quick) {
        std::vector<float> t;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mix_copy, dim3(8192), dim3(256), 0, 0, (const uint4 *)d_seq,
                               (float4 *)d_out, rows * 2);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms);
        }
        const double ms = median(t);
        printf("mix_copy(1R+4W)    ms=%8.3f  GB/s=%8.1f\n", ms, rows * 32 * 5 / ms * 1e-6);
        t.clear();
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(fill_f4, dim3(8192), dim3(256), 0, 0, (float4 *)d_out, rows * 8,
                               1.0f);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms);
        }
        printf("fill(4W)           ms=%8.3f  GB/s=%8.1f\n", median(t), rows * 32 * 4 / median(t) * 1e-6);
        auto timeit = [&](const char *name, double bytes, auto &&launch) {
            std::vector<float> tt;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                tt.push_back(ms);
            }
            printf("%-18s ms=%8.3f  GB/s=%8.1f\n", name, median(tt), bytes / median(tt) * 1e-6);
        };
        for (int g : {2048, 8192, 32768}) {
            char nm[64];
            snprintf(nm, sizeof nm, "fill_nt g=%d", g);
            timeit(nm, rows * 32.0 * 4, [&] { hipLaunchKernelGGL(fill_nt, dim3(g), dim3(256), 0, 0, (f32x4 *)d_out, rows * 8, 1.0f); });
            snprintf(nm, sizeof nm, "mix_nt(1R+4W) g=%d", g);
            timeit(nm, rows * 32.0 * 5, [&] { hipLaunchKernelGGL(mix_copy_nt, dim3(g), dim3(256), 0, 0, (const uint4 *)d_seq, (f32x4 *)d_out, rows * 2); });
            snprintf(nm, sizeof nm, "mix_dw_nt g=%d", g);
            timeit(nm, rows * 32.0 * 5, [&] { hipLaunchKernelGGL(mix_copy_dw, dim3(g), dim3(256), 0, 0, (const unsigned *)d_seq, (f32x4 *)d_out, rows * 8, 1); });
            snprintf(nm, sizeof nm, "mix_dw_plain g=%d", g);
            timeit(nm, rows * 32.0 * 5, [&] { hipLaunchKernelGGL(mix_copy_dw, dim3(g), dim3(256), 0, 0, (const unsigned *)d_seq, (f32x4 *)d_out, rows * 8, 0); });
        }
        timeit("copy_f4(2R+2W GB)", rows * 32.0 * 4, [&] { hipLaunchKernelGGL(copy_f4, dim3(8192), dim3(256), 0, 0, (const float4 *)d_ref, (float4 *)d_out, rows * 4); });
    }

    // VALU probes
    for (int pk = 0; pk < 2 && !quick; ++pk) {
        const int iters = 4096;
        float *d_tmp = d_out;
        CK(hipMemset(d_tmp, 0, 64));
        CK(hipEventRecord(e0));
        if (pk)
            hipLaunchKernelGGL(valu_probe<1>, dim3(256 * 8), dim3(256), 0, 0, d_tmp, iters);
        else
            hipLaunchKernelGGL(valu_probe<0>, dim3(256 * 8), dim3(256), 0, 0, d_tmp, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double adds = 256.0 * 8 * 256 * iters * 16;
        printf("valu_probe pk=%d     ms=%8.3f  Tadd/s=%8.2f\n", pk, ms, adds / ms * 1e-9);
    }

    if (argc > 5 && !strcmp(argv[5], "place")) {
        // does the relative placement of the input and output buffers matter?
        char *arena;
        const size_t GB = 1ull << 30;
        CK(hipMalloc(&arena, 11 * GB));
        const size_t seq_bytes = (rows + wrap) * 32;
        const size_t lds2 = std::max<size_t>((size_t)K * table_stride(M) * 4, 64);
        struct P { const char *name; size_t seq_off, out_off; };
        const size_t adj = (seq_bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20);  // next 2 MiB
        std::vector<P> places = {
            {"out=seq+adj(2MiB aligned)", 0, adj},      {"out=seq+adj+4K", 0, adj + 4096},
            {"out=seq+adj+64K", 0, adj + 65536},        {"out=seq+adj+1M", 0, adj + (1u << 20)},
            {"out=seq+1GiB", 0, GB},                    {"out=seq+5GiB", 0, 5 * GB},
            {"out=seq+5GiB+adjrem", 0, 5 * GB + adj - GB + 0}, {"out first, seq after (+4GiB)", 4 * GB + (2u << 20), 0},
            {"seq+128B, out=seq+adj", 128, adj + (2u << 20)},
        };
        struct KC { const char *name; ScoreC32Launcher fn; int q; };
        std::vector<KC> kcs = {{"x0_T61", launch_cfg<12, 256, 0, 6>, 3}, {"x0_T121", launch_cfg<12, 256, 0, 6>, 6},
                               {"x1_T61", launch_cfg<12, 256, 1, 6>, 3}, {"x1_T121", launch_cfg<12, 256, 1, 6>, 6}};
        std::vector<std::vector<float>> times(places.size() * kcs.size());
        for (size_t pi = 0; pi < places.size(); ++pi)
            CK(hipMemcpy(arena + places[pi].seq_off, d_seq, seq_bytes, hipMemcpyDeviceToDevice));
        for (int r = 0; r < reps; ++r)
            for (size_t pi = 0; pi < places.size(); ++pi) {
                // (re)copy the sequence: placements overlap each other's buffers
                CK(hipMemcpy(arena + places[pi].seq_off, d_seq, seq_bytes, hipMemcpyDeviceToDevice));
                for (size_t ki = 0; ki < kcs.size(); ++ki) {
                    const unsigned long long T = (unsigned long long)kcs[ki].q * M + 1;
                    const unsigned long long ns = (rows + T - 1) / T;
                    const dim3 grid((unsigned)((ns + 7) / 8));
                    CK(hipEventRecord(e0));
                    CK(kcs[ki].fn(grid, lds2, 0, (const uint8_t *)(arena + places[pi].seq_off), d_table, K, 0,
                                  rows, T, ns, (float *)(arena + places[pi].out_off), fo));
                    CK(hipEventRecord(e1));
                    CK(hipDeviceSynchronize());
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    times[pi * kcs.size() + ki].push_back(ms);
                }
            }
        for (size_t pi = 0; pi < places.size(); ++pi) {
            printf("%-34s", places[pi].name);
            for (size_t ki = 0; ki < kcs.size(); ++ki) {
                auto &t = times[pi * kcs.size() + ki];
                std::sort(t.begin(), t.end());
                printf("  %s %.4f", kcs[ki].name, t[t.size() / 2]);
            }
            printf("\n");
        }
        return 0;
    }

    if (argc > 5 && !strcmp(argv[5], "ab")) {
        // interleaved A/B: every config once per round, R rounds, report median and min
        std::vector<Config> cfgs;
        const int qlist[] = {4, 6, 12, 20};
#define ADD(PF, BLK, XCD, MINW)                                                             \
    for (int q : qlist)                                                                      \
        cfgs.push_back({"pf" #PF "_b" #BLK "_x" #XCD "_w" #MINW, launch_cfg<PF, BLK, XCD, MINW>, BLK, q});
        ADD(12, 256, 0, 6) ADD(12, 256, 1, 6) ADD(12, 256, 1, 4)
        ADD(12, 128, 0, 6) ADD(12, 128, 1, 6) ADD(12, 128, 1, 4)
        ADD(12, 64, 0, 6) ADD(12, 64, 1, 6) ADD(12, 64, 1, 4) ADD(12, 64, 0, 4)
        ADD(16, 64, 1, 4) ADD(16, 128, 1, 4) ADD(16, 256, 1, 4)
#undef ADD
        const size_t lds2 = std::max<size_t>((size_t)K * table_stride(M) * 4, 64);
        std::vector<std::vector<float>> times(cfgs.size());
        std::vector<unsigned long long> bads(cfgs.size(), 0);
        const int rounds = reps;
        for (int r = -1; r < rounds; ++r) {
            for (size_t c = 0; c < cfgs.size(); ++c) {
                const unsigned long long T = (unsigned long long)cfgs[c].q * M + 1;
                const unsigned long long ns = (rows + T - 1) / T;
                const unsigned spb = cfgs[c].blk / 32;
                const dim3 grid((unsigned)((ns + spb - 1) / spb));
                if (r < 0) {  // correctness pass
                    CK(hipMemset(d_out, 0xff, rows * 32 * 4));
                    CK(cfgs[c].fn(grid, lds2, 0, d_seq, d_table, K, 0, rows, T, ns, d_out, fo));
                    CK(hipMemset(d_bad, 0, 8));
                    hipLaunchKernelGGL(compare_bits, dim3(4096), dim3(256), 0, 0,
                                       (const unsigned *)d_ref, (const unsigned *)d_out, rows * 32, d_bad);
                    CK(hipMemcpy(&bads[c], d_bad, 8, hipMemcpyDeviceToHost));
                    continue;
                }
                CK(hipEventRecord(e0));
                CK(cfgs[c].fn(grid, lds2, 0, d_seq, d_table, K, 0, rows, T, ns, d_out, fo));
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                times[c].push_back(ms);
            }
        }
        for (size_t c = 0; c < cfgs.size(); ++c) {
            std::sort(times[c].begin(), times[c].end());
            const double med = times[c][times[c].size() / 2];
            printf("%-20s T=%4d  med=%7.4f min=%7.4f  GB/s=%7.1f  mismatches=%llu\n",
                   cfgs[c].name.c_str(), cfgs[c].q * M + 1, med, times[c][0],
                   rows * 32 * 5 / med * 1e-6, bads[c]);
        }
        return 0;
    }

    std::vector<Variant> vars = {
        {"pf8_lp0", launch_v<8, 0>},   {"pf12_lp0", launch_v<12, 0>}, {"pf16_lp0", launch_v<16, 0>},
        {"pf19_lp0", launch_v<19, 0>}, {"pf12_lp1", launch_v<12, 1>}, {"pf16_lp1", launch_v<16, 1>},
    };
    if (quick)
        vars = {{"pf12_lp0", launch_v<12, 0>}, {"pf16_lp0", launch_v<16, 0>}, {"pf8_lp0", launch_v<8, 0>}};
    std::vector<int> qs = {1, 2, 3, 4, 5, 6, 8, 12, 25};
    if (quick)
        qs = {2, 3, 4, 6, 12, 50};
    if (argc > 6) {
        qs.clear();
        for (char *tok = strtok(argv[6], ","); tok; tok = strtok(nullptr, ","))
            qs.push_back(atoi(tok));
    }
    const size_t lds = std::max<size_t>((size_t)K * table_stride(M) * 4, 64);
    for (auto &v : vars) {
        for (int q : qs) {
            const unsigned long long T = (unsigned long long)q * M + 1;
            if (T > rows)
                continue;
            const unsigned long long ns = (rows + T - 1) / T;
            const dim3 grid((unsigned)((ns + kStreamsPerBlock - 1) / kStreamsPerBlock));
            CK(hipMemset(d_out, 0xff, rows * 32 * 4));
            CK(v.fn(grid, lds, 0, d_seq, d_table, K, 0, rows, T, ns, d_out, fo));
            CK(hipMemset(d_bad, 0, 8));
            hipLaunchKernelGGL(compare_bits, dim3(4096), dim3(256), 0, 0, (const unsigned *)d_ref,
                               (const unsigned *)d_out, rows * 32, d_bad);
            unsigned long long bad = 0;
            CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
            std::vector<float> t;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                CK(v.fn(grid, lds, 0, d_seq, d_table, K, 0, rows, T, ns, d_out, fo));
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            const double ms = t[t.size() / 2];
            printf("%-10s T=%5llu grid=%6u  ms=%8.3f (min %7.3f)  Gpos/s=%8.1f  GB/s=%8.1f  mismatches=%llu\n",
                   v.name, T, grid.x, ms, t[0], rows * 32 / ms * 1e-6, rows * 32 * 5 / ms * 1e-6,
                   bad);
            fflush(stdout);
        }
    }

    // fused argmax variants (no store): LDS/VALU-bound
    if (!quick) {
        ArgmaxRecord *d_rec;
        CK(hipMalloc(&d_rec, sizeof(ArgmaxRecord) * 1000000));
        fo.block_best = d_rec;
        Variant avars[] = {{"am_pf6_lp0", launch_am<6, 0>}, {"am_pf6_lp1", launch_am<6, 1>}};
        for (auto &v : avars)
            for (int q : {12, 50, 200}) {
                const unsigned long long T = (unsigned long long)q * M + 1;
                if (T > rows)
                    continue;
                const unsigned long long ns = (rows + T - 1) / T;
                const dim3 grid((unsigned)((ns + kStreamsPerBlock - 1) / kStreamsPerBlock));
                std::vector<float> t;
                for (int r = 0; r < reps; ++r) {
                    CK(hipEventRecord(e0));
                    CK(v.fn(grid, lds, 0, d_seq, d_table, K, 0, rows, T, ns, nullptr, fo));
                    CK(hipEventRecord(e1));
                    CK(hipDeviceSynchronize());
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    t.push_back(ms);
                }
                std::sort(t.begin(), t.end());
                printf("%-10s T=%5llu grid=%6u  ms=%8.3f (min %7.3f)  Gpos/s=%8.1f\n", v.name, T,
                       grid.x, t[t.size() / 2], t[0], rows * 32 / t[t.size() / 2] * 1e-6);
            }
    }
    return 0;
}
