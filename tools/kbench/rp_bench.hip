// rp_bench -- score_c32_rp<M> (row-pair wavefronts) against score_c32<M,0> and the generic kernel:
// bit-exactness and an interleaved timing sweep over the region length.  Development tool.
//   hipcc ... -DKB_M=20 rp_bench.hip -o rp_bench_20 ;  ./rp_bench_20 [L] [K] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "score_rowpair.hpp"  // tools/kbench/ (experiment)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
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

__global__ void fill_seq(uint8_t *d, unsigned long long rows, unsigned long long wrap, int nsym, int defsym)
{
    const unsigned long long n = (rows + wrap) * 32;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long r = i / 32, c = i % 32;
        if (r >= rows) { r -= rows; c += 1; }
        d[i] = (c >= 32) ? defsym : (uint8_t)(splitmix(r * 32 + c) % nsym);
    }
}

__global__ void compare_bits(const unsigned *a, const unsigned *b, unsigned long long n, unsigned long long *bad)
{
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        local += a[i] != b[i];
    if (local) atomicAdd(bad, local);
}

struct Cfg { std::string name; int kind; unsigned long long T; };  // kind 0 = score_c32, 1 = rp, 2 = rp byte loads

int main(int argc, char **argv)
{
    const unsigned long long L = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const int K = argc > 2 ? atoi(argv[2]) : 5;
    const int rounds = argc > 3 ? atoi(argv[3]) : 15;
    constexpr int M = KB_M;
    const unsigned long long rows = (L + 31) / 32, wrap = M - 1;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s  L=%llu rows=%llu M=%d K=%d  rp: H=%d G=%d TS=%d last=%d ql=%d pfb=%d\n", prop.name, L, rows, M, K,
           rp_slots(M), rp_group(M), rp_table_stride(M), rp_last_steps(M), (int)rp_quad_loads(M), rp_pf_blocks(M));
    uint8_t *d_seq; float *d_ref, *d_out, *d_table, *d_rp, *d_dense; unsigned long long *d_bad;
    // EXACTLY rows + wrap rows, at the end of the allocation granule as far as possible: reads past
    // the last wrap row would at least hit the guard pattern check below
    CK(hipMalloc(&d_seq, (rows + wrap) * 32));
    CK(hipMalloc(&d_ref, rows * 32 * 4));
    CK(hipMalloc(&d_out, rows * 32 * 4));
    CK(hipMalloc(&d_bad, 8));
    hipLaunchKernelGGL(fill_seq, dim3(4096), dim3(256), 0, 0, d_seq, rows, wrap, K - 1, K - 1);
    std::vector<float> pssm(M * K), table(K * table_stride(M), 0.0f), rp(rp_table_floats(M, K));
    srand(12345);
    for (int j = 0; j < M; ++j)
        for (int s = 0; s < K; ++s)
            pssm[j * K + s] = (s == K - 1) ? -INFINITY : (float)(rand() % 100000) / 7919.0f - 6.0f;
    for (int s = 0; s < K; ++s)
        for (int j = 0; j < M; ++j)
            table[s * table_stride(M) + j] = pssm[j * K + s];
    rp_build_table(pssm.data(), M, K, rp.data());
    CK(hipMalloc(&d_table, table.size() * 4)); CK(hipMalloc(&d_rp, rp.size() * 4)); CK(hipMalloc(&d_dense, pssm.size() * 4));
    CK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dense, pssm.data(), pssm.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    FusedOut fo{};
    hipLaunchKernelGGL((score_generic<MODE_STORE>), dim3(8192), dim3(kBlock), std::max<size_t>(M * K * 4, 64), 0, d_seq,
                       32ull, 32, d_dense, M, K, 1, 0ull, rows, d_ref, 32ull, fo);
    CK(hipDeviceSynchronize());

    std::vector<Cfg> cfgs;
    if (M <= kMaxFastM)
        for (unsigned long long q : {2ull, 3ull, 4ull, 6ull, 10ull, 16ull})
            cfgs.push_back({"c32      T=" + std::to_string(q * M + 1), 0, q * M + 1});
    const int G = rp_group(M);
    for (unsigned long long R : {64ull, 128ull, 192ull, 256ull, 320ull, 512ull, 1024ull}) {
        const unsigned long long q = std::max<unsigned long long>((R + G / 2) / G, 1);
        cfgs.push_back({"rp       R=" + std::to_string(q * G), 1, q * G});
    }
    cfgs.push_back({"rp_bytes R=" + std::to_string((128 / G) * G), 2, (unsigned long long)(128 / G) * G});
    for (unsigned long long R : {128ull, 256ull}) {
        const unsigned long long q = std::max<unsigned long long>((R + G / 2) / G, 1);
        cfgs.push_back({"rp_w7    R=" + std::to_string(q * G), 3, q * G});
        cfgs.push_back({"rp_w6    R=" + std::to_string(q * G), 4, q * G});
    }

    auto launch = [&](const Cfg &c) {
        if (c.kind == 0) {
            const unsigned long long ns = (rows + c.T - 1) / c.T;
            const dim3 grid((unsigned)((ns + 7) / 8));
            const size_t lds = std::max<size_t>((size_t)K * table_stride(M) * 4, 64);
            if constexpr (M <= kMaxFastM)
                hipLaunchKernelGGL((score_c32<M, MODE_STORE, LM_SCORE_PF, 0, 256, 0, LM_SCORE_MIN_WAVES(M), 1>), grid, dim3(256),
                                   lds, 0, d_seq, d_table, K, 0ull, rows, c.T, ns, d_out, fo);
        } else {
            const unsigned long long nr = (rows + c.T - 1) / c.T;
            const dim3 grid((unsigned)((nr + 3) / 4));
            const size_t lds = rp_table_floats(M, K) * 4;
            if (c.kind == 1)
                hipLaunchKernelGGL((score_c32_rp<M, MODE_STORE, 1>), grid, dim3(256), lds, 0, d_seq, d_rp, K, 0ull, rows, c.T, nr, d_out, fo);
            else if (c.kind == 2)
                hipLaunchKernelGGL((score_c32_rp<M, MODE_STORE, 0>), grid, dim3(256), lds, 0, d_seq, d_rp, K, 0ull, rows, c.T, nr, d_out, fo);
            else if (c.kind == 3)
                hipLaunchKernelGGL((score_c32_rp<M, MODE_STORE, 1, 12, 256, 7>), grid, dim3(256), lds, 0, d_seq, d_rp, K, 0ull, rows, c.T, nr, d_out, fo);
            else
                hipLaunchKernelGGL((score_c32_rp<M, MODE_STORE, 1, 12, 256, 6>), grid, dim3(256), lds, 0, d_seq, d_rp, K, 0ull, rows, c.T, nr, d_out, fo);
        }
        CK(hipGetLastError());
    };
    if (argc > 4 && !strcmp(argv[4], "c32only"))
        cfgs.erase(std::remove_if(cfgs.begin(), cfgs.end(), [](const Cfg &c) { return c.kind != 0; }), cfgs.end());
    std::vector<std::vector<float>> times(cfgs.size());
    std::vector<unsigned long long> bads(cfgs.size(), 0);
    for (int r = -1; r < rounds; ++r)
        for (size_t c = 0; c < cfgs.size(); ++c) {
            if (cfgs[c].T > rows) continue;
            if (r < 0) {
                CK(hipMemset(d_out, 0xff, rows * 32 * 4));
                launch(cfgs[c]);
                CK(hipMemset(d_bad, 0, 8));
                hipLaunchKernelGGL(compare_bits, dim3(4096), dim3(256), 0, 0, (const unsigned *)d_ref, (const unsigned *)d_out, rows * 32, d_bad);
                CK(hipMemcpy(&bads[c], d_bad, 8, hipMemcpyDeviceToHost));
                for (int w = 0; w < 20; ++w) launch(cfgs[c]);  // warm
                continue;
            }
            CK(hipEventRecord(e0));
            launch(cfgs[c]);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            times[c].push_back(ms);
        }
    for (size_t c = 0; c < cfgs.size(); ++c) {
        if (times[c].empty()) continue;
        std::sort(times[c].begin(), times[c].end());
        const double med = times[c][times[c].size() / 2];
        printf("%-22s med=%7.4f min=%7.4f ms  %7.1f GB/s  %6.1f Gpos/s  mismatches=%llu\n", cfgs[c].name.c_str(), med,
               times[c][0], rows * 32 * 5 / med * 1e-6, rows * 32 / med * 1e-6, bads[c]);
    }
    return 0;
}
