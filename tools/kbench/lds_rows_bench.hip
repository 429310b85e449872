// lds_rows_bench -- what a 16-byte LDS gather costs when the lanes of a wavefront address R distinct table rows:
// R = 5 / 16 (DNA symbols / pairs without N), 21 (protein symbols), 25 (DNA pairs), 441 (protein PAIRS).
// Decides whether a pair-symbol prefilter for protein (441 rows, half the reads) can beat the one-symbol
// prefilter (21 rows).  Development tool:  ./lds_rows_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// every lane reads NREAD consecutive 16-byte chunks of a row picked per (lane, iteration) from `rows`
template <int NREAD>
__global__ __launch_bounds__(256) void gather(const unsigned *__restrict__ rows, int nidx, int R, int stride_b, int iters, unsigned *out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < R * stride_b / 4; i += 256)
        reinterpret_cast<unsigned *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    unsigned acc[4] = {0, 0, 0, 0};
    unsigned idx = rows[(blockIdx.x * 256 + threadIdx.x) % (unsigned)nidx] * 2654435761u + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        idx = idx * 1664525u + 1013904223u;                      // per-lane LCG: uniform rows, no memory traffic
        const unsigned r = __umulhi(idx, (unsigned)R);
        const char *row = lds + r * stride_b;
#pragma unroll
        for (int q = 0; q < NREAD; ++q) {
            const uint4 v = *reinterpret_cast<const uint4 *>(row + 16 * q);
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

int main()
{
    const int blocks = 256 * 8, iters = 2000, nidx = 1 << 20;
    unsigned *d_rows, *d_out;
    CK(hipMalloc(&d_rows, nidx * 4)); CK(hipMalloc(&d_out, blocks * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct Case { const char *name; int R, stride_b, nread; };
    // strides: 4 * odd dwords >= the row's dwords (as the library lays its tables out)
    const Case cases[] = {{"DNA symbols, 5 rows x 48 B, 2 reads (one-symbol prefilter M'=12.. style)", 5, 48, 2},
                          {"DNA pairs, 16 rows x 48 B, 2 reads", 16, 48, 2},
                          {"DNA pairs incl. N, 25 rows x 48 B, 2 reads", 25, 48, 2},
                          {"protein symbols, 21 rows x 48 B, 2 reads (one-symbol prefilter, M = 12: 24 B/pos -> 2 reads per position)", 21, 48, 2},
                          {"protein PAIRS, 441 rows x 48 B, 2 reads (pair prefilter, M' = 15: 32 B per TWO positions)", 441, 48, 2},
                          {"protein symbols, 21 rows x 80 B, 4 reads (M ~ 28)", 21, 80, 4},
                          {"protein PAIRS, 441 rows x 80 B, 4 reads (M' = 31)", 441, 80, 4}};
    for (const Case &c : cases) {
        std::vector<unsigned> h(nidx);
        srand(1);
        for (int i = 0; i < nidx; ++i) h[i] = (unsigned)(rand() % c.R);
        CK(hipMemcpy(d_rows, h.data(), nidx * 4, hipMemcpyHostToDevice));
        const size_t lds = (size_t)c.R * c.stride_b;
        auto run = [&] {
            if (c.nread == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), lds, 0, d_rows, nidx, c.R, c.stride_b, iters, d_out);
            else hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), lds, 0, d_rows, nidx, c.R, c.stride_b, iters, d_out);
        };
        run(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); run(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double reads = (double)blocks * 4 * iters * c.nread;   // wave-level 16-byte read instructions
        printf("%-112s %7.3f ms  %6.2f ns per wave-read per CU\n", c.name, ms, ms * 1e6 / (reads / 256.0));
    }
    return 0;
}
