// stripe_bench -- variants of the C = 32 stripe (transpose) kernel, to find what bounds it.
// Development tool (not part of the product library).  Build: hipcc --offload-arch=gfx950 -O3
// stripe_bench.hip -o stripe_bench ; run: ./stripe_bench [length]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

constexpr int kBlock = 256;

// VAR 0: full; 1: no LDS reads in phase 2 (stores constants); 2: no global loads
template <int VAR, int TR>
__global__ __launch_bounds__(kBlock) void stripe_v(const uint8_t *__restrict__ enc,
                                                   const unsigned long long len,
                                                   const unsigned long long rows,
                                                   uint8_t *__restrict__ data)
{
    __shared__ unsigned tile[32][TR / 4];
    const unsigned long long r0 = (unsigned long long)blockIdx.x * TR;
    for (int sub = 0; sub < TR / (4 * kBlock); ++sub) {
        const unsigned t = sub * kBlock + threadIdx.x;
        const unsigned long long r = r0 + 4ull * t;
#pragma unroll 8
        for (unsigned c = 0; c < 32; ++c) {
            const unsigned long long i = (unsigned long long)c * rows + r;
            unsigned v = c;
            if (VAR != 2 && r + 3 < rows && i + 3 < len)
                __builtin_memcpy(&v, enc + i, 4);
            tile[c][t] = v;
        }
    }
    __syncthreads();
    const uint8_t *tb = reinterpret_cast<const uint8_t *>(&tile[0][0]);
#pragma unroll 1
    for (int it = 0; it < TR / kBlock; ++it) {
        const unsigned lr = it * kBlock + threadIdx.x;
        const unsigned long long row = r0 + lr;
        if (row < rows) {
            unsigned w[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (VAR == 1) {
                    w[g] = lr + g;
                } else {
                    const unsigned b0 = tb[(4 * g + 0) * TR + lr], b1 = tb[(4 * g + 1) * TR + lr];
                    const unsigned b2 = tb[(4 * g + 2) * TR + lr], b3 = tb[(4 * g + 3) * TR + lr];
                    w[g] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                }
            }
            uint4 *dst = reinterpret_cast<uint4 *>(data + row * 32);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
}

// 16-byte loads: a wavefront reads 1 KB of one column per instruction; phase 2 reads
// dwords (4 rows x 1 column) and transposes 4x4 byte blocks in registers
template <int TR>
__global__ __launch_bounds__(kBlock) void stripe_w(const uint8_t *__restrict__ enc,
                                                   const unsigned long long len,
                                                   const unsigned long long rows,
                                                   uint8_t *__restrict__ data)
{
    __shared__ uint4 tile[32][TR / 16];
    const unsigned long long r0 = (unsigned long long)blockIdx.x * TR;
    constexpr int per_col = TR / 16;            // 16-byte pieces per column
    for (unsigned p = threadIdx.x; p < 32 * per_col; p += kBlock) {
        const unsigned c = p / per_col, q = p % per_col;
        const unsigned long long r = r0 + 16ull * q;
        const unsigned long long i = (unsigned long long)c * rows + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r + 15 < rows && i + 15 < len)
            __builtin_memcpy(&v, enc + i, 16);
        tile[c][q] = v;
    }
    __syncthreads();
    const uint8_t *tb = reinterpret_cast<const uint8_t *>(&tile[0][0]);
#pragma unroll 1
    for (int it = 0; it < TR / kBlock; ++it) {
        const unsigned lr = it * kBlock + threadIdx.x;
        const unsigned long long row = r0 + lr;
        if (row < rows) {
            unsigned w[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const unsigned b0 = tb[(4 * g + 0) * TR + lr], b1 = tb[(4 * g + 1) * TR + lr];
                const unsigned b2 = tb[(4 * g + 2) * TR + lr], b3 = tb[(4 * g + 3) * TR + lr];
                w[g] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            }
            uint4 *dst = reinterpret_cast<uint4 *>(data + row * 32);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
}

__global__ void copy16(const uint4 *__restrict__ a, uint4 *__restrict__ b, unsigned long long n)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        b[i] = a[i];
}

template <typename F>
static float timeit(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i)
        f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i)
        f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const unsigned long long len = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const unsigned long long rows = (len + 31) / 32;
    uint8_t *enc, *data;
    CK(hipMalloc(&enc, len + 64));
    CK(hipMalloc(&data, rows * 32 + 64));
    CK(hipMemset(enc, 1, len + 64));
    const int reps = 30;
    auto report = [&](const char *name, float ms) {
        printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, 2.0 * len / ms / 1e6);
    };
#define RUN(NAME, KERNEL, TR)                                                                      \
    report(NAME, timeit([&] {                                                                      \
               hipLaunchKernelGGL(KERNEL, dim3((unsigned)((rows + TR - 1) / TR)), dim3(kBlock), 0, 0, \
                                  enc, len, rows, data);                                           \
           }, reps));
    RUN("full TR=1024", (stripe_v<0, 1024>), 1024)
    RUN("no-LDS-read TR=1024", (stripe_v<1, 1024>), 1024)
    RUN("no-global-load TR=1024", (stripe_v<2, 1024>), 1024)
    RUN("full TR=2048", (stripe_v<0, 2048>), 2048)
    RUN("full TR=512 (blk256 -> n/a)", (stripe_v<0, 1024>), 1024)
    RUN("16B loads TR=1024", (stripe_w<1024>), 1024)
    RUN("16B loads TR=2048", (stripe_w<2048>), 2048)
    report("copy16 1R:1W", timeit([&] {
               hipLaunchKernelGGL(copy16, dim3(16384), dim3(256), 0, 0, (const uint4 *)enc, (uint4 *)data,
                                  len / 16);
           }, reps));
    CK(hipDeviceSynchronize());
    return 0;
}
