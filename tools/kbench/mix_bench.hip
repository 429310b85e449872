// mix_bench -- how fast can a trivial kernel move the score kernel's traffic mix (1 B read : 4 B written)?
// Development tool.  ./mix_bench [bytes_in]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 widen(unsigned w)
{
    f32x4 o = {(float)(w & 0xff), (float)((w >> 8) & 0xff), (float)((w >> 16) & 0xff), (float)(w >> 24)};
    return o;
}

// dword read -> float4 write, grid-stride (the earlier best: mix_dw_nt)
__global__ __launch_bounds__(256) void mix_dw(const unsigned *in, f32x4 *out, unsigned long long n4)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4;
         i += (unsigned long long)gridDim.x * 256)
        __builtin_nontemporal_store(widen(in[i]), &out[i]);
}

// 16-byte reads staged through LDS: a wavefront reads 1 KB per load and writes 1 KB per store
template <int CHUNKS>
__global__ __launch_bounds__(256) void mix_lds(const uint4 *in, f32x4 *out, unsigned long long n16)
{
    __shared__ uint4 tile[256];
    const unsigned *td = reinterpret_cast<const unsigned *>(tile);
    for (unsigned long long base = (unsigned long long)blockIdx.x * 256 * CHUNKS; base < n16;
         base += (unsigned long long)gridDim.x * 256 * CHUNKS) {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const unsigned long long i = base + c * 256 + threadIdx.x;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i < n16)
                v = in[i];
            __syncthreads();
            tile[threadIdx.x] = v;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long o = (base + c * 256) * 4 + q * 256 + threadIdx.x;
                if (o < n16 * 4)
                    __builtin_nontemporal_store(widen(td[q * 256 + threadIdx.x]), &out[o]);
            }
        }
    }
}

// all reads of a block first (registers), then all writes: longer read and write bursts
template <int R>
__global__ __launch_bounds__(256) void mix_burst(const unsigned *in, f32x4 *out, unsigned long long n4)
{
    for (unsigned long long base = (unsigned long long)blockIdx.x * 256 * R; base < n4;
         base += (unsigned long long)gridDim.x * 256 * R) {
        unsigned w[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned long long i = base + r * 256 + threadIdx.x;
            w[r] = i < n4 ? in[i] : 0;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned long long i = base + r * 256 + threadIdx.x;
            if (i < n4)
                __builtin_nontemporal_store(widen(w[r]), &out[i]);
        }
    }
}

__global__ __launch_bounds__(256) void fill(f32x4 *out, unsigned long long n16)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * 256) {
        f32x4 o = {1, 2, 3, 4};
        __builtin_nontemporal_store(o, &out[i]);
    }
}

template <typename F>
static float timeit(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const unsigned long long n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    uint8_t *in; float *out;
    CK(hipMalloc(&in, n + 64));
    CK(hipMalloc(&out, n * 4 + 64));
    CK(hipMemset(in, 3, n));
    auto rep = [&](const char *name, float ms) { printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, 5.0 * n / ms / 1e6); };
    for (int g : {8192, 32768, 131072}) {
        char nm[64];
        snprintf(nm, sizeof nm, "mix_dw g=%d", g);
        rep(nm, timeit([&] { hipLaunchKernelGGL(mix_dw, dim3(g), dim3(256), 0, 0, (const unsigned *)in, (f32x4 *)out, n / 4); }, 20));
        snprintf(nm, sizeof nm, "mix_lds<1> g=%d", g);
        rep(nm, timeit([&] { hipLaunchKernelGGL(mix_lds<1>, dim3(g), dim3(256), 0, 0, (const uint4 *)in, (f32x4 *)out, n / 16); }, 20));
        snprintf(nm, sizeof nm, "mix_lds<4> g=%d", g);
        rep(nm, timeit([&] { hipLaunchKernelGGL(mix_lds<4>, dim3(g), dim3(256), 0, 0, (const uint4 *)in, (f32x4 *)out, n / 16); }, 20));
        snprintf(nm, sizeof nm, "mix_burst<4> g=%d", g);
        rep(nm, timeit([&] { hipLaunchKernelGGL(mix_burst<4>, dim3(g), dim3(256), 0, 0, (const unsigned *)in, (f32x4 *)out, n / 4); }, 20));
        snprintf(nm, sizeof nm, "mix_burst<16> g=%d", g);
        rep(nm, timeit([&] { hipLaunchKernelGGL(mix_burst<16>, dim3(g), dim3(256), 0, 0, (const unsigned *)in, (f32x4 *)out, n / 4); }, 20));
        snprintf(nm, sizeof nm, "fill(4W) g=%d", g);
        float ms = timeit([&] { hipLaunchKernelGGL(fill, dim3(g), dim3(256), 0, 0, (f32x4 *)out, n / 4); }, 20);
        printf("%-28s %8.3f ms  %7.1f GB/s\n", nm, ms, 4.0 * n / ms / 1e6);
    }
    return 0;
}
