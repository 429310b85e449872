// valu_bench -- issue rate of v_add_f32 vs v_pk_add_f32 vs v_pk_add_u16 on one GPU (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr int N = 16;      // independent accumulators per lane
constexpr int ITER = 4096;

__global__ __launch_bounds__(256) void k_add(float *out, float seed)
{
    float a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = seed + i + threadIdx.x;
    const float w = seed * 0.5f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk_add(float *out, float seed)
{
    f2 a[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) a[i] = (f2){seed + i + threadIdx.x, seed - i};
    const f2 w = (f2){seed * 0.5f, seed * 0.25f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk_u16(float *out, float seed)
{
    unsigned a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = (unsigned)seed + i + threadIdx.x;
    const unsigned w = 0x00010002u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(w));
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

template <typename K>
static void run(const char *name, K kernel, double ops_per_instr, int instr_per_iter)
{
    float *out;
    const int blocks = 256 * 32;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double lane_instr = (double)blocks * 256 * ITER * instr_per_iter;
    printf("%-14s %8.3f ms  %7.2f T lane-instr/s  %7.2f T elementary ops/s\n", name, ms,
           lane_instr / ms / 1e9, lane_instr * ops_per_instr / ms / 1e9);
    CK(hipFree(out));
}

int main()
{
    run("v_add_f32", k_add, 1, N);
    run("v_pk_add_f32", k_pk_add, 2, N / 2);
    run("v_pk_add_u16", k_pk_u16, 2, N);
    return 0;
}
