// h2d_bench -- how fast can 1 GB of a caller's pageable buffer reach the device?  (ingest path, DESIGN "ingest")
//   1 pageable hipMemcpy, whole buffer            2 pageable hipMemcpy2D tiles (32 pieces per tile, pitch R)
//   3 hipHostRegister + hipMemcpy + unregister    4 N host threads stage into pinned double buffers + hipMemcpyAsync
//   5 kernel reads the registered buffer directly (zero copy)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void sum_kernel(const uint4 *p, size_t n, unsigned *out)
{
    unsigned s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        s += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345678u) *out = s;
}
int main(int argc, char **argv)
{
    const size_t L = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ull;
    const int nthreads = argc > 2 ? atoi(argv[2]) : 8;
    uint8_t *h = (uint8_t *)malloc(L);
    for (size_t i = 0; i < L; i += 4096) h[i] = (uint8_t)i;  // touch
    memset(h, 'A', L);
    uint8_t *d; CK(hipMalloc(&d, L + (64 << 20)));
    unsigned *d_out; CK(hipMalloc(&d_out, 4));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); CK(hipMemcpy(d, h, L, hipMemcpyHostToDevice)); double t = now() - t0;
        printf("1 pageable hipMemcpy %zu B: %.1f ms = %.1f GB/s\n", L, t * 1e3, L / t / 1e9);
    }
    {   // tiles of 32 pieces
        const size_t R = L / 32, TR = 1 << 20;
        double t0 = now();
        for (size_t k = 0; k * TR < R; ++k) {
            const size_t w = R - k * TR < TR ? R - k * TR : TR;
            CK(hipMemcpy2DAsync(d + (k & 1) * 32 * TR, TR, h + k * TR, R, w, 32, hipMemcpyHostToDevice, s1));
        }
        CK(hipStreamSynchronize(s1));
        double t = now() - t0;
        printf("2 pageable hipMemcpy2DAsync tiles (1 Mi rows x 32): %.1f ms = %.1f GB/s\n", t * 1e3, L / t / 1e9);
    }
    {
        double t0 = now(); CK(hipHostRegister(h, L, hipHostRegisterDefault)); double tr = now() - t0;
        t0 = now(); CK(hipMemcpy(d, h, L, hipMemcpyHostToDevice)); double tc = now() - t0;
        t0 = now();
        {   const size_t R = L / 32, TR = 1 << 20;
            for (size_t k = 0; k * TR < R; ++k) {
                const size_t w = R - k * TR < TR ? R - k * TR : TR;
                CK(hipMemcpy2DAsync(d + (k & 1) * 32 * TR, TR, h + k * TR, R, w, 32, hipMemcpyHostToDevice, s1));
            }
            CK(hipStreamSynchronize(s1)); }
        double t2 = now() - t0;
        t0 = now();
        hipLaunchKernelGGL(sum_kernel, dim3(2048), dim3(256), 0, s1, (const uint4 *)h, L / 16, d_out);
        CK(hipStreamSynchronize(s1));
        double tk = now() - t0;
        t0 = now(); CK(hipHostUnregister(h)); double tu = now() - t0;
        printf("3 hipHostRegister %.1f ms, copy %.1f ms (%.1f GB/s), 2D tiles from registered %.1f ms (%.1f GB/s), unregister %.1f ms\n",
               tr * 1e3, tc * 1e3, L / tc / 1e9, t2 * 1e3, L / t2 / 1e9, tu * 1e3);
        printf("5 kernel reading registered host memory: %.1f ms = %.1f GB/s\n", tk * 1e3, L / tk / 1e9);
    }
    for (int nt : {1, 2, 4, nthreads, 16}) {
        const size_t R = L / 32, TR = 1 << 19;  // tile = 32 x 512 Ki = 16 MB
        uint8_t *pin[2]; CK(hipHostMalloc((void **)&pin[0], 32 * TR)); CK(hipHostMalloc((void **)&pin[1], 32 * TR));
        hipEvent_t done[2]; CK(hipEventCreate(&done[0])); CK(hipEventCreate(&done[1]));
        double t0 = now();
        const size_t ntiles = (R + TR - 1) / TR;
        for (size_t k = 0; k < ntiles; ++k) {
            const int b = k & 1;
            if (k >= 2) CK(hipEventSynchronize(done[b]));
            const size_t w = R - k * TR < TR ? R - k * TR : TR;
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&, t] { for (int c = t; c < 32; c += nt) memcpy(pin[b] + c * TR, h + c * R + k * TR, w); });
            for (auto &x : th) x.join();
            CK(hipMemcpyAsync(d + b * 32 * TR, pin[b], 32 * TR, hipMemcpyHostToDevice, s1));
            CK(hipEventRecord(done[b], s1));
        }
        CK(hipStreamSynchronize(s1));
        double t = now() - t0;
        printf("4 staged through pinned double buffers, %d host threads: %.1f ms = %.1f GB/s\n", nt, t * 1e3, L / t / 1e9);
        CK(hipHostFree(pin[0])); CK(hipHostFree(pin[1]));
    }
    return 0;
}
