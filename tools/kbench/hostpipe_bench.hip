// hostpipe_bench -- what the host-pointer entry points (lm_hip_score_f32 & co.) can reach over PCIe, measured before the
// pipeline was designed (DESIGN "host-pointer path").  Pageable caller buffers, as the Rust shim hands them over.
//   small (the reference's bench loop, 464 165 bp; a Scanner block, 256 rows): latency of each way to move the bytes
//   large (1 Gbp): D2H rates, whether pageable async copies return early, full-duplex from one / two host threads
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> static double med_us(int reps, F f)
{
    std::vector<double> t;
    for (int i = 0; i < reps + 3; ++i) {
        double t0 = now();
        f();
        if (i >= 3) t.push_back(now() - t0);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e6;
}

__global__ void fill_kernel(float *p, size_t n, float v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v + (float)(i & 1023);
}
__global__ void read_kernel(const uint4 *p, size_t n, unsigned *out)
{
    unsigned s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        s += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345678u) *out = s;
}

static void par_memcpy(void *dst, const void *src, size_t n, int nt)
{
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t piece = (n / nt + 4095) / 4096 * 4096;
    for (int t = 0; t < nt; ++t) {
        const size_t lo = std::min(n, t * piece), hi = std::min(n, lo + piece);
        if (lo < hi) th.emplace_back([=] { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    for (auto &x : th) x.join();
}

int main(int argc, char **argv)
{
    const size_t L = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ull;
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    unsigned *d_flag; CK(hipMalloc(&d_flag, 4));
    printf("host threads: %u\n", std::thread::hardware_concurrency());

    // ---------------- small sizes ----------------
    for (size_t in_bytes : {(size_t)8192 + 19 * 32, (size_t)479 * 1024, (size_t)4 << 20}) {
        const size_t out_bytes = in_bytes * 4;
        uint8_t *h_in = (uint8_t *)malloc(in_bytes), *h_out = (uint8_t *)malloc(out_bytes);
        memset(h_in, 1, in_bytes); memset(h_out, 2, out_bytes);
        uint8_t *d_in, *d_out, *p_in, *p_out;
        CK(hipMalloc(&d_in, in_bytes)); CK(hipMalloc(&d_out, out_bytes));
        CK(hipHostMalloc((void **)&p_in, in_bytes)); CK(hipHostMalloc((void **)&p_out, out_bytes));
        printf("--- in %zu B, out %zu B\n", in_bytes, out_bytes);
        printf("H2D in : pageable async+sync %.1f us | staged (memcpy+async+sync) %.1f us | pinned only %.1f us\n",
               med_us(50, [&] { CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }),
               med_us(50, [&] { memcpy(p_in, h_in, in_bytes); CK(hipMemcpyAsync(d_in, p_in, in_bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }),
               med_us(50, [&] { CK(hipMemcpyAsync(d_in, p_in, in_bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }));
        printf("D2H out: pageable async+sync %.1f us | staged (async+sync+memcpy) %.1f us | pinned only %.1f us | memcpy only %.1f us\n",
               med_us(50, [&] { CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); }),
               med_us(50, [&] { CK(hipMemcpyAsync(p_out, d_out, out_bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); memcpy(h_out, p_out, out_bytes); }),
               med_us(50, [&] { CK(hipMemcpyAsync(p_out, d_out, out_bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); }),
               med_us(50, [&] { memcpy(h_out, p_out, out_bytes); }));
        for (int nchunk : {2, 4, 8}) {
            hipEvent_t ev[8];
            for (int i = 0; i < nchunk; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            const size_t piece = (out_bytes / nchunk + 255) / 256 * 256;
            printf("D2H out: %d chunks pipelined with the host memcpy %.1f us\n", nchunk, med_us(50, [&] {
                for (int i = 0; i < nchunk; ++i) {
                    const size_t lo = std::min(out_bytes, i * piece), hi = std::min(out_bytes, lo + piece);
                    CK(hipMemcpyAsync(p_out + lo, d_out + lo, hi - lo, hipMemcpyDeviceToHost, s1));
                    CK(hipEventRecord(ev[i], s1));
                }
                for (int i = 0; i < nchunk; ++i) {
                    const size_t lo = std::min(out_bytes, i * piece), hi = std::min(out_bytes, lo + piece);
                    CK(hipEventSynchronize(ev[i]));
                    memcpy(h_out + lo, p_out + lo, hi - lo);
                }
            }));
        }
        printf("H2D out-sized (argmax_f32's upload): pageable %.1f us | staged %.1f us | staged 2 threads %.1f us\n",
               med_us(50, [&] { CK(hipMemcpyAsync(d_out, h_out, out_bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }),
               med_us(50, [&] { memcpy(p_out, h_out, out_bytes); CK(hipMemcpyAsync(d_out, p_out, out_bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }),
               med_us(20, [&] { par_memcpy(p_out, h_out, out_bytes, 2); CK(hipMemcpyAsync(d_out, p_out, out_bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }));
        printf("register+unregister out buffer: %.1f us\n",
               med_us(20, [&] { CK(hipHostRegister(h_out, out_bytes, hipHostRegisterDefault)); CK(hipHostUnregister(h_out)); }));
        // kernels touching pinned host memory directly
        printf("kernel writes out to PINNED host memory: %.1f us | to device memory: %.1f us | kernel reads in from pinned: %.1f us\n",
               med_us(50, [&] { hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s1, (float *)p_out, out_bytes / 4, 1.0f); CK(hipStreamSynchronize(s1)); }),
               med_us(50, [&] { hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s1, (float *)d_out, out_bytes / 4, 1.0f); CK(hipStreamSynchronize(s1)); }),
               med_us(50, [&] { hipLaunchKernelGGL(read_kernel, dim3(256), dim3(256), 0, s1, (const uint4 *)p_in, in_bytes / 16, d_flag); CK(hipStreamSynchronize(s1)); }));
        // whole small iteration, three ways: (a) runtime pageable copies, (b) own staging, (c) zero-copy kernel + memcpy
        printf("iteration in->kernel->out: pageable %.1f us | staged %.1f us | zero-copy write + memcpy %.1f us\n",
               med_us(50, [&] {
                   CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s1));
                   hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s1, (float *)d_out, out_bytes / 4, 1.0f);
                   CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s1));
                   CK(hipStreamSynchronize(s1)); }),
               med_us(50, [&] {
                   memcpy(p_in, h_in, in_bytes);
                   CK(hipMemcpyAsync(d_in, p_in, in_bytes, hipMemcpyHostToDevice, s1));
                   hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s1, (float *)d_out, out_bytes / 4, 1.0f);
                   CK(hipMemcpyAsync(p_out, d_out, out_bytes, hipMemcpyDeviceToHost, s1));
                   CK(hipStreamSynchronize(s1));
                   memcpy(h_out, p_out, out_bytes); }),
               med_us(50, [&] {
                   memcpy(p_in, h_in, in_bytes);
                   hipLaunchKernelGGL(read_kernel, dim3(256), dim3(256), 0, s1, (const uint4 *)p_in, in_bytes / 16, d_flag);
                   hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s1, (float *)p_out, out_bytes / 4, 1.0f);
                   CK(hipStreamSynchronize(s1));
                   memcpy(h_out, p_out, out_bytes); }));
        free(h_in); free(h_out);
        CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipHostFree(p_in)); CK(hipHostFree(p_out));
    }

    // ---------------- large: 1 B/pos up, 4 B/pos down ----------------
    const size_t in_bytes = L, out_bytes = 4 * L;
    uint8_t *h_in = (uint8_t *)malloc(in_bytes), *h_out = (uint8_t *)malloc(out_bytes);
    memset(h_in, 1, in_bytes); memset(h_out, 2, out_bytes);
    uint8_t *d_in, *d_out;
    CK(hipMalloc(&d_in, in_bytes)); CK(hipMalloc(&d_out, out_bytes));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s1, (float *)d_out, out_bytes / 4, 1.0f);
    CK(hipStreamSynchronize(s1));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); CK(hipMemcpy(d_in, h_in, in_bytes, hipMemcpyHostToDevice)); double t = now() - t0;
        printf("H2D pageable whole %zu B: %.1f ms = %.1f GB/s\n", in_bytes, t * 1e3, in_bytes / t / 1e9);
        t0 = now(); CK(hipMemcpy(h_out, d_out, out_bytes, hipMemcpyDeviceToHost)); t = now() - t0;
        printf("D2H pageable whole %zu B: %.1f ms = %.1f GB/s\n", out_bytes, t * 1e3, out_bytes / t / 1e9);
    }
    {   // does a pageable async copy return before it completes?
        const size_t tin = 32u << 20, tout = 128u << 20;
        double t0 = now(); CK(hipMemcpyAsync(d_in, h_in, tin, hipMemcpyHostToDevice, s1)); double tc = now() - t0;
        CK(hipStreamSynchronize(s1)); double tt = now() - t0;
        printf("pageable H2D 32 MB async: call returns after %.2f ms, complete after %.2f ms\n", tc * 1e3, tt * 1e3);
        t0 = now(); CK(hipMemcpyAsync(h_out, d_out, tout, hipMemcpyDeviceToHost, s2)); tc = now() - t0;
        CK(hipStreamSynchronize(s2)); tt = now() - t0;
        printf("pageable D2H 128 MB async: call returns after %.2f ms, complete after %.2f ms\n", tc * 1e3, tt * 1e3);
    }
    for (size_t tin : {(size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20}) {
        const size_t tout = 4 * tin, nt = (in_bytes + tin - 1) / tin;
        // one host thread, three streams, tiles: up(t+1) | kernel(t) | down(t-1)
        hipEvent_t up[2], done[2], down[2];
        for (int b = 0; b < 2; ++b) { CK(hipEventCreateWithFlags(&up[b], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&down[b], hipEventDisableTiming)); }
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            for (size_t t = 0; t < nt; ++t) {
                const size_t wi = std::min(tin, in_bytes - t * tin), wo = 4 * wi;
                CK(hipMemcpyAsync(d_in + t * tin, h_in + t * tin, wi, hipMemcpyHostToDevice, s1));
                CK(hipEventRecord(up[t & 1], s1));
                CK(hipStreamWaitEvent(s2, up[t & 1], 0));
                hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(256), 0, s2, (const uint4 *)(d_in + t * tin), wi / 16, d_flag);
                CK(hipEventRecord(done[t & 1], s2));
                CK(hipStreamWaitEvent(s3, done[t & 1], 0));
                CK(hipMemcpyAsync(h_out + t * tout, d_out + t * tout, wo, hipMemcpyDeviceToHost, s3));
            }
            CK(hipStreamSynchronize(s3));
            double t = now() - t0;
            printf("tiles of %zu MB in, ONE host thread, 3 streams: %.1f ms = %.2f Gpos/s\n", tin >> 20, t * 1e3, L / t / 1e9);
        }
        // two host threads: uploader runs ahead (bounded by 3 tiles), main does kernel + download
        for (int rep = 0; rep < 2; ++rep) {
            std::atomic<size_t> uploaded{0}, consumed{0};
            double t0 = now();
            std::thread upl([&] {
                for (size_t t = 0; t < nt; ++t) {
                    while (t >= consumed.load(std::memory_order_acquire) + 3) std::this_thread::yield();
                    const size_t wi = std::min(tin, in_bytes - t * tin);
                    CK(hipMemcpyAsync(d_in + t * tin, h_in + t * tin, wi, hipMemcpyHostToDevice, s1));
                    CK(hipStreamSynchronize(s1));
                    uploaded.store(t + 1, std::memory_order_release);
                }
            });
            for (size_t t = 0; t < nt; ++t) {
                const size_t wi = std::min(tin, in_bytes - t * tin), wo = 4 * wi;
                while (uploaded.load(std::memory_order_acquire) <= t) std::this_thread::yield();
                hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(256), 0, s2, (const uint4 *)(d_in + t * tin), wi / 16, d_flag);
                CK(hipMemcpyAsync(h_out + t * tout, d_out + t * tout, wo, hipMemcpyDeviceToHost, s2));
                CK(hipStreamSynchronize(s2));
                consumed.store(t + 1, std::memory_order_release);
            }
            upl.join();
            double t = now() - t0;
            printf("tiles of %zu MB in, TWO host threads (uploader + kernel/download): %.1f ms = %.2f Gpos/s\n", tin >> 20, t * 1e3, L / t / 1e9);
        }
        // three host threads: uploader, and two downloaders taking alternate tiles
        for (int rep = 0; rep < 2; ++rep) {
            std::atomic<size_t> uploaded{0};
            double t0 = now();
            std::thread upl([&] {
                for (size_t t = 0; t < nt; ++t) {
                    const size_t wi = std::min(tin, in_bytes - t * tin);
                    CK(hipMemcpyAsync(d_in + t * tin, h_in + t * tin, wi, hipMemcpyHostToDevice, s1));
                    CK(hipStreamSynchronize(s1));
                    uploaded.store(t + 1, std::memory_order_release);
                }
            });
            auto down_fn = [&](size_t first, hipStream_t st) {
                for (size_t t = first; t < nt; t += 2) {
                    const size_t wi = std::min(tin, in_bytes - t * tin), wo = 4 * wi;
                    while (uploaded.load(std::memory_order_acquire) <= t) std::this_thread::yield();
                    hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(256), 0, st, (const uint4 *)(d_in + t * tin), wi / 16, d_flag);
                    CK(hipMemcpyAsync(h_out + t * tout, d_out + t * tout, wo, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                }
            };
            std::thread d2([&] { down_fn(1, s3); });
            down_fn(0, s2);
            d2.join();
            upl.join();
            double t = now() - t0;
            printf("tiles of %zu MB in, THREE host threads (uploader + 2 downloaders): %.1f ms = %.2f Gpos/s\n", tin >> 20, t * 1e3, L / t / 1e9);
        }
    }
    {   // D2H alone in tiles from two threads (is one pageable D2H stream the limit?)
        const size_t tout = 128u << 20, nt = out_bytes / tout;
        double t0 = now();
        auto fn = [&](size_t first, hipStream_t st) {
            for (size_t t = first; t < nt; t += 2) { CK(hipMemcpyAsync(h_out + t * tout, d_out + t * tout, tout, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
        };
        std::thread d2([&] { fn(1, s3); });
        fn(0, s2);
        d2.join();
        double t = now() - t0;
        printf("D2H alone, 128 MB tiles, two threads: %.1f ms = %.1f GB/s\n", t * 1e3, nt * tout / t / 1e9);
    }
    {   // registered output: pin cost vs rate
        double t0 = now(); CK(hipHostRegister(h_out, out_bytes, hipHostRegisterDefault)); double tr = now() - t0;
        t0 = now(); CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); double tc = now() - t0;
        t0 = now();
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s1, (float *)h_out, out_bytes / 4, 1.0f);
        CK(hipStreamSynchronize(s1));
        double tk = now() - t0;
        t0 = now(); CK(hipHostUnregister(h_out)); double tu = now() - t0;
        printf("registered out: register %.1f ms, D2H %.1f ms (%.1f GB/s), kernel writing it directly %.1f ms (%.1f GB/s), unregister %.1f ms\n",
               tr * 1e3, tc * 1e3, out_bytes / tc / 1e9, tk * 1e3, out_bytes / tk / 1e9, tu * 1e3);
    }
    return 0;
}
