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

// the score kernel's own access pattern without its arithmetic: a half-wave walks T
// consecutive rows of 32 columns (1 byte read, 1 float written per lane and step)
template <int PFD>
__global__ __launch_bounds__(256) void mix_rows(const uint8_t *in, float *out, unsigned long long rows,
                                                unsigned long long T)
{
    const unsigned long long stream = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + ((threadIdx.x & 63) >> 5);
    const unsigned col = threadIdx.x & 31;
    unsigned long long r0 = stream * T;
    if (r0 >= rows) return;
    unsigned long long r1 = r0 + T < rows ? r0 + T : rows;
    const uint8_t *ip = in + r0 * 32 + col;
    float *op = out + r0 * 32 + col;
    unsigned ring[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) ring[k] = ip[k * 32];
    for (unsigned long long r = r0; r < r1; r += PFD) {
#pragma unroll
        for (int k = 0; k < PFD; ++k) {
            const unsigned s = ring[k];
            ring[k] = ip[(k + PFD) * 32];
            if (r + k < r1)
                __builtin_nontemporal_store((float)s, op + k * 32);
        }
        ip += PFD * 32;
        op += PFD * 32;
    }
}

// XCD <-> memory affinity test: workgroup b (dispatched to XCD b % 8) writes the contiguous chunk of `chunk16` float4
// number c, where c = b (mode 0), b ^ 1 (mode 1), b rotated by 3 inside its group of eight (mode 2) or bit-reversed
// inside its group of 256 (mode 3).  If a workgroup's XCD had "near" and "far" addresses, the modes would differ.
__global__ __launch_bounds__(256) void fill_chunks(f32x4 *out, unsigned long long chunk16, int mode)
{
    unsigned long long b = blockIdx.x, c = b;
    if (mode == 1) c = b ^ 1ull;
    if (mode == 2) c = (b & ~7ull) | ((b + 3) & 7ull);
    if (mode == 3) c = (b & ~255ull) | (__brev((unsigned)(b & 255)) >> 24);
    f32x4 *p = out + c * chunk16;
    const f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (unsigned long long i = threadIdx.x; i < chunk16; i += 256)
        __builtin_nontemporal_store(v, &p[i]);
}

// mix_rows with the store's cache-policy bits chosen by hand (gfx940-family syntax: sc0 / sc1 = coherence scope,
// nt = non-temporal): does any policy change what the row pattern costs the memory side?
template <int POL>
__device__ __forceinline__ void store_policy(float *p, float v)
{
    if (POL == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int PFD, int POL>
__global__ __launch_bounds__(256) void mix_rows_pol(const uint8_t *in, float *out, unsigned long long rows,
                                                    unsigned long long T)
{
    const unsigned long long stream = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + ((threadIdx.x & 63) >> 5);
    const unsigned col = threadIdx.x & 31;
    unsigned long long r0 = stream * T;
    if (r0 >= rows) return;
    unsigned long long r1 = r0 + T < rows ? r0 + T : rows;
    const uint8_t *ip = in + r0 * 32 + col;
    float *op = out + r0 * 32 + col;
    unsigned ring[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) ring[k] = ip[k * 32];
    for (unsigned long long r = r0; r < r1; r += PFD) {
#pragma unroll
        for (int k = 0; k < PFD; ++k) {
            const unsigned s = ring[k];
            ring[k] = ip[(k + PFD) * 32];
            if (r + k < r1)
                store_policy<POL>(op + k * 32, (float)s);
        }
        ip += PFD * 32;
        op += PFD * 32;
    }
}

// the score kernel's pattern with 4 rows gathered per store: lane (c) writes 16 bytes of row
// r + (c & 3), columns 4*(c >> 2).. -- what a 4x4 quad transpose (DPP, no LDS) would allow:
// a half-wave store covers 512 contiguous bytes
template <int PFD>
__global__ __launch_bounds__(256) void mix_rows_q(const uint8_t *in, float *out, unsigned long long rows,
                                                  unsigned long long T)
{
    const unsigned long long stream = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + ((threadIdx.x & 63) >> 5);
    const unsigned col = threadIdx.x & 31;
    unsigned long long r0 = stream * T;
    if (r0 >= rows) return;
    unsigned long long r1 = r0 + T < rows ? r0 + T : rows;
    const uint8_t *ip = in + r0 * 32 + col;
    f32x4 *op = reinterpret_cast<f32x4 *>(out + (r0 + (col & 3)) * 32 + (col >> 2) * 4);
    unsigned ring[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) ring[k] = ip[k * 32];
    for (unsigned long long r = r0; r < r1; r += PFD) {
#pragma unroll
        for (int k = 0; k < PFD; k += 4) {
            unsigned s[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[q] = ring[k + q];
                ring[k + q] = ip[(k + q + PFD) * 32];
            }
            if (r + k + 3 < r1) {
                f32x4 v = {(float)s[0], (float)s[1], (float)s[2], (float)s[3]};
                __builtin_nontemporal_store(v, op + k * 8);
            }
        }
        ip += PFD * 32;
        op += PFD * 8;
    }
}

// 2 x 2 variant: lane pairs transpose two completed rows, every lane writes 8 bytes (2 columns of one
// row); a half-wave store covers 2 rows = 256 contiguous bytes
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PFD>
__global__ __launch_bounds__(256) void mix_rows_p2(const uint8_t *in, float *out, unsigned long long rows,
                                                   unsigned long long T)
{
    const unsigned long long stream = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + ((threadIdx.x & 63) >> 5);
    const unsigned col = threadIdx.x & 31;
    unsigned long long r0 = stream * T;
    if (r0 >= rows) return;
    unsigned long long r1 = r0 + T < rows ? r0 + T : rows;
    const uint8_t *ip = in + r0 * 32 + col;
    f32x2 *op = reinterpret_cast<f32x2 *>(out + (r0 + (col & 1)) * 32 + (col >> 1) * 2);
    unsigned ring[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) ring[k] = ip[k * 32];
    for (unsigned long long r = r0; r < r1; r += PFD) {
#pragma unroll
        for (int k = 0; k < PFD; k += 2) {
            unsigned s[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                s[q] = ring[k + q];
                ring[k + q] = ip[(k + q + PFD) * 32];
            }
            if (r + k + 1 < r1) {
                f32x2 v = {(float)s[0], (float)s[1]};
                __builtin_nontemporal_store(v, op + k * 16);
            }
        }
        ip += PFD * 32;
        op += PFD * 16;
    }
}

// the two halves of a wavefront on ADJACENT rows of one stream (row-phase split): a wavefront store
// covers 256 contiguous bytes (dword per lane), a stream is swept by the whole wavefront
template <int PFD>
__global__ __launch_bounds__(256) void mix_rows_w(const uint8_t *in, float *out, unsigned long long rows,
                                                  unsigned long long T)
{
    const unsigned long long stream = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    unsigned long long r0 = stream * T;   // T rows per wavefront, even
    if (r0 >= rows) return;
    unsigned long long r1 = r0 + T < rows ? r0 + T : rows;
    const uint8_t *ip = in + r0 * 32 + lane;          // rows r, r+1 = 64 contiguous bytes
    float *op = out + r0 * 32 + lane;
    unsigned ring[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) ring[k] = ip[k * 64];
    for (unsigned long long r = r0; r < r1; r += 2 * PFD) {
#pragma unroll
        for (int k = 0; k < PFD; ++k) {
            const unsigned s = ring[k];
            ring[k] = ip[(k + PFD) * 64];
            if (r + 2 * k + 1 < r1)
                __builtin_nontemporal_store((float)s, op + k * 64);
        }
        ip += PFD * 64;
        op += PFD * 64;
    }
}

// same, but 8 completed rows of a stream are staged in LDS and written as 1 KB by the whole wavefront
__global__ __launch_bounds__(256) void mix_rows_lds(const uint8_t *in, f32x4 *out, unsigned long long rows,
                                                    unsigned long long T)
{
    __shared__ float stage[8][8 * 32];  // per stream of the block: 8 rows x 32 floats
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
    const unsigned long long stream = ((unsigned long long)blockIdx.x * 4 + wave) * 2 + half;
    const unsigned col = lane & 31;
    const unsigned long long r0 = stream * T;     // T multiple of 8; rows multiple of T assumed by the caller
    if (r0 >= rows) return;
    const uint8_t *ip = in + r0 * 32 + col;
    float *mine = stage[wave * 2 + half];
    for (unsigned long long r = 0; r < T; r += 8) {
        unsigned s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = ip[(r + k) * 32];
#pragma unroll
        for (int k = 0; k < 8; ++k) mine[k * 32 + col] = (float)s[k];
        __builtin_amdgcn_wave_barrier();
        // each stream's 1 KB goes out as 64 lanes x 16 B; the wavefront does its two streams in turn
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float *src = stage[wave * 2 + h];
            const unsigned long long sr0 = (stream - half + h) * T + r;
            f32x4 v = *reinterpret_cast<const f32x4 *>(src + lane * 4);
            __builtin_nontemporal_store(v, out + (sr0 * 32) / 4 + lane);
        }
        __builtin_amdgcn_wave_barrier();
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
    {
        const unsigned long long rows = n / 32;
        for (unsigned long long T : {56ull, 60ull, 61ull, 64ull, 72ull, 80ull, 96ull, 128ull, 61ull, 64ull}) {
            const unsigned long long nstreams = (rows + T - 1) / T;
            const unsigned grid = (unsigned)((nstreams + 7) / 8);
            char nm[64];
            snprintf(nm, sizeof nm, "mix_rows<12> T=%llu", T);
            rep(nm, timeit([&] { hipLaunchKernelGGL(mix_rows<12>, dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20));
            snprintf(nm, sizeof nm, "mix_rows_q<12> T=%llu", T);
            rep(nm, timeit([&] { hipLaunchKernelGGL(mix_rows_q<12>, dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20));
            snprintf(nm, sizeof nm, "mix_rows_p2<12> T=%llu", T);
            rep(nm, timeit([&] { hipLaunchKernelGGL(mix_rows_p2<12>, dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20));
            snprintf(nm, sizeof nm, "mix_rows_w<12> T=%llu", 2 * T);
            rep(nm, timeit([&] { hipLaunchKernelGGL(mix_rows_w<12>, dim3((unsigned)(((rows + 2 * T - 1) / (2 * T) + 3) / 4)), dim3(256), 0, 0, in, out, rows, 2 * T); }, 20));
            snprintf(nm, sizeof nm, "mix_rows_lds T=%llu", T);
            const unsigned long long rows8 = rows / T * T;
            rep(nm, timeit([&] { hipLaunchKernelGGL(mix_rows_lds, dim3((unsigned)(rows8 / T / 8)), dim3(256), 0, 0, in, (f32x4 *)out, rows8, T); }, 20));
        }
    }
    {   // XCD <-> address affinity: 4 GB written in chunks of 4 KB ... 1 MB per workgroup under four chunk permutations
        for (unsigned long long chunk_bytes : {4096ull, 16384ull, 65536ull, 1048576ull}) {
            const unsigned long long chunk16 = chunk_bytes / 16, nchunks = (n * 4) / chunk_bytes / 256 * 256;
            for (int mode = 0; mode < 4; ++mode) {
                char nm[64];
                snprintf(nm, sizeof nm, "fill_chunks %lluK mode %d", chunk_bytes >> 10, mode);
                float ms = timeit([&] { hipLaunchKernelGGL(fill_chunks, dim3((unsigned)nchunks), dim3(256), 0, 0, (f32x4 *)out, chunk16, mode); }, 20);
                printf("%-28s %8.3f ms  %7.1f GB/s\n", nm, ms, (double)nchunks * chunk_bytes / ms / 1e6);
            }
        }
    }
    {   // the row pattern under every store cache policy (two rounds, interleaved)
        const unsigned long long rows = n / 32, T = 61;
        const unsigned grid = (unsigned)(((rows + T - 1) / T + 7) / 8);
        const char *names[7] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt"};
        for (int round = 0; round < 2; ++round) {
            float ms[7];
            ms[0] = timeit([&] { hipLaunchKernelGGL((mix_rows_pol<12, 0>), dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20);
            ms[1] = timeit([&] { hipLaunchKernelGGL((mix_rows_pol<12, 1>), dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20);
            ms[2] = timeit([&] { hipLaunchKernelGGL((mix_rows_pol<12, 2>), dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20);
            ms[3] = timeit([&] { hipLaunchKernelGGL((mix_rows_pol<12, 3>), dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20);
            ms[4] = timeit([&] { hipLaunchKernelGGL((mix_rows_pol<12, 4>), dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20);
            ms[5] = timeit([&] { hipLaunchKernelGGL((mix_rows_pol<12, 5>), dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20);
            ms[6] = timeit([&] { hipLaunchKernelGGL((mix_rows_pol<12, 6>), dim3(grid), dim3(256), 0, 0, in, out, rows, T); }, 20);
            for (int p = 0; p < 7; ++p) {
                char nm[64];
                snprintf(nm, sizeof nm, "mix_rows T=61 store %s", names[p]);
                rep(nm, ms[p]);
            }
        }
    }
    {   // the row pattern at reduced occupancy (unused dynamic LDS as ballast): fewer streams open at once
        const unsigned long long rows = n / 32, T = 61;
        const unsigned grid = (unsigned)(((rows + T - 1) / T + 7) / 8);
        for (unsigned lds : {0u, 20u << 10, 40u << 10, 53u << 10, 80u << 10, 160u << 10}) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mix_rows<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
            char nm[64];
            snprintf(nm, sizeof nm, "mix_rows T=61 lds=%uK", lds >> 10);
            rep(nm, timeit([&] { hipLaunchKernelGGL(mix_rows<12>, dim3(grid), dim3(256), lds, 0, in, out, rows, T); }, 20));
        }
    }
    return 0;
}