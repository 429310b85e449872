// read_bench -- what bounds the 4 B/cell streaming reads of reduce.hip (argmax_flat, threshold_count)?
// Development tool: the shipped kernels' loops next to variants (loads in flight per thread, lane-contiguous
// 16-byte loads, cells per workgroup), all over one 4 GB f32 matrix, interleaved rounds.
//   ./read_bench [cells = 1e9] [rounds = 7]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Rec { float v; long long i; };

__device__ __forceinline__ void upd(float &v, long long &bi, const f32x4 x, long long base)
{
    if (x.x >= v) { v = x.x; bi = base; }
    if (x.y >= v) { v = x.y; bi = base + 1; }
    if (x.z >= v) { v = x.z; bi = base + 2; }
    if (x.w >= v) { v = x.w; bi = base + 3; }
}

__device__ __forceinline__ void wave_out(float v, long long bi, Rec *out)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const long long oi = __shfl_xor(bi, off);
        if (ov > v || (ov == v && oi > bi)) { v = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0)
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Rec{v, bi};
}

// the shipped loop: one 16-byte load per thread and trip, grid-stride
template <bool NT>
__global__ __launch_bounds__(256) void am_base(const f32x4 *__restrict__ s, unsigned long long n4, Rec *out)
{
    float v = -INFINITY;
    long long bi = -1;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4;
         i += (unsigned long long)gridDim.x * 256) {
        const f32x4 x = NT ? __builtin_nontemporal_load(&s[i]) : s[i];
        upd(v, bi, x, (long long)(i * 4));
    }
    wave_out(v, bi, out);
}

// U independent loads per trip (ascending addresses, so the `>=` rule still keeps the later cell)
template <int U, bool NT>
__global__ __launch_bounds__(256) void am_unroll(const f32x4 *__restrict__ s, unsigned long long n4, Rec *out)
{
    float v = -INFINITY;
    long long bi = -1;
    const unsigned long long step = (unsigned long long)gridDim.x * 256;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * step < n4; i += U * step) {
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = NT ? __builtin_nontemporal_load(&s[i + u * step]) : s[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u)
            upd(v, bi, x[u], (long long)((i + u * step) * 4));
    }
    for (; i < n4; i += step)
        upd(v, bi, s[i], (long long)(i * 4));
    wave_out(v, bi, out);
}

// a workgroup owns a contiguous span (block-contiguous instead of grid-stride), U loads in flight
template <int U>
__global__ __launch_bounds__(256) void am_span(const f32x4 *__restrict__ s, unsigned long long n4, Rec *out)
{
    float v = -INFINITY;
    long long bi = -1;
    const unsigned long long per = (n4 + gridDim.x - 1) / gridDim.x;
    const unsigned long long a = (unsigned long long)blockIdx.x * per, b = a + per < n4 ? a + per : n4;
    unsigned long long i = a + threadIdx.x;
    for (; i + (U - 1) * 256 < b; i += U * 256) {
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = __builtin_nontemporal_load(&s[i + u * 256]);
#pragma unroll
        for (int u = 0; u < U; ++u)
            upd(v, bi, x[u], (long long)((i + u * 256) * 4));
    }
    for (; i < b; i += 256)
        upd(v, bi, s[i], (long long)(i * 4));
    wave_out(v, bi, out);
}

// plain sum: the ceiling of a read stream with this loop shape
template <int U>
__global__ __launch_bounds__(256) void rd_sum(const f32x4 *__restrict__ s, unsigned long long n4, float *out)
{
    float acc = 0;
    const unsigned long long step = (unsigned long long)gridDim.x * 256;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * step < n4; i += U * step) {
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = __builtin_nontemporal_load(&s[i + u * step]);
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc += x[u].x + x[u].y + x[u].z + x[u].w;
    }
    if (acc == 12345.678f)
        out[0] = acc;
}

// ---- threshold_count -------------------------------------------------------------------
constexpr int kChunk = 4096;

// shipped: a thread reads its 16 consecutive cells (lane stride 64 B), one chunk per workgroup
__global__ __launch_bounds__(256) void tc_base(const float *__restrict__ s, unsigned long long ncells, float t,
                                               unsigned *__restrict__ counts)
{
    __shared__ unsigned sm[4];
    const unsigned long long e0 = (unsigned long long)blockIdx.x * kChunk + threadIdx.x * 16;
    unsigned c = 0;
    if (e0 + 16 <= ncells) {
        const float4 *p = reinterpret_cast<const float4 *>(s + e0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = p[q];
            c += (x.x >= t) + (x.y >= t) + (x.z >= t) + (x.w >= t);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0)
        sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// lane-contiguous 16-byte loads (a wavefront reads 1 KB per instruction), one chunk per workgroup
template <bool NT>
__global__ __launch_bounds__(256) void tc_coal(const f32x4 *__restrict__ s, unsigned long long n4, float t,
                                               unsigned *__restrict__ counts)
{
    __shared__ unsigned sm[4];
    const unsigned long long b = (unsigned long long)blockIdx.x * (kChunk / 4);
    unsigned c = 0;
    f32x4 x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long i = b + q * 256 + threadIdx.x;
        x[q] = i < n4 ? (NT ? __builtin_nontemporal_load(&s[i]) : s[i]) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        c += (x[q].x >= t) + (x[q].y >= t) + (x[q].z >= t) + (x[q].w >= t);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0)
        sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// a WAVEFRONT owns a chunk (16 lane-contiguous loads of 1 KB), a workgroup NCH chunks per trip, grid-stride over
// chunk groups: no barrier, no LDS, few workgroups
template <int PER, bool NT>
__global__ __launch_bounds__(256) void tc_wave(const f32x4 *__restrict__ s, unsigned long long n4, float t,
                                               unsigned *__restrict__ counts, unsigned long long nchunks)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned long long ch = (unsigned long long)blockIdx.x * 4 + wave; ch < nchunks; ch += (unsigned long long)gridDim.x * 4) {
        const unsigned long long b = ch * (kChunk / 4);
        unsigned c = 0;
#pragma unroll
        for (int h = 0; h < 16 / PER; ++h) {
            f32x4 x[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const unsigned long long i = b + (h * PER + q) * 64 + lane;
                x[q] = i < n4 ? (NT ? __builtin_nontemporal_load(&s[i]) : s[i])
                              : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
#pragma unroll
            for (int q = 0; q < PER; ++q)
                c += (x[q].x >= t) + (x[q].y >= t) + (x[q].z >= t) + (x[q].w >= t);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            c += __shfl_xor(c, off);
        if (lane == 0)
            counts[ch] = c;
    }
}

// two chunks per workgroup (waves 0-1 the first, waves 2-3 the second): 8 loads in flight per lane
__global__ __launch_bounds__(256) void tc_coal2(const f32x4 *__restrict__ s, unsigned long long n4, float t,
                                                unsigned *__restrict__ counts, unsigned long long nchunks)
{
    __shared__ unsigned sm[4];
    const int half = threadIdx.x >> 7, tl = threadIdx.x & 127;
    const unsigned long long ch = (unsigned long long)blockIdx.x * 2 + half;
    const unsigned long long b = ch * (kChunk / 4);
    unsigned c = 0;
    f32x4 x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned long long i = b + q * 128 + tl;
        x[q] = i < n4 ? __builtin_nontemporal_load(&s[i]) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
        c += (x[q].x >= t) + (x[q].y >= t) + (x[q].z >= t) + (x[q].w >= t);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0)
        sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (tl == 0 && ch < nchunks)
        counts[ch] = sm[2 * half] + sm[2 * half + 1];
}

__global__ void fill(float *s, unsigned long long n)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27;
        s[i] = -30.0f + (float)(z & 0xffff) * (40.0f / 65536.0f);
    }
}

int main(int argc, char **argv)
{
    const unsigned long long n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ull;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const unsigned long long n4 = n / 4, nchunks = (n + kChunk - 1) / kChunk;
    float *s;
    Rec *recs;
    unsigned *counts;
    float *sink;
    CK(hipMalloc(&s, n * 4));
    CK(hipMalloc(&recs, sizeof(Rec) * (1 << 22)));
    CK(hipMalloc(&counts, 4 * nchunks));
    CK(hipMalloc(&sink, 64));
    fill<<<4096, 256>>>(s, n);
    CK(hipDeviceSynchronize());
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(s);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    struct V { const char *name; std::vector<float> ms; int kind; };
    std::vector<V> vs;
    auto add = [&](const char *nm, int kind) { vs.push_back(V{nm, {}, kind}); };
    const float t = 9.99f;
    add("am_base nt g4096 (shipped)", 0);
    add("am_unroll4 nt g4096", 3);
    add("am_span4 g16384", 30);
    add("am_span4 g32768", 31);
    add("am_span4 g65536", 32);
    add("am_span8 g16384", 8);
    add("am_span8 g32768", 33);
    add("am_span8 g65536", 34);
    add("am_span16 g16384", 35);
    add("am_span16 g32768", 36);
    add("am_span16 g61036 (one trip)", 37);
    add("am_span8 g122071 (one trip)", 38);
    add("rd_sum8 g2048", 10);
    add("tc_base (shipped)", 20);
    add("tc_coal nt", 22);
    add("tc_coal2 nt (2 chunks per workgroup)", 40);
    add("tc_wave per8 nt g16384", 41);
    for (int r = 0; r < rounds + 1; ++r) {
        for (auto &v : vs) {
            CK(hipEventRecord(a));
            switch (v.kind) {
            case 0: am_base<true><<<4096, 256>>>(s4, n4, recs); break;
            case 1: am_base<false><<<4096, 256>>>(s4, n4, recs); break;
            case 2: am_unroll<2, true><<<4096, 256>>>(s4, n4, recs); break;
            case 3: am_unroll<4, true><<<4096, 256>>>(s4, n4, recs); break;
            case 4: am_unroll<4, true><<<2048, 256>>>(s4, n4, recs); break;
            case 5: am_unroll<8, true><<<2048, 256>>>(s4, n4, recs); break;
            case 6: am_unroll<4, false><<<2048, 256>>>(s4, n4, recs); break;
            case 7: am_span<4><<<8192, 256>>>(s4, n4, recs); break;
            case 8: am_span<8><<<16384, 256>>>(s4, n4, recs); break;
            case 9: rd_sum<4><<<2048, 256>>>(s4, n4, sink); break;
            case 10: rd_sum<8><<<2048, 256>>>(s4, n4, sink); break;
            case 30: am_span<4><<<16384, 256>>>(s4, n4, recs); break;
            case 31: am_span<4><<<32768, 256>>>(s4, n4, recs); break;
            case 32: am_span<4><<<65536, 256>>>(s4, n4, recs); break;
            case 33: am_span<8><<<32768, 256>>>(s4, n4, recs); break;
            case 34: am_span<8><<<65536, 256>>>(s4, n4, recs); break;
            case 35: am_span<16><<<16384, 256>>>(s4, n4, recs); break;
            case 36: am_span<16><<<32768, 256>>>(s4, n4, recs); break;
            case 37: am_span<16><<<(unsigned)((n4 + 4095) / 4096), 256>>>(s4, n4, recs); break;
            case 38: am_span<8><<<(unsigned)((n4 + 2047) / 2048), 256>>>(s4, n4, recs); break;
            case 40: tc_coal2<<<(unsigned)((nchunks + 1) / 2), 256>>>(s4, n4, t, counts, nchunks); break;
            case 41: tc_wave<8, true><<<16384, 256>>>(s4, n4, t, counts, nchunks); break;
            case 20: tc_base<<<(unsigned)nchunks, 256>>>(s, n, t, counts); break;
            case 21: tc_coal<false><<<(unsigned)nchunks, 256>>>(s4, n4, t, counts); break;
            case 22: tc_coal<true><<<(unsigned)nchunks, 256>>>(s4, n4, t, counts); break;
            case 23: tc_wave<4, true><<<4096, 256>>>(s4, n4, t, counts, nchunks); break;
            case 24: tc_wave<8, true><<<4096, 256>>>(s4, n4, t, counts, nchunks); break;
            case 25: tc_wave<16, true><<<2048, 256>>>(s4, n4, t, counts, nchunks); break;
            case 26: tc_wave<8, false><<<4096, 256>>>(s4, n4, t, counts, nchunks); break;
            case 27: tc_wave<8, true><<<8192, 256>>>(s4, n4, t, counts, nchunks); break;
            }
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (r)
                v.ms.push_back(ms);
        }
    }
    printf("read_bench: %llu cells (%.2f GB), %d rounds interleaved\n", n, n * 4 / 1e9, rounds);
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        printf("%-32s median %.4f ms  min %.4f ms  %.0f GB/s (%.3f of 8 TB/s)\n", v.name, med, v.ms[0], n * 4 / med / 1e6,
               n * 4 / med / 1e6 / 8000.0);
    }
    return 0;
}
