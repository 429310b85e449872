// symload_bench -- what bounds the symbol loads of the pair scans (score_prefilter2.hpp: load_block)?
// The scans of M <= 20 take ~217 us per Gbp whatever their table size, occupancy and the residency of the sequence: 33
// cycles per 256-position item and CU that neither the LDS nor the VALU model explains.  This tool issues ONLY the loads,
// with the scan's stream geometry (wavefront = 2 streams x 32 columns, stream = T consecutive rows, one dword per lane and
// 4-row block, PF blocks in flight), in three lane -> address maps:
//   quad    the scan's: lane q of quad i reads row r + q, columns 4i .. 4i + 3 (a half-wave covers 128 contiguous bytes,
//           adjacent lanes 32 bytes apart)
//   linear  lane l of a half-wave reads dword l of the same 128 bytes (adjacent lanes adjacent)
//   x4      lane l reads 16 bytes (four blocks at once): 512 contiguous bytes per half-wave instruction
//   bytes   one byte per lane and ROW, the one-symbol (protein) scans' loads
//   ./symload_bench [positions = 1e9] [T = 1024] [rounds = 5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MAP, int PF>
__global__ __launch_bounds__(256) void loads_only(const uint8_t *__restrict__ seq, const unsigned long long T,
                                                  const unsigned long long nstreams, unsigned *__restrict__ out)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
    unsigned long long stream = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    if (stream >= nstreams)
        stream = nstreams - 1;
    const uint8_t *p = seq + stream * T * 32;
    unsigned acc = 0;
    const unsigned long long nblk = T / 4;
    if (MAP == 3) {  // the one-symbol scans: a byte per lane and row (32 contiguous bytes per half-wave instruction)
        const uint8_t *q = p + l;
        unsigned ring[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i)
            ring[i] = q[(size_t)i * 32];
        for (unsigned long long r = 0; r + PF < T; r += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const unsigned v = ring[i];
                ring[i] = q[(size_t)(r + PF + i) * 32];
                acc += v;
            }
        }
    } else if (MAP == 2) {
        const uint8_t *q = p + l * 16;  // 512 contiguous bytes per half-wave and step of 16 rows
        u32x4 ring[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i)
            ring[i] = *reinterpret_cast<const u32x4 *>(q + (size_t)i * 512);
        for (unsigned long long b = 0; b + PF < nblk / 4; b += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const u32x4 v = ring[i];
                ring[i] = *reinterpret_cast<const u32x4 *>(q + (size_t)(b + PF + i) * 512);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    } else {
        const uint8_t *q = MAP == 0 ? p + (l & 3) * 32 + (l >> 2) * 4 : p + l * 4;
        unsigned ring[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i)
            ring[i] = *reinterpret_cast<const unsigned *>(q + (size_t)i * 128);
        for (unsigned long long b = 0; b + PF < nblk; b += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const unsigned v = ring[i];
                ring[i] = *reinterpret_cast<const unsigned *>(q + (size_t)(b + PF + i) * 128);
                acc += v;
            }
        }
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

int main(int argc, char **argv)
{
    const unsigned long long n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const unsigned long long T = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1024;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    const unsigned long long rows = n / 32, nstreams = rows / T;
    uint8_t *seq;
    unsigned *out;
    CK(hipMalloc(&seq, rows * 32 + 65536));
    CK(hipMemset(seq, 1, rows * 32 + 65536));
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((nstreams + 7) / 8);
    struct V { const char *name; void (*k)(const uint8_t *, unsigned long long, unsigned long long, unsigned *); };
    const V vs[] = {{"quad   PF6", loads_only<0, 6>}, {"linear PF6", loads_only<1, 6>}, {"quad   PF12", loads_only<0, 12>},
                    {"linear PF12", loads_only<1, 12>}, {"x4     PF2", loads_only<2, 2>}, {"x4     PF4", loads_only<2, 4>},
                    {"bytes  PF12", loads_only<3, 12>}, {"bytes  PF24", loads_only<3, 24>}};
    for (int r = 0; r < rounds; ++r)
        for (const V &v : vs) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, seq, T, nstreams, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r)
                printf("%-12s %8.1f us  %6.2f TB/s\n", v.name, ms * 1e3, (double)nstreams * T * 32 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
