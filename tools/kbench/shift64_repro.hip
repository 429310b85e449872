// shift64_repro -- minimal reproducer of the MI355X (gfx950) fault behind the round-5 parity loss.
//
//     v_lshlrev_b64 vdst, vN, vsrc      with vN the LAST VGPR of the wavefront's allocation (N % 8 == 7, N + 1 not allocated)
//
// returns a wrong result for part of the wavefronts, depending on what lies behind the allocation.  LLVM knows the erratum
// for gfx11 (FeatureShift64HighRegBug; GCNHazardRecognizer::fixShift64HighRegBug copies the amount elsewhere); hipcc 7.0 does
// not work around it on gfx950.  In the library it hit `score_c32_prefilter<7|8, 12, WIDE>`, whose "groups per flag bit" shift
// (`gbit <<= 1`, compiled to a v_lshlrev_b64 by a 0 / 1 in v55 of 56 registers) then moved flags to wrong bits: a quarter of
// the candidates lost, differently from run to run.  tools/isa_audit.py scans every shipped kernel for the pattern.
//
// Three kernels that differ ONLY in the register holding the shift amount (pinned through an inline-asm clobber; the
// kernels need fewer than 14 registers, so the allocation is 16):
//     amount in v15   the last allocated register          -> wrong results expected
//     amount in v14   one below                            -> right
//     amount in v15, and v16 clobbered (allocation 24)     -> right: the register behind the amount exists
//   ./shift64_repro [wavefront-lanes = 64 Mi] [launches = 5]      exit status 0 when the fault showed as described
//   hipcc --offload-arch=gfx950 -O3 shift64_repro.hip -o shift64_repro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define SHIFT_KERNEL(NAME, REG, ...)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *__restrict__ out, const unsigned *__restrict__ amounts) \
    {                                                                                                                   \
        const unsigned i = blockIdx.x * 256u + threadIdx.x;                                                             \
        const unsigned amt = amounts[i & 1023u];                                                                        \
        const unsigned long long x = 0x0000000100000001ull;                                                             \
        unsigned long long r;                                                                                           \
        asm volatile("v_mov_b32 " REG ", %1\n\tv_lshlrev_b64 %0, " REG ", %2" : "=v"(r) : "v"(amt), "v"(x) : __VA_ARGS__); \
        out[i] = r;                                                                                                     \
    }

SHIFT_KERNEL(amount_in_last_vgpr, "v15", "v15")
SHIFT_KERNEL(amount_one_below, "v14", "v14")
SHIFT_KERNEL(amount_in_v15_of_24, "v15", "v15", "v16")

// keeps the register file of the chip full of other wavefronts' live data while the kernels above run
__global__ __launch_bounds__(256) void filler(unsigned *out, unsigned rounds)
{
    unsigned v[48];
#pragma unroll
    for (int k = 0; k < 48; ++k)
        v[k] = 0xdead0000u + threadIdx.x * 48u + k;
    for (unsigned r = 0; r < rounds; ++r)
#pragma unroll
        for (int k = 0; k < 48; ++k)
            v[k] = v[k] * 1664525u + v[(k + 1) % 48];
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < 48; ++k)
        s ^= v[k];
    if (s == 0x12345u)
        out[0] = s;
}

int main(int argc, char **argv)
{
    const size_t n = (argc > 1 ? (size_t)atof(argv[1]) : (size_t)64 << 20) / 256 * 256;
    const int launches = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<unsigned> h_amt(1024);
    for (int i = 0; i < 1024; ++i)
        h_amt[i] = (unsigned)(i * 7 + 1) % 31u;
    unsigned *amounts, *sink;
    unsigned long long *out;
    CK(hipMalloc(&amounts, 4096));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&out, n * 8));
    CK(hipMemcpy(amounts, h_amt.data(), 4096, hipMemcpyHostToDevice));
    hipStream_t side;
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    std::vector<unsigned long long> h(n);
    typedef void (*kernel_t)(unsigned long long *, const unsigned *);
    const struct { const char *name; kernel_t k; bool expect_wrong; } cases[] = {
        {"amount in v15, 16 VGPRs allocated (the LAST one)", amount_in_last_vgpr, true},
        {"amount in v14, 16 VGPRs allocated", amount_one_below, false},
        {"amount in v15, 24 VGPRs allocated", amount_in_v15_of_24, false}};
    bool as_described = true;
    for (const auto &c : cases) {
        unsigned long long wrong = 0, wrong_launches = 0;
        for (int it = 0; it < launches; ++it) {
            CK(hipMemsetAsync(out, 0xff, n * 8, 0));
            hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 0, side, sink, 2000u);
            hipLaunchKernelGGL(c.k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, out, amounts);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, n * 8, hipMemcpyDeviceToHost));
            unsigned long long w = 0;
            for (size_t i = 0; i < n; ++i)
                w += h[i] != (0x0000000100000001ull << h_amt[i & 1023]);
            wrong += w;
            wrong_launches += w != 0;
        }
        printf("%-52s %llu wrong results of %zu x %d (%llu of %d launches affected)\n", c.name, wrong, n, launches, wrong_launches, launches);
        as_described = as_described && ((wrong != 0) == c.expect_wrong);
    }
    printf("RESULT %s\n", as_described ? "fault_reproduced_as_described" : "NOT_as_described");
    return as_described ? 0 : 1;
}
