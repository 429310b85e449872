// score_xlong_inst.hip -- the plain f32 STORE kernel for ONE padded motif length LM_XLONG_M in {72, 80, 88}.
// Motifs of 65 ... 88 rows are padded with leading zero rows to the next multiple of 8 and stored in ONE pass:
// against a bound of two wavefronts per SIMD the accumulators still fit the register file (217-218 VGPRs at M' = 80 / 88,
// no scratch), and one LDS-bound pass beats two slices with a 4 + 4 B/pos round trip between them (1 Gbp: M = 65 3.06 ->
// 2.38 ms, 72 3.15 -> 2.46, 80 3.35 -> 2.92, 88 3.67 -> 3.12).  That is where it ends: M' = 96 / 104 still compile
// without (much) scratch at two wavefronts (256 / 234 VGPRs) but run slower than the slices (4.25 vs 3.90, 5.28 vs
// 4.05 ms -- two wavefronts no longer hide the gather latency of 24+ b128 reads per step), one wavefront per SIMD (262 VGPRs
// at M' = 96, 331 at 128) is slower again (7.9 / 5.3 ... 6.9 ms), and from M' = 112 the bound of two spills by the hundred
// (profiles/r03_xlong_ab.txt).  Compiled with the raised unroll budget of the long family (score_long_inst.hip).
#include "score_u8.hpp"

#ifndef LM_XLONG_M
#error "LM_XLONG_M must be defined"
#endif

namespace lm {

#define LM_CAT2(a, b) a##b
#define LM_CAT(a, b) LM_CAT2(a, b)

static_assert(LM_XLONG_M > kMaxLongM && LM_XLONG_M <= kMaxStoreM && LM_XLONG_M % 8 == 0, "padded very long motif length");

void LM_CAT(register_score_c32_xlong_, LM_XLONG_M)(const KernelRegistry &r)
{
    constexpr int M = LM_XLONG_M;
    ScoreC32Launcher *tab = r.c32[M];
    tab[MODE_STORE] = tab[7] = &score_c32_launch<M, MODE_STORE, 1>;  // dword symbol loads
    // alphabets of more than 16 symbols (8-byte LDS reads; 148 VGPRs at M' = 72, 203 at 88)
    ScoreC32Launcher *tw = r.c32w[M];
    tw[MODE_STORE] = tw[7] = &score_c32_launch<M, MODE_STORE, 1, 32, 1>;
}

}  // namespace lm
