// score_long_inst.hip -- the exact f32 kernels for ONE padded motif length LM_LONG_M in {40, 44 ... 64}.
// Compiled once per length (build.py) with -mllvm -pragma-unroll-threshold raised: the M x M step / weight loops
// of a group only become register-indexed accumulators when they unroll completely, and LLVM's default
// budget for `#pragma unroll` (16 K unrolled instructions) ends at M ~ 40 -- past it the accumulators fall into
// scratch (2 564 scratch instructions at M = 40 without the flag, none with it).
#include "score_u8.hpp"

#ifndef LM_LONG_M
#error "LM_LONG_M must be defined"
#endif

namespace lm {

#define LM_CAT2(a, b) a##b
#define LM_CAT(a, b) LM_CAT2(a, b)

static_assert(LM_LONG_M > kMaxFastM && LM_LONG_M <= kMaxLongM && LM_LONG_M % 4 == 0, "padded long motif length");

void LM_CAT(register_score_c32_long_, LM_LONG_M)(const KernelRegistry &r)
{
    constexpr int M = LM_LONG_M;
    ScoreC32Launcher *tab = r.c32[M];
    // every kernel of this family fetches symbols with dword loads (M % 4 == 0, 4-byte aligned matrix)
    tab[MODE_STORE] = &score_c32_launch<M, MODE_STORE, 1>;
    tab[MODE_ARGMAX] = &score_c32_launch<M, MODE_ARGMAX, 1>;
    tab[MODE_THRESHOLD] = &score_c32_launch<M, MODE_THRESHOLD, 1>;
    tab[7] = tab[MODE_STORE];
    tab[8] = &score_c32_launch<M, MODE_STORE_ARGMAX, 1>;
    tab[9] = &score_c32_launch<M, MODE_CONTINUE, 1>;
    tab[11] = &score_c32_launch<M, MODE_STORE_TRACK, 1>;
    // the pair-symbol prefilter scan (score_prefilter2.hpp) of the four exact lengths that pad to M: the fused
    // threshold / argmax of 36 < M <= 64 flag candidates with it like the shorter motifs do (DNA)
    r.pre2[M - 3] = &score_c32_prefilter2_launch<M - 3>;
    r.pre2[M - 2] = &score_c32_prefilter2_launch<M - 2>;
    r.pre2[M - 1] = &score_c32_prefilter2_launch<M - 1>;
    r.pre2[M] = &score_c32_prefilter2_launch<M>;
    // the same for alphabets of more than 16 symbols (8-byte LDS reads)
    ScoreC32Launcher *tw = r.c32w[M];
    tw[MODE_STORE] = tw[7] = &score_c32_launch<M, MODE_STORE, 1, 32, 1>;
    tw[MODE_ARGMAX] = &score_c32_launch<M, MODE_ARGMAX, 1, 32, 1>;
    tw[MODE_THRESHOLD] = &score_c32_launch<M, MODE_THRESHOLD, 1, 32, 1>;
    tw[8] = &score_c32_launch<M, MODE_STORE_ARGMAX, 1, 32, 1>;
    tw[9] = &score_c32_launch<M, MODE_CONTINUE, 1, 32, 1>;
    tw[11] = &score_c32_launch<M, MODE_STORE_TRACK, 1, 32, 1>;
}

}  // namespace lm
