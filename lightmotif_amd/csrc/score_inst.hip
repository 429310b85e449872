// score_inst.hip -- instantiates score_c32<M, MODE> for M in [LM_M_LO, LM_M_HI].
// Compiled several times with different -DLM_M_LO/-DLM_M_HI/-DLM_INST_ID so the
// fully unrolled kernels build in parallel (see build.py).
#include "score_u8.hpp"

#ifndef LM_M_LO
#error "LM_M_LO / LM_M_HI / LM_INST_ID must be defined"
#endif

namespace lm {

#define LM_CAT2(a, b) a##b
#define LM_CAT(a, b) LM_CAT2(a, b)

template <int M>
struct RegisterRange {
    static void run(const KernelRegistry &r)
    {
        r.pre[M] = &score_c32_prefilter_launch<M>;
        r.u8[M] = &score_c32_u8_launch<M>;  // DiscreteMatrix scores (score_u8.hpp)
        if constexpr (M >= 2) {
            r.pre2[M] = &score_c32_prefilter2_launch<M>;
            r.pre2_protein[M] = &score_c32_prefilter2_launch<M, 21>;
            r.u8_pairs[M] = &score_c32_u8_pairs_launch<M>;
            if constexpr (prefilter2_multi(M) > 1)
                r.pre2_multi[M] = &score_c32_prefilter2_multi_launch<M>;
        }
        ScoreC32Launcher *tab = r.c32[M];
        tab[MODE_STORE] = &score_c32_launch<M, MODE_STORE>;
        tab[MODE_ARGMAX] = &score_c32_launch<M, MODE_ARGMAX>;
        tab[MODE_THRESHOLD] = &score_c32_launch<M, MODE_THRESHOLD>;
        if constexpr (M % 4 == 0) {
            tab[7] = &score_c32_launch<M, MODE_STORE, 1>;
            tab[10] = &score_c32_launch<M, MODE_STORE, 1, 16>;
            tab[11] = &score_c32_launch<M, MODE_STORE_TRACK, 1>;
        }
        tab[8] = &score_c32_launch<M, MODE_STORE_ARGMAX, 1>;
        tab[9] = &score_c32_launch<M, MODE_CONTINUE, 1>;
        // wide alphabets (protein): the slots the launchers use at C = 32
        r.prew[M] = &score_c32_prefilter_launch<M, 1>;
        r.preblk[M] = &score_c32_prefilter_blk_launch<M>;
        r.u8w[M] = &score_c32_u8_launch<M, 1>;
        ScoreC32Launcher *tw = r.c32w[M];
        tw[MODE_STORE] = &score_c32_launch<M, MODE_STORE, 0, 32, 1>;
        tw[MODE_ARGMAX] = &score_c32_launch<M, MODE_ARGMAX, 0, 32, 1>;
        tw[MODE_THRESHOLD] = &score_c32_launch<M, MODE_THRESHOLD, 0, 32, 1>;
        if constexpr (M % 4 == 0) {
            tw[7] = &score_c32_launch<M, MODE_STORE, 1, 32, 1>;
            tw[10] = &score_c32_launch<M, MODE_STORE, 1, 16, 1>;
            tw[11] = &score_c32_launch<M, MODE_STORE_TRACK, 1, 32, 1>;
        }
        tw[8] = &score_c32_launch<M, MODE_STORE_ARGMAX, 1, 32, 1>;
        tw[9] = &score_c32_launch<M, MODE_CONTINUE, 1, 32, 1>;
        if constexpr (M < LM_M_HI)
            RegisterRange<M + 1>::run(r);
    }
};

void LM_CAT(register_score_c32_, LM_INST_ID)(const KernelRegistry &r) { RegisterRange<LM_M_LO>::run(r); }

}  // namespace lm
