// score_pair_inst.hip -- the DNA pair-symbol prefilter scan (score_prefilter2.hpp) for the motif lengths
// LM_PAIR_LO .. LM_PAIR_HI beyond the exact kernel families (kMaxLongM < M <= kMaxPairM): the fused threshold /
// argmax scans of such motifs flag their candidates with it; the exact re-scoring is length-generic.
#include "score_u8.hpp"

#if !defined(LM_PAIR_LO) || !defined(LM_PAIR_HI)
#error "LM_PAIR_LO / LM_PAIR_HI must be defined"
#endif

namespace lm {

#define LM_CAT2(a, b) a##b
#define LM_CAT(a, b) LM_CAT2(a, b)

static_assert(LM_PAIR_LO > kMaxLongM && LM_PAIR_HI <= kMaxPairM && LM_PAIR_LO <= LM_PAIR_HI, "pair-scan-only lengths");

template <int M>
struct RegisterPairRange {
    static void run(const KernelRegistry &r)
    {
        r.pre2[M] = &score_c32_prefilter2_launch<M>;
        if constexpr (M < LM_PAIR_HI)
            RegisterPairRange<M + 1>::run(r);
    }
};

void LM_CAT(register_score_pair_, LM_PAIR_LO)(const KernelRegistry &r) { RegisterPairRange<LM_PAIR_LO>::run(r); }

}  // namespace lm
