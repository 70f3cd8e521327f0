// Instantiates the fused PT-RWM kernel for the ThreeMixture target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(three_mixture_variants, ThreeMixture);
}  // namespace ptrwm
