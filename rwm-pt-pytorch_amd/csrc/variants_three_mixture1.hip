// Instantiates the fused PT-RWM kernel for the ThreeMixture specialisation whose means differ in the first coordinate only.
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(three_mixture1_variants, ThreeMixture1);
}  // namespace ptrwm
