// Instantiates the lane-split (quad) PT-RWM kernel for the ThreeMixture1 target (means differing in the first coordinate only).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(three_mixture1_variants, QThreeMixture1);
}  // namespace ptrwm
