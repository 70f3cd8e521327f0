// Instantiates the lane-split (quad) PT-RWM kernel for the ThreeMixture target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(three_mixture_variants, QThreeMixture);
}  // namespace ptrwm
