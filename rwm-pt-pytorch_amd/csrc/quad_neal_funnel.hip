// Instantiates the lane-split (quad) PT-RWM kernel for the NealFunnel target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(neal_funnel_variants, QNealFunnel);
}  // namespace ptrwm
