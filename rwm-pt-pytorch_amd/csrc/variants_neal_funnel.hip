// Instantiates the fused PT-RWM kernel for the NealFunnel target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(neal_funnel_variants, NealFunnel);
}  // namespace ptrwm
