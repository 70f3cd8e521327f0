// Instantiates the fused PT-RWM kernel for the RoughCarpet target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(rough_carpet_variants, RoughCarpet);
}  // namespace ptrwm
