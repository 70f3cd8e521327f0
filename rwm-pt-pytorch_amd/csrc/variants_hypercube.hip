// Instantiates the fused PT-RWM kernel for the Hypercube target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(hypercube_variants, Hypercube);
}  // namespace ptrwm
