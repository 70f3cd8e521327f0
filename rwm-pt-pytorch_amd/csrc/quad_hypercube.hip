// Instantiates the lane-split (quad) PT-RWM kernel for the Hypercube target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(hypercube_variants, QHypercube);
}  // namespace ptrwm
