// Instantiates the lane-split (quad) PT-RWM kernel for the HybridRosenbrock target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(hybrid_rosenbrock_variants, QHybridRosenbrock);
}  // namespace ptrwm
