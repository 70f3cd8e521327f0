// Instantiates the lane-split (quad) PT-RWM kernel for the FullRosenbrock target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(full_rosenbrock_variants, QFullRosenbrock);
}  // namespace ptrwm
