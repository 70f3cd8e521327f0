// Instantiates the lane-split (quad) PT-RWM kernel for the EvenRosenbrock target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(even_rosenbrock_variants, QEvenRosenbrock);
}  // namespace ptrwm
