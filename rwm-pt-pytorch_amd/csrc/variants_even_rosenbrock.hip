// Instantiates the fused PT-RWM kernel for the EvenRosenbrock target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(even_rosenbrock_variants, EvenRosenbrock);
}  // namespace ptrwm
