// Instantiates the fused PT-RWM kernel for the FullRosenbrock target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(full_rosenbrock_variants, FullRosenbrock);
}  // namespace ptrwm
