// Instantiates the fused PT-RWM kernel for the HybridRosenbrock target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(hybrid_rosenbrock_variants, HybridRosenbrock);
}  // namespace ptrwm
