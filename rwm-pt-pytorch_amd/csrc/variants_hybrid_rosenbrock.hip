// Instantiates the fused PT-RWM kernel for the HybridRosenbrock target (all proposals, all register widths).
#define PTRWM_TU_EXTRA_DIMS  // this target also has kernels for PTRWM_WIDTHS_EXTRA (variants.h): its data dims 9, 19, 29
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(hybrid_rosenbrock_variants, HybridRosenbrock);
}  // namespace ptrwm
