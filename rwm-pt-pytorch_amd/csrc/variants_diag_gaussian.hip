// Instantiates the fused PT-RWM kernel for the DiagGaussian target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(diag_gaussian_variants, DiagGaussian);
}  // namespace ptrwm
