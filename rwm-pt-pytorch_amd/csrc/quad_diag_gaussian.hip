// Instantiates the lane-split (quad) PT-RWM kernel for the DiagGaussian target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(diag_gaussian_variants, QDiagGaussian);
}  // namespace ptrwm
