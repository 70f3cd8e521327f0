// Instantiates the lane-split (quad) PT-RWM kernel for the IIDGamma target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(iid_gamma_variants, QIIDGamma);
}  // namespace ptrwm
