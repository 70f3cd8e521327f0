// Instantiates the fused PT-RWM kernel for the IIDGamma target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(iid_gamma_variants, IIDGamma);
}  // namespace ptrwm
