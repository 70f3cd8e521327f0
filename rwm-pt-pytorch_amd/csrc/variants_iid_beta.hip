// Instantiates the fused PT-RWM kernel for the IIDBeta target (all proposals, all register widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(iid_beta_variants, IIDBeta);
}  // namespace ptrwm
