// Instantiates the lane-split (quad) PT-RWM kernel for the IIDBeta target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(iid_beta_variants, QIIDBeta);
}  // namespace ptrwm
