// Instantiates the lane-split (quad) PT-RWM kernel for the RoughCarpet target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(rough_carpet_variants, QRoughCarpet);
}  // namespace ptrwm
