// Instantiates the lane-split (quad) PT-RWM kernel for the RoughCarpet2 (two-term) target (all proposals, all lane widths).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_QUAD_VARIANTS(rough_carpet2_variants, QRoughCarpet2);
}  // namespace ptrwm
