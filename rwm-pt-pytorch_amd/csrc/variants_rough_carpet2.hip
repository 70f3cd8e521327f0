// Instantiates the fused PT-RWM kernel for the two-term RoughCarpet specialisation (well-separated modes).
#include "variants.h"

namespace ptrwm {
PTRWM_DEFINE_TARGET_VARIANTS(rough_carpet2_variants, RoughCarpet2);
}  // namespace ptrwm
