// Group 3 of the fused k-NN kernel's variants (fused_variants.h).
#include "fused_knn.h"
#include "fused_variants.h"

namespace scmiac {
SCMIAC_DEFINE_FUSED_TU(3, SCMIAC_FUSED_VARIANTS_3)
}  // namespace scmiac
