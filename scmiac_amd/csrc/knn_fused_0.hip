// Group 0 of the fused k-NN kernel's variants (fused_variants.h).
#include "fused_knn.h"
#include "fused_variants.h"

namespace scmiac {
SCMIAC_DEFINE_FUSED_TU(0, SCMIAC_FUSED_VARIANTS_0)
}  // namespace scmiac
