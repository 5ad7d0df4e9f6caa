// The pruned product kernels of the library's own variants for rows of more than 64 features
// (fused_variants.h).
#include "fused_knn.h"
#include "fused_variants.h"

namespace scmiac {
SCMIAC_DEFINE_PRUNED_TU(6, SCMIAC_FUSED_VARIANTS_PRUNED_WIDE)
}  // namespace scmiac
