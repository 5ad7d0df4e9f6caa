// The pruned product kernels: the library's own variants up to 64 features (fused_variants.h).
#include "fused_knn.h"
#include "fused_variants.h"

namespace scmiac {
SCMIAC_DEFINE_PRUNED_TU(4, SCMIAC_FUSED_VARIANTS_PRUNED)
}  // namespace scmiac
