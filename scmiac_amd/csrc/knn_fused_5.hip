// The stream-sharing kernels: the library's own variants up to 64 features (fused_variants.h).
#include "fused_knn_share.h"
#include "fused_variants.h"

namespace scmiac {
SCMIAC_DEFINE_SHARE_TU(5, SCMIAC_FUSED_VARIANTS_PRUNED)
}  // namespace scmiac
