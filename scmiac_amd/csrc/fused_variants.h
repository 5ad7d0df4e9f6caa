// Compiled variants of the fused k-NN kernel: (D2 = feature pairs, waves per block, reference
// rows per LDS stage, pending candidates per query, 1 = pending candidates in global memory),
// in four groups of similar compile time, one per translation unit knn_fused_<group>.hip
// (dense product kernel + diagnostic instantiation each).
// The library's own choice per feature size is fused_variant() in knn_kernels.hip; the
// others are reachable through scmiac_knn_set_tuning only (tests, tuning).
#pragma once
#define SCMIAC_FUSED_VARIANTS_0(X) \
    X(8, 12, 320, 64, 1) X(15, 12, 320, 64, 1) X(16, 12, 320, 64, 1) X(20, 12, 320, 64, 1) X(25, 12, 128, 32, 0)
#define SCMIAC_FUSED_VARIANTS_1(X) \
    X(25, 12, 320, 64, 1) X(32, 12, 256, 64, 1) X(25, 12, 128, 32, 1) X(25, 12, 256, 16, 1)
#define SCMIAC_FUSED_VARIANTS_2(X) \
    X(40, 12, 128, 64, 1) X(50, 12, 128, 64, 1) X(64, 8, 128, 64, 1) X(25, 12, 256, 32, 1) X(25, 12, 320, 32, 1)
#define SCMIAC_FUSED_VARIANTS_3(X) \
    X(80, 8, 64, 64, 1) X(96, 8, 64, 64, 1) X(100, 8, 64, 64, 1) X(112, 8, 64, 64, 1) X(128, 8, 64, 64, 1)
// the library's own variants up to 64 features also exist as pruned product kernels (unit 4) and as
// stream-sharing kernels (unit 5, fused_knn_share.h) ...
#define SCMIAC_FUSED_VARIANTS_PRUNED(X) \
    X(8, 12, 320, 64, 1) X(15, 12, 320, 64, 1) X(16, 12, 320, 64, 1) X(20, 12, 320, 64, 1) X(25, 12, 320, 64, 1) \
    X(32, 12, 256, 64, 1)
// ... and so do the library's own variants for wider rows (unit 6: pruned product kernels only)
#define SCMIAC_FUSED_VARIANTS_PRUNED_WIDE(X) \
    X(40, 12, 128, 64, 1) X(50, 12, 128, 64, 1) X(64, 8, 128, 64, 1) X(80, 8, 64, 64, 1) X(96, 8, 64, 64, 1) \
    X(100, 8, 64, 64, 1) X(112, 8, 64, 64, 1) X(128, 8, 64, 64, 1)
#define SCMIAC_FUSED_VARIANTS(X) \
    SCMIAC_FUSED_VARIANTS_0(X) SCMIAC_FUSED_VARIANTS_1(X) SCMIAC_FUSED_VARIANTS_2(X) SCMIAC_FUSED_VARIANTS_3(X)
