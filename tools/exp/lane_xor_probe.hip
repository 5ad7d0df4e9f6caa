#include <hip/hip_runtime.h>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
template <int J> __device__ __forceinline__ unsigned lane_xor(unsigned v, int lane) {
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false);
    else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, false);
    else if constexpr (J == 4) {
        const unsigned a = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x104, 0xF, 0xF, false);  // row_shl:4: lane i <- i + 4
        const unsigned b = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x114, 0xF, 0xF, false);  // row_shr:4: lane i <- i - 4
        return (lane & 4) ? b : a;
    } else if constexpr (J == 8) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
    else if constexpr (J == 16) {
        uint2v r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        uint2v r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}
__global__ void k(unsigned* out) {
    const int lane = threadIdx.x & 63;
    unsigned v = lane * 3 + 1000;
    out[0 * 64 + lane] = lane_xor<1>(v, lane);
    out[1 * 64 + lane] = lane_xor<2>(v, lane);
    out[2 * 64 + lane] = lane_xor<4>(v, lane);
    out[3 * 64 + lane] = lane_xor<8>(v, lane);
    out[4 * 64 + lane] = lane_xor<16>(v, lane);
    out[5 * 64 + lane] = lane_xor<32>(v, lane);
}
int main() {
    unsigned* d; hipMalloc(&d, 6 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0; const int js[6] = {1, 2, 4, 8, 16, 32};
    for (int t = 0; t < 6; ++t) for (int l = 0; l < 64; ++l) if (h[t * 64 + l] != (unsigned)((l ^ js[t]) * 3 + 1000)) { ++bad; if (bad < 10) printf("J=%d lane %d got %u\n", js[t], l, h[t*64+l]); }
    printf("bad=%d\n", bad);
    return bad != 0;
}
