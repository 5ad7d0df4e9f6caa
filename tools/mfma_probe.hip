// Probe of the fp32 MFMA pipe on gfx950: dependent v_mfma_f32_32x32x2_f32 chains with
// optional LDS fragment reads and VALU work mixed in, at 1-4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: MFMA only (one chain)  1: two alternating chains  2: + LDS reads  3: + LDS reads + VALU
__global__ __launch_bounds__(1024) void probe(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 128 * 60; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    float a[28], b[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) { a[i] = seed + i + lane; b[i] = seed - i; }
    float vmin = 1e30f;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) {
            const float* p = lds + ((it & 3) * 32 + (lane & 31)) * 60 + (lane >> 5) * 28;
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(
                    (const __attribute__((address_space(3))) void*)(p + 4 * c));
                a[4 * c] = v[0]; a[4 * c + 1] = v[1]; a[4 * c + 2] = v[2]; a[4 * c + 3] = v[3];
            }
        }
#pragma unroll
        for (int s = 0; s < 25; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc0, 0, 0, 0);
        if (MODE >= 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float k = __uint_as_float((__float_as_uint(acc1[r]) & 0xfffffff0u) | r);
                vmin = fminf(vmin, k);
            }
        }
        if (MODE >= 1) {
#pragma unroll
            for (int s = 0; s < 25; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc1, 0, 0, 0);
            if (MODE >= 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float k = __uint_as_float((__float_as_uint(acc0[r]) & 0xfffffff0u) | r);
                    vmin = fminf(vmin, k);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 25; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc0, 0, 0, 0);
        }
    }
    float r = vmin;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(int waves_per_simd, float* out) {
    const int iters = 4000;
    dim3 grid(256), block(waves_per_simd * 4 * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, grid, block, 128 * 60 * 4, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, grid, block, 128 * 60 * 4, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * waves_per_simd * 4 * iters * 50.0 * 4096.0;
    printf("mode %d waves/SIMD %d: %8.3f ms  %7.2f TFLOP/s (%5.1f %% of 157.3)\n", MODE, waves_per_simd, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int w = 1; w <= 4; ++w) { run<0>(w, out); run<1>(w, out); run<2>(w, out); run<3>(w, out); }
    hipDeviceSynchronize();
    return 0;
}
