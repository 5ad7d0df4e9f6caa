// Probe: a wave that interleaves a DEPENDENT VALU chain with its own MFMA chain, next to
// waves that issue MFMAs only.  Does the dependent VALU work still wait "turns", or does it
// ride along with the wave's own MFMA issue?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o interleave_probe tools/interleave_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VALU_PER_MFMA>
__global__ __launch_bounds__(768) void probe(unsigned long long* out, float* sink, int iters, float seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc = {0};
    float a = seed + lane, b = seed - lane, x = seed;
    unsigned long long t0 = clock64();
    if (wave < 4) {  // one mixed wave per SIMD
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 25; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VALU_PER_MFMA; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 25; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    unsigned long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 12 + wave] = t1 - t0;
    float r = x;
    for (int i = 0; i < 16; ++i) r += acc[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V>
void run(unsigned long long* out, float* sink) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<V>), dim3(256), dim3(768), 0, 0, out, sink, iters, 1.0f);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 12];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double mixed = 0, pure = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 12; ++w) (w < 4 ? mixed : pure) += (double)h[b * 12 + w];
    mixed /= 256.0 * 4 * iters * 25; pure /= 256.0 * 8 * iters * 25;
    printf("dependent VALU per MFMA %d: mixed wave %.1f cycles per MFMA, MFMA-only partners %.1f (ideal 3 x 64 = 192)\n", V, mixed, pure);
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 12 * 8); hipMalloc(&sink, 256 * 768 * 4);
    run<0>(out, sink); run<1>(out, sink); run<2>(out, sink); run<4>(out, sink); run<8>(out, sink);
    return 0;
}
