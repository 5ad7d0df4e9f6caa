// Probe: how long does VALU work of one wave take while the other waves of its SIMD run
// back-to-back fp32 MFMA chains?  Groups of G independent VALU instructions; each group
// depends on the previous one.  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o valu_latency_probe tools/valu_latency_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int G, int MFMA_GAP>
__global__ __launch_bounds__(768) void probe(unsigned long long* out, float* sink, int iters, float seed, int mfma_waves) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {  // one timing wave per SIMD (waves are dealt round-robin to the 4 SIMDs)
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = seed + lane + i;
        float z = seed;
        unsigned long long total = 0;
        for (int it = 0; it < iters; ++it) {
            unsigned long long t0 = clock64();
#pragma unroll
            for (int g = 0; g < 32; ++g) {
#pragma unroll
                for (int i = 0; i < G; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(z));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(z) : "v"(x[G - 1]), "v"(seed));
            }
            unsigned long long t1 = clock64();
            total += t1 - t0;
            __builtin_amdgcn_s_sleep(20);
        }
        if (lane == 0) out[blockIdx.x * 4 + wave] = total;
        float r = z;
        for (int i = 0; i < 16; ++i) r += x[i];
        sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    } else if (wave < 4 + mfma_waves) {
        f32x16 acc = {0};
        float a = seed + lane, b = seed - lane;
        for (int it = 0; it < iters * 6; ++it) {
#pragma unroll
            for (int s = 0; s < 25; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                if (MFMA_GAP && (s % 5) == 4) asm volatile("s_nop 0");
            }
        }
        float r = 0;
        for (int i = 0; i < 16; ++i) r += acc[i];
        sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    }
}

template <int G, int MFMA_GAP>
void run(int mfma_waves, unsigned long long* out, float* sink) {
    const int iters = 1000;
    hipLaunchKernelGGL((probe<G, MFMA_GAP>), dim3(256), dim3(768), 0, 0, out, sink, iters, 1.0f, mfma_waves);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < 1024; ++i) sum += (double)h[i];
    printf("G %2d gap %d mfma waves/SIMD %d: %8.1f cycles per group (%d+1 VALU)\n", G, MFMA_GAP, mfma_waves / 4,
           sum / 1024 / iters / 32, G);
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 1024 * 8); hipMalloc(&sink, 256 * 768 * 4);
    for (int mw = 0; mw <= 8; mw += 4) {
        run<1, 0>(mw, out, sink); run<4, 0>(mw, out, sink); run<16, 0>(mw, out, sink);
        run<1, 1>(mw, out, sink); run<16, 1>(mw, out, sink);
    }
    return 0;
}
