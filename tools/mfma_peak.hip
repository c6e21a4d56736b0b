// Calibration: what the fp32 matrix cores of this MI355X deliver for v_mfma_f32_32x32x2_f32 in (a) a bare
// register loop, (b) the same loop with the GEMM's LDS fragment reads, (c) with a barrier per 64 MFMAs.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 36];
    for (int i = threadIdx.x; i < 2 * 256 * 36; i += 256) lds[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 a[2] = {{1.f, 2.f, 3.f, 4.f}, {1.5f, 2.5f, 3.5f, 4.5f}}, b[2] = {{.1f, .2f, .3f, .4f}, {.5f, .6f, .7f, .8f}};
    const float* af = lds + ((w >> 1) * 64 + r) * 36 + 4 * h;
    const float* bf = lds + 128 * 36 + ((w & 1) * 64 + r) * 36 + 4 * h;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE >= 1) {
                a[0] = *reinterpret_cast<const f32x4*>(af + q * 8);
                a[1] = *reinterpret_cast<const f32x4*>(af + 32 * 36 + q * 8);
                b[0] = *reinterpret_cast<const f32x4*>(bf + q * 8);
                b[1] = *reinterpret_cast<const f32x4*>(bf + 32 * 36 + q * 8);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-34s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, best, flops / best / 1e9);
    hipFree(out);
}

int main() {
    const int it = 400;
    run<0>("bare MFMA loop, 1 wave/SIMD", 256, it);
    run<0>("bare MFMA loop, 2 waves/SIMD", 512, it);
    run<1>("+ ds_read_b128 fragments, 1 w/SIMD", 256, it);
    run<1>("+ ds_read_b128 fragments, 2 w/SIMD", 512, it);
    run<2>("+ barrier per 64 MFMAs, 1 w/SIMD", 256, it);
    run<2>("+ barrier per 64 MFMAs, 2 w/SIMD", 512, it);
    run<2>("+ barrier per 64 MFMAs, 4 rounds", 2048, it);
    return 0;
}
