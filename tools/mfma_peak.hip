// Calibration: what the fp32 matrix cores of this MI355X deliver for v_mfma_f32_32x32x2_f32 in (a) a bare
// register loop, (b) the same loop with the GEMM's LDS fragment reads, (c) with a barrier per 64 MFMAs.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 3: + 8 global_load_dwordx4 per thread per step (A streams, W re-read), results kept live
// MODE 4: + the 8 ds_write_b128 that publish them to the other LDS stage (the real GEMM's step)
static __device__ int POOL = 512;
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const float* __restrict__ gA = nullptr,
                                            const float* __restrict__ gW = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 36];
    for (int i = threadIdx.x; i < 2 * 256 * 36; i += 256) lds[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 a[2] = {{1.f, 2.f, 3.f, 4.f}, {1.5f, 2.5f, 3.5f, 4.5f}}, b[2] = {{.1f, .2f, .3f, .4f}, {.5f, .6f, .7f, .8f}};
    const float* af = lds + ((w >> 1) * 64 + r) * 36 + 4 * h;
    const float* bf = lds + 128 * 36 + ((w & 1) * 64 + r) * 36 + 4 * h;
    const int crow = threadIdx.x >> 3, kc = (threadIdx.x & 7) * 4;
    f32x4 sa[4], sw[4];
    for (int i = 0; i < 4; ++i) sa[i] = sw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa = gA ? gA + ((size_t)(blockIdx.x % POOL) * 128 + crow) * 4096 + kc : nullptr;
    const float* pw = gW ? gW + (size_t)crow * 4096 + kc : nullptr;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3) {
            const int k0 = (it * 32) & 4095;
#pragma unroll
            for (int i = 0; i < 4; ++i) sa[i] = *reinterpret_cast<const f32x4*>(pa + (size_t)i * 32 * 4096 + k0);
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[i] = *reinterpret_cast<const f32x4*>(pw + (size_t)i * 32 * 4096 + k0);
        }
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
        if (MODE >= 4) {
            float* nst = lds + ((it + 1) & 1) * 0;      // same stage: the calibration only needs the traffic
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(nst + (crow + 32 * i) * 36 + kc) = sa[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(nst + 128 * 36 + (crow + 32 * i) * 36 + kc) = sw[i];
        }
        if (MODE >= 2) __syncthreads();
    }
    if (MODE == 3) for (int i = 0; i < 4; ++i) acc[0][0][i] += sa[i][0] + sw[i][1];
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    static float *gA = nullptr, *gW = nullptr;
    if (!gA) {
        hipMalloc(&gA, (size_t)512 * 128 * 4096 * 4);      // 1 GiB A panel pool (streams from HBM / MALL)
        hipMalloc(&gW, (size_t)128 * 4096 * 4);            // 2 MiB W panel (L2 resident)
        hipMemset(gA, 0, (size_t)512 * 128 * 4096 * 4);
        hipMemset(gW, 0, (size_t)128 * 4096 * 4);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, gA, gW);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, gA, gW);
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

int main(int argc, char** argv) {
    const int it = 400;
    int pool = argc > 1 ? atoi(argv[1]) : 512;         // A panels of 2 MiB each the workgroups stream from
    hipMemcpyToSymbol(HIP_SYMBOL(POOL), &pool, sizeof(int));
    printf("A pool = %d panels (%d MiB)\n", pool, pool * 2);
    run<0>("bare MFMA loop, 1 wave/SIMD", 256, it);
    run<0>("bare MFMA loop, 2 waves/SIMD", 512, it);
    run<1>("+ ds_read_b128 fragments, 1 w/SIMD", 256, it);
    run<1>("+ ds_read_b128 fragments, 2 w/SIMD", 512, it);
    run<2>("+ barrier per 64 MFMAs, 1 w/SIMD", 256, it);
    run<2>("+ barrier per 64 MFMAs, 2 w/SIMD", 512, it);
    run<2>("+ barrier per 64 MFMAs, 4 rounds", 2048, it);
    run<3>("+ 8 global loads / step, 2 w/SIMD", 512, it);
    run<4>("+ 8 ds_write_b128 / step, 2 w/SIMD", 512, it);
    run<4>("+ 8 ds_write_b128 / step, 4 rounds", 2048, it);
    return 0;
}
