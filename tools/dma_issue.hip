// Tuning aid: how long does a wavefront take to ISSUE a group of four 1 KiB operand loads (8 rows x 128 B each, the
// access shape of the 64 x 64 GEMM tiles), as LDS-DMA (global_load_lds_dwordx4, M0 rewritten per instruction or held
// constant with instruction offsets) and as plain global_load_dwordx4, for several row strides and wavefronts per CU?
// tools/bin/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: LDS-DMA, M0 per instruction; 1: LDS-DMA, one M0 + instruction offsets; 2: loads into VGPRs
__global__ __launch_bounds__(256) void k(const float* buf, long row_stride_f, int rows_total, int steps, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * 1024];   // 4 waves x 2 stages x 4 KiB
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid;
    unsigned long long t_issue = 0, t_wait = 0;
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; ++s) {
        const float* src[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ((wave * 4 + i) * 8 + (lane >> 3) + s * 37) % rows_total;
            src[i] = buf + (size_t)row * row_stride_f + (s % 4) * 32 + (lane & 7) * 4;
        }
        float* stage = lds + (wid * 2 + (s & 1)) * 1024;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds(src[i], (__attribute__((address_space(3))) void*)(stage + i * 256), 16, 0, 0);
        } else if constexpr (MODE == 1) {
            __builtin_amdgcn_global_load_lds(src[0], (__attribute__((address_space(3))) void*)stage, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(src[1] - 256, (__attribute__((address_space(3))) void*)stage, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds(src[2] - 512, (__attribute__((address_space(3))) void*)stage, 16, 2048, 0);
            __builtin_amdgcn_global_load_lds(src[3] - 768, (__attribute__((address_space(3))) void*)stage, 16, 3072, 0);
        } else {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(src[i]) : "memory");
            const unsigned long long t1 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t2 = __builtin_readcyclecounter();
            t_issue += t1 - t0, t_wait += t2 - t1;
#pragma unroll
            for (int i = 0; i < 4; ++i) sink += v[i];
            continue;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        t_issue += t1 - t0, t_wait += t2 - t1;
    }
    if (lane == 0) {
        atomicAdd(&out[0], t_issue), atomicAdd(&out[1], t_wait), atomicAdd(&out[2], 1ull);
        if (sink[0] == 12345.f) out[3] = 1;
    }
    if (lane == 1 && lds[threadIdx.x] == 12345.f) out[3] = 2;
}

template <int MODE>
void run(const float* buf, long stride_f, int rows, int wgs_per_cu, unsigned long long* out) {
    const int steps = 64;
    hipMemset(out, 0, 32);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, buf, stride_f, rows, steps, out);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, buf, stride_f, rows, steps, out);
    hipMemset(out, 0, 32);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, buf, stride_f, rows, steps, out);
    unsigned long long h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("  mode %d  %2d waves/CU: issue of 4 loads %7.0f cycles, wait for them %7.0f cycles\n", MODE, 4 * wgs_per_cu,
           (double)h[0] / h[2] / steps, (double)h[1] / h[2] / steps);
}

int main() {
    unsigned long long* out;
    hipMalloc(&out, 32);
    float* buf;
    const size_t bytes = 64ull << 20;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    for (long stride_b : {512L, 2048L, 4608L}) {
        for (int rows : {512, 4096}) {                  // footprint: rows x 128..512 B touched -> L2-resident
            if ((size_t)rows * stride_b > bytes) continue;
            printf("row stride %ld B, %d rows (%.1f MB span)\n", stride_b, rows, rows * stride_b / 1e6);
            for (int w : {1, 2, 4}) {
                run<0>(buf, stride_b / 4, rows, w, out);
                run<1>(buf, stride_b / 4, rows, w, out);
                run<2>(buf, stride_b / 4, rows, w, out);
            }
        }
    }
    return 0;
}
