// Do the vector-memory path and the matrix pipe of a CU overlap?  One 512-thread workgroup per CU: waves 0-3 multiply (24
// v_mfma_f32_32x32x16_bf16 per "step" on register operands, no LDS, no barrier), waves 4-7 stream conv2's operand shape (20 KB per
// step: A fp32 rows + W planes) either into VGPRs (global_load_dwordx4) or into LDS (global_load_lds_dwordx4, no VGPR written).
// Each role alone, then both together: cycles per step of each role.
// hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/bin/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 512, KT = K / 16, ROWS = 10752 * 2, NCOL = 512;

// MODE bit 0: MFMA waves work; bit 1: loader waves work; DMA: loads go to LDS
template <int MODE, bool DMA>
__global__ __launch_bounds__(512, 2) void probe(const float* __restrict__ A, const u32x4* __restrict__ W, uint32_t* out, unsigned long long* cyc,
                                                int steps_per_wg) {
    __shared__ u32x4 lds[4 * 1280];            // 80 KB: four 20 KB landing zones for the DMA form
    const int role = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (role == 0) {
        if (!(MODE & 1)) return;
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        bf16x8 a[2][3], b[2][3];
        for (int u = 0; u < 2; ++u) for (int p = 0; p < 3; ++p) {
            const uint32_t s = (lane * 7 + u * 3 + p) * 2654435761u;
            const u32x4 v = {0x3E003E00u ^ (s & 0x807F807Fu), 0x3D803F00u ^ ((s >> 3) & 0x807F807Fu), 0x3E103E20u ^ (s & 0x007F007Fu), 0xBE003D00u ^ ((s >> 5) & 0x007F007Fu)};
            a[u][p] = __builtin_bit_cast(bf16x8, v), b[u][p] = __builtin_bit_cast(bf16x8, v ^ 0x00010001u);
        }
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
        for (int s = 0; s < steps_per_wg; ++s) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[k]], b[j][PB[k]], acc[i][j], 0, 0, 0);
        }
        float sum = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 12345.678f) out[1] = 1;
        if (tid == 0) atomicAdd(&cyc[0], __builtin_readcyclecounter() - t0), atomicAdd(&cyc[1], (unsigned long long)steps_per_wg);
    } else {
        if (!(MODE & 2)) return;
        u32x4 accv = {0, 0, 0, 0};
        const int crow = tid >> 2, kq = tid & 3;
        int done = 0;
        for (int tile = blockIdx.x; done < steps_per_wg; tile = (tile + gridDim.x) % ((ROWS / 128) * (NCOL / 128))) {
            const int total = (ROWS / 128) * (NCOL / 128), q = total >> 3, xcd = tile & 7, local = tile >> 3;
            const int wg = xcd * q + local;
            const int m0 = (wg / 4) * 128, n0 = (wg % 4) * 128;
            const float* a0 = A + (size_t)(m0 + crow) * K + kq * 4;
            const float* a1 = A + (size_t)(m0 + crow + 64) * K + kq * 4;
            const u32x4* w[3];
            for (int i = 0; i < 3; ++i) {
                const int ch = tid + i * 256, wrow = ch / 6, wch = ch - wrow * 6;
                w[i] = W + (size_t)(n0 + wrow) * (3 * K / 8) + wch;
            }
            if constexpr (!DMA) {
#pragma unroll 4
                for (int s = 0; s < KT; ++s) {
                    accv ^= __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(a0 + s * 16));
                    accv ^= __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(a1 + s * 16));
                    for (int i = 0; i < 3; ++i) accv ^= w[i][s * 6];
                }
            } else {
#if defined(__HIP_DEVICE_COMPILE__)
                for (int s = 0; s < KT; ++s) {
                    u32x4* zone = lds + (s & 3) * 1280 + (tid >> 6) * 320;          // this wave's 5 KB of the landing zone
                    __builtin_amdgcn_global_load_lds(a0 + s * 16, (__attribute__((address_space(3))) void*)(zone), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(a1 + s * 16, (__attribute__((address_space(3))) void*)(zone + 64), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(w[0] + s * 6, (__attribute__((address_space(3))) void*)(zone + 128), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(w[1] + s * 6, (__attribute__((address_space(3))) void*)(zone + 192), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(w[2] + s * 6, (__attribute__((address_space(3))) void*)(zone + 256), 16, 0, 0);
                    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");               // three steps in flight
                }
#endif
            }
            done += KT;
        }
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            accv ^= lds[tid];
        }
        if (accv[0] == 0x12345678u && accv[1] == 77u) out[0] = accv[2] ^ accv[3];
        if (tid == 0) atomicAdd(&cyc[2], __builtin_readcyclecounter() - t0), atomicAdd(&cyc[3], (unsigned long long)done);
    }
}

template <int MODE, bool DMA>
static void run(const char* name, const float* A, const u32x4* W, uint32_t* out, unsigned long long* cyc) {
    const int steps = 64 * 20;
    hipLaunchKernelGGL((probe<MODE, DMA>), dim3(256), dim3(512), 0, 0, A, W, out, cyc, 64);
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, DMA>), dim3(256), dim3(512), 0, 0, A, W, out, cyc, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4];
    hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-58s %7.1f us | MFMA waves %6.0f cycles per step (768 = the pipe's rate) | loader waves %6.0f cycles per 20 KB step\n", name, ms * 1e3,
           h[1] ? (double)h[0] / h[1] : 0.0, h[3] ? (double)h[2] / h[3] : 0.0);
}

int main() {
    float* A;
    u32x4* W;
    uint32_t* out;
    unsigned long long* cyc;
    hipMalloc(&A, (size_t)ROWS * K * 4), hipMemset(A, 0x3c, (size_t)ROWS * K * 4);
    hipMalloc(&W, (size_t)NCOL * 3 * K * 2), hipMemset(W, 0x3d, (size_t)NCOL * 3 * K * 2);
    hipMalloc(&out, 64), hipMalloc(&cyc, 64);
    run<1, false>("MFMA waves alone", A, W, out, cyc);
    run<2, false>("loader waves alone, loads into VGPRs", A, W, out, cyc);
    run<3, false>("both, loads into VGPRs", A, W, out, cyc);
    run<2, true>("loader waves alone, LDS-DMA", A, W, out, cyc);
    run<3, true>("both, LDS-DMA", A, W, out, cyc);
    return 0;
}
