// Tuning aid: which workgroups of a 1024-workgroup launch (256 threads, 36 KB of LDS: four per CU, the shape of the
// one-role GEMM) share a compute unit?  Every workgroup records (XCC_ID, HW_ID) - tools/bin/cu_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
    __shared__ float lds[9216];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float x = lds[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;      // stay resident so that all 1024 workgroups coexist
    if (threadIdx.x == 0) out[blockIdx.x * 2] = hw, out[blockIdx.x * 2 + 1] = xcc;
    if (x == 12345.f) out[0] = 0;
}
int main() {
    const int G = 1024;
    unsigned* d;
    hipMalloc(&d, G * 8);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 2);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
        cu[key].push_back(b);
    }
    printf("%zu distinct (xcc, se, sh, cu) keys for %d workgroups\n", cu.size(), G);
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ < 12) {
            printf("xcc %u se %u sh %u cu %2u :", kv.first >> 16, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 0xf);
            for (int b : kv.second) printf(" %4d", b);
            printf("\n");
        }
    }
    // histogram of id differences between workgroups that share a CU
    std::map<int, int> diff;
    for (auto& kv : cu)
        for (size_t i = 1; i < kv.second.size(); ++i) diff[kv.second[i] - kv.second[i - 1]]++;
    printf("differences between consecutive ids on one CU:");
    for (auto& d2 : diff) printf("  %d x%d", d2.first, d2.second);
    printf("\n");
    return 0;
}
