// Probe for the split-bf16 GEMM's inner loop (csrc/gemm_split.hpp, gemm_split_big_kernel): the same six exact partial
// products per 16 k of a 64 x 64 wave tile, fragments re-read from LDS every step, issued as
//   (A) 24 x v_mfma_f32_32x32x16_bf16 fed by 12 ds_read_b128            (what the kernel does today)
//   (B) 48 x v_mfma_f32_16x16x32_bf16 fed by 20 ds_read_b128, two products per instruction:
//       [a_hi | a_lo] . [b_lo | b_hi],  [a_hi | a_mid] . [b_mid | b_hi],  [a_mid | a_hi] . [b_mid | b_hi]
// at the kernel's occupancy (256-thread workgroups, 56 KB of LDS: two per CU), no global traffic.  The loop is power-bound
// on this part: the question is which shape leaves the higher clock.  Prints issued bf16 TFLOP/s and the shader clock /
// package power rocm-smi reports while each variant runs.
// hipcc -O3 --offload-arch=gfx950 tools/split_shapes_probe.hip -o tools/bin/split_shapes_probe && tools/bin/split_shapes_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROW = 7;                       // 16-byte chunks per LDS row: 3 planes x 2 chunks + 1 pad (the kernel's image)
constexpr int ROWS = 256;                    // 128 A rows + 128 W rows
constexpr int BUF = ROWS * ROW;

__device__ __forceinline__ void fill_lds(u32x4* lds) {
    // finite, mixed-sign bf16 pairs (exponents around 2^-3 .. 2^0): 0x3Exx / 0xBExx patterns
    for (int i = threadIdx.x; i < 2 * BUF; i += 256) {
        const uint32_t s = (uint32_t)i * 2654435761u;
        const uint32_t w0 = 0x3E003E00u ^ (s & 0x807F807Fu), w1 = 0x3D803F00u ^ ((s >> 3) & 0x807F807Fu);
        lds[i] = u32x4{w0, w1, w1 ^ 0x00100010u, w0 ^ 0x00080008u};
    }
    __syncthreads();
}

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    __shared__ u32x4 lds[2 * BUF];           // 57 344 bytes: two workgroups per CU, like the kernel
    fill_lds(lds);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 1, wn = w & 1;
    float sum = 0.f;
    if constexpr (SHAPE == 0) {
        const int r = lane & 31, h = lane >> 5;
        const int a_off = (wm * 64 + r) * ROW + h, w_off = (128 + wn * 64 + r) * ROW + h;
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const u32x4* cst = lds + (it & 1) * BUF;
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[u][p] = __builtin_bit_cast(bf16x8, cst[a_off + u * 32 * ROW + 2 * p]);
                    b[u][p] = __builtin_bit_cast(bf16x8, cst[w_off + u * 32 * ROW + 2 * p]);
                }
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[k]], b[j][PB[k]], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    } else {
        const int r = lane & 15, g = lane >> 4;
        // chunk of plane pair [X | Y] for this lane's k group: g < 2 -> plane X chunk g, else plane Y chunk g - 2
        auto chunk = [&](int X, int Y) { return g < 2 ? 2 * X + g : 2 * Y + g - 2; };
        const int ca[3] = {chunk(0, 2), chunk(0, 1), chunk(1, 0)};      // A: [hi|lo], [hi|mid], [mid|hi]
        const int cb[2] = {chunk(2, 0), chunk(1, 0)};                   // B: [lo|hi], [mid|hi]
        const int a_row = (wm * 64 + r) * ROW, w_row = (128 + wn * 64 + r) * ROW;
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            const u32x4* cst = lds + (it & 1) * BUF;
            bf16x8 a[4][3], b[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int p = 0; p < 3; ++p) a[u][p] = __builtin_bit_cast(bf16x8, cst[a_row + u * 16 * ROW + ca[p]]);
#pragma unroll
                for (int p = 0; p < 2; ++p) b[u][p] = __builtin_bit_cast(bf16x8, cst[w_row + u * 16 * ROW + cb[p]]);
            }
            constexpr int KA[3] = {0, 1, 2}, KB[3] = {0, 1, 1};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][KA[k]], b[j][KB[k]], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) sum += acc[i][j][e];
    }
    if (sum == 12345.678f) out[0] = sum;     // keep the chain alive
}

static void smi(const char* tag) {
    FILE* p = popen("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Package Power'", "r");
    if (!p) return;
    char line[256];
    while (fgets(line, sizeof line, p)) printf("    [%s] %s", tag, line);
    pclose(p);
}

template <int SHAPE>
static void run(const char* name, float* out) {
    const int grid = 512, warm = 2000;
    hipLaunchKernelGGL(probe<SHAPE>, dim3(grid), dim3(256), 0, 0, out, warm);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 400000;                // ~1.5 - 2.5 s
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<SHAPE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    smi(name);                               // sampled while the kernel runs (the launch is asynchronous)
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 64 * 64 * 16 * 6 * 4.0 * grid * (double)iters;       // issued bf16 MFMA FLOPs
    printf("%s: %.1f ms, %.0f TFLOP/s issued on the bf16 pipe = %.0f TF of fp32-accurate FLOPs (six products per term)\n", name, ms,
           flop / (ms * 1e-3) / 1e12, flop / 6 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, 64);
    smi("idle");
    run<0>("32x32x16, 6 products, 12 fragment reads", out);
    run<1>("16x16x32, 3 dual products, 20 fragment reads", out);
    run<0>("32x32x16 again", out);
    return 0;
}
