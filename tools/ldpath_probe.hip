// What does the vector-memory path of a CU deliver for the operand stream of the 128 x 128 split GEMM?  256 loader threads per
// CU (four waves, nothing else on the CU) stream conv2's operands - A: fp32 rows 2 KB apart, W: bf16 planes - in the access
// shapes below, data only xor-ed together.  Cycles per 16-k step of one tile (20 KB useful: 8 KB of A, 12 KB of W planes).
//   P0  the kernel's shape today: per step and thread 2 x 16 B of A (4 lanes = 64 B of a row: half lines) + 3 x 16 B of W
//       (6 lanes = 96 B of a 3 KB plane row, misaligned against the 128-byte lines), one step after the other
//   P1  the same addresses, the loads of FOUR consecutive steps issued back to back (every line's pieces arrive together)
//   P2  full lines: A 128 B per row and instruction (8 rows x 128 B per wave-instruction, two steps' worth), W from a BLOCKED
//       image ([column tile][step][128 rows x 96 B] contiguous: 1 KB per wave-instruction)
//   P3  P2's W with P0's A          P4  P2's A with P0's W
// hipcc -O3 --offload-arch=gfx950 tools/ldpath_probe.hip -o tools/bin/ldpath_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 512, KT = K / 16, ROWS = 10752, NCOL = 512;     // 84 row panels x 4 column tiles = 336 tiles

template <int P>
__global__ __launch_bounds__(256) void ldprobe(const float* __restrict__ A, const u32x4* __restrict__ W, const u32x4* __restrict__ Wb,
                                               uint32_t* out, unsigned long long* cyc) {
    const int tid = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    auto eat = [&](u32x4 v) { acc ^= v; };
    auto eatf = [&](f32x4 v) { acc ^= __builtin_bit_cast(u32x4, v); };
    const unsigned long long t0 = __builtin_readcyclecounter();
    int steps = 0;
    for (int tile = blockIdx.x; tile < (ROWS / 128) * (NCOL / 128); tile += gridDim.x) {
        // XCD-contiguous order as in the kernel: the four column tiles of a row panel are neighbours on one XCD
        const int total = (ROWS / 128) * (NCOL / 128), q = total >> 3, xcd = tile & 7, local = tile >> 3;
        const int wg = xcd * q + local;
        const int m0 = (wg / 4) * 128, n0 = (wg % 4) * 128;
        if constexpr (P == 0 || P == 1) {
            const int crow = tid >> 2, kq = tid & 3;
            const float* a0 = A + (size_t)(m0 + crow) * K + kq * 4;
            const float* a1 = A + (size_t)(m0 + crow + 64) * K + kq * 4;
            const u32x4* w[3];
            for (int i = 0; i < 3; ++i) {
                const int ch = tid + i * 256, wrow = ch / 6, wch = ch - wrow * 6;
                w[i] = W + (size_t)(n0 + wrow) * (3 * K / 8) + wch;
            }
            if constexpr (P == 0) {
#pragma unroll 4
                for (int s = 0; s < KT; ++s) {
                    eatf(*reinterpret_cast<const f32x4*>(a0 + s * 16)), eatf(*reinterpret_cast<const f32x4*>(a1 + s * 16));
                    for (int i = 0; i < 3; ++i) eat(w[i][s * 6]);
                }
            } else {
                for (int s4 = 0; s4 < KT; s4 += 4) {
                    f32x4 ra[4][2];
                    u32x4 rw[4][3];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        ra[s][0] = *reinterpret_cast<const f32x4*>(a0 + (s4 + s) * 16), ra[s][1] = *reinterpret_cast<const f32x4*>(a1 + (s4 + s) * 16);
#pragma unroll
                        for (int i = 0; i < 3; ++i) rw[s][i] = w[i][(s4 + s) * 6];
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        eatf(ra[s][0]), eatf(ra[s][1]);
#pragma unroll
                        for (int i = 0; i < 3; ++i) eat(rw[s][i]);
                    }
                }
            }
        } else {
            const bool fullA = (P == 2 || P == 4), blockW = (P == 2 || P == 3);
            const int crow = tid >> 2, kq = tid & 3;
            const float* a0 = A + (size_t)(m0 + crow) * K + kq * 4;
            const float* a1 = A + (size_t)(m0 + crow + 64) * K + kq * 4;
            const int frow = tid >> 3, fc = tid & 7;            // full lines: 8 lanes per 128-byte row piece, 32 rows per pass, 4 passes
            const float* fa = A + (size_t)(m0 + frow) * K + fc * 4;
            const u32x4* w[3];
            for (int i = 0; i < 3; ++i) {
                const int ch = tid + i * 256, wrow = ch / 6, wch = ch - wrow * 6;
                w[i] = W + (size_t)(n0 + wrow) * (3 * K / 8) + wch;
            }
            const u32x4* wb = Wb + (size_t)(n0 / 128) * KT * 768 + tid;
            for (int s2 = 0; s2 < KT; s2 += 2) {
                if (fullA) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) eatf(*reinterpret_cast<const f32x4*>(fa + (size_t)p * 32 * K + s2 * 16));
                } else {
                    for (int s = s2; s < s2 + 2; ++s)
                        eatf(*reinterpret_cast<const f32x4*>(a0 + s * 16)), eatf(*reinterpret_cast<const f32x4*>(a1 + s * 16));
                }
                for (int s = s2; s < s2 + 2; ++s) {
                    if (blockW) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) eat(wb[(size_t)s * 768 + i * 256]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; ++i) eat(w[i][s * 6]);
                    }
                }
            }
        }
        steps += KT;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc[0] == 0x12345678u && acc[1] == 77u) out[0] = acc[2] ^ acc[3];
    if (tid == 0) atomicAdd(&cyc[0], t1 - t0), atomicAdd(&cyc[1], (unsigned long long)steps), atomicAdd(&cyc[2], 1ull);
}

template <int P>
static void run(const char* name, const float* A, const u32x4* W, const u32x4* Wb, uint32_t* out, unsigned long long* cyc, int grid) {
    hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL(ldprobe<P>, dim3(grid), dim3(256), 0, 0, A, W, Wb, out, cyc);
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ldprobe<P>, dim3(grid), dim3(256), 0, 0, A, W, Wb, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3];
    hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
    const double us = ms / 20 * 1e3, bytes = 336.0 * KT * 20480;
    printf("%-64s grid %3d: %6.1f us per pass, %5.2f TB/s, %6.0f cycles per step (20 KB: %4.1f B/clk per CU)\n", name, grid, us, bytes / us / 1e6,
           (double)h[0] / (double)h[1], 20480.0 / ((double)h[0] / (double)h[1]));
}

int main() {
    float* A;
    u32x4 *W, *Wb;
    uint32_t* out;
    unsigned long long* cyc;
    hipMalloc(&A, (size_t)ROWS * K * 4), hipMemset(A, 0x3c, (size_t)ROWS * K * 4);
    hipMalloc(&W, (size_t)NCOL * 3 * K * 2), hipMemset(W, 0x3d, (size_t)NCOL * 3 * K * 2);
    hipMalloc(&Wb, (size_t)NCOL * 3 * K * 2), hipMemset(Wb, 0x3e, (size_t)NCOL * 3 * K * 2);
    hipMalloc(&out, 64), hipMalloc(&cyc, 64);
    for (int grid : {256, 512}) {
        run<0>("P0 today: half lines of A, 96-byte pieces of W, step by step", A, W, Wb, out, cyc, grid);
        run<1>("P1 same addresses, four steps' loads back to back", A, W, Wb, out, cyc, grid);
        run<2>("P2 full lines of A + blocked W (1 KB per wave-instruction)", A, W, Wb, out, cyc, grid);
        run<3>("P3 blocked W, A as today", A, W, Wb, out, cyc, grid);
        run<4>("P4 full lines of A, W as today", A, W, Wb, out, cyc, grid);
    }
    return 0;
}
