// Shared host/device helpers for libmelissa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/melissa_hip.h"

namespace mel {

void set_error(const char* fmt, ...);

inline mel_status fail(mel_status code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline mel_status fail(mel_status code, const char* fmt, ...) {
    char buf[480];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

// hipGetLastError() is sticky per thread: an error left behind by somebody else's HIP call (e.g. a
// device-count probe) must not be blamed on our launch, so every entry point clears it first.
inline void clear_stale_error() { (void)hipGetLastError(); }

inline mel_status check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MEL_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MEL_OK;
}

// ---- optional stage timer (see mel_prof_* in melissa_hip.h) ----
// While a profiler is attached and a StageScope is open, every kernel launch made through MEL_LAUNCH carries its own
// pair of HIP events ON THE DISPATCH ITSELF (hipExtLaunchKernelGGL's start / stop events: the begin / end timestamps of
// that kernel, what a kernel trace reports) and is booked under the open stage; a stage's time is the sum of its
// launches.  Bracketing a launch with hipEventRecord calls instead also measures the two event packets and the idle gaps
// around them (4-7 us per stage on this stack), which is not kernel time.
struct Profiler {
    int capacity = 0;
    int count = 0;
    hipEvent_t* ev = nullptr;     // 2 per record: begin, end
    int* stage = nullptr;
    int open_stage = -1;
};
Profiler* current_profiler();

struct StageScope {
    Profiler* p;
    int prev = -1;
    StageScope(int stage, hipStream_t) : p(current_profiler()) {
        if (p) prev = p->open_stage, p->open_stage = stage;
    }
    ~StageScope() {
        if (p) p->open_stage = prev;
    }
};

// event pair for the next launch, or false when nothing is being timed
inline bool launch_events(hipEvent_t* begin, hipEvent_t* end) {
    Profiler* p = current_profiler();
    if (!p || p->open_stage < 0 || p->count >= p->capacity) return false;
    const int idx = p->count++;
    p->stage[idx] = p->open_stage;
    *begin = p->ev[2 * idx], *end = p->ev[2 * idx + 1];
    return true;
}

#define MEL_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                  \
    do {                                                                                                     \
        hipEvent_t mel_ev0_, mel_ev1_;                                                                       \
        if (mel::launch_events(&mel_ev0_, &mel_ev1_))                                                        \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, mel_ev0_, mel_ev1_, 0, __VA_ARGS__);   \
        else                                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                             \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller-provided workspace (256-byte granules).
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + off);
        off += align_up(count * sizeof(T), 256);
        return p;
    }
};

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// number of set bits of m strictly below bit i: the position of node i in a mask-ordered row list
__device__ __forceinline__ int rank_below(uint64_t m, int i) {
    return __popcll(m & ((1ull << i) - 1ull));
}

__device__ __forceinline__ int lowest_bit(uint64_t m) { return __ffsll((long long)m) - 1; }

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src, 64);
    uint32_t hi = __shfl((uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// ---- cross-lane reads with a WAVE-UNIFORM source lane: v_readlane (VALU -> SGPR), no LDS round trip ----
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int lane_i32(int v, int src_uniform) {
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src_uniform));
}
__device__ __forceinline__ float lane_f32(float v, int src_uniform) {
    return __builtin_bit_cast(float, lane_i32(__builtin_bit_cast(int, v), src_uniform));
}
__device__ __forceinline__ uint64_t lane_u64(uint64_t v, int src_uniform) {
    const int src = __builtin_amdgcn_readfirstlane(src_uniform);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double lane_f64(double v, int src_uniform) {
    return __longlong_as_double((long long)lane_u64((uint64_t)__double_as_longlong(v), src_uniform));
}
// sum of an int over the wave: DPP butterfly inside each 16-lane row, then four readlanes
__device__ __forceinline__ int wave_sum_i32_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// sum of a float over each 16-lane row (every lane of the row gets it): the same DPP butterfly
__device__ __forceinline__ float row16_sum_f32(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
    return x;
}

// OR of a 64-bit value over the wave: the DPP butterfly of wave_sum_i32_dpp on each half, then four readlanes
__device__ __forceinline__ uint32_t wave_or_u32_dpp(uint32_t u) {
    int v = (int)u;
    v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    return (uint32_t)(__builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) |
                      __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v) {
    return ((uint64_t)wave_or_u32_dpp((uint32_t)(v >> 32)) << 32) | wave_or_u32_dpp((uint32_t)v);
}

}  // namespace mel
