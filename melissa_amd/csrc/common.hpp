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

// ---- node sets of up to 64 * W nodes -------------------------------------------------------------------------------
// Graphs of up to 64 nodes (the sizes BASELINE quotes: 20, 50) hold one node per wavefront lane and a node set in ONE
// 64-bit mask (W = 1).  The reference CLI's third size, --n-agents 100 (common.py:49), takes W = 2: a lane holds the
// two nodes lane and lane + 64, a node set is two words (word k = nodes 64 k .. 64 k + 63), a ballot yields one word.
// Every kernel that touches node sets is a template on W; with W = 1 the wrappers below compile to the plain uint64_t
// arithmetic the kernels were written with.  In memory a set is W consecutive uint64 (MEL_SET_WORDS(n_nodes)).
template <int W>
struct NodeSet {
    uint64_t w[W];
};
#define MEL_W_FOR(k) _Pragma("unroll") for (int k = 0; k < W; ++k)

template <int W> __device__ __forceinline__ NodeSet<W> ns_zero() {
    NodeSet<W> r;
    MEL_W_FOR(k) r.w[k] = 0ull;
    return r;
}
template <int W> __device__ __forceinline__ NodeSet<W> ns_bit(int i) {
    NodeSet<W> r;
    if constexpr (W == 1) r.w[0] = 1ull << i;
    else MEL_W_FOR(k) r.w[k] = ((i >> 6) == k) ? (1ull << (i & 63)) : 0ull;
    return r;
}
// nodes 0 .. n-1
template <int W> __device__ __forceinline__ NodeSet<W> ns_full(int n) {
    NodeSet<W> r;
    MEL_W_FOR(k) {
        const int c = n - 64 * k;
        r.w[k] = c >= 64 ? ~0ull : (c <= 0 ? 0ull : ((1ull << c) - 1ull));
    }
    return r;
}
template <int W> __device__ __forceinline__ NodeSet<W> operator|(NodeSet<W> a, const NodeSet<W>& b) {
    MEL_W_FOR(k) a.w[k] |= b.w[k];
    return a;
}
template <int W> __device__ __forceinline__ NodeSet<W> operator&(NodeSet<W> a, const NodeSet<W>& b) {
    MEL_W_FOR(k) a.w[k] &= b.w[k];
    return a;
}
template <int W> __device__ __forceinline__ NodeSet<W> operator~(NodeSet<W> a) {
    MEL_W_FOR(k) a.w[k] = ~a.w[k];
    return a;
}
template <int W> __device__ __forceinline__ NodeSet<W>& operator|=(NodeSet<W>& a, const NodeSet<W>& b) {
    MEL_W_FOR(k) a.w[k] |= b.w[k];
    return a;
}
template <int W> __device__ __forceinline__ NodeSet<W>& operator&=(NodeSet<W>& a, const NodeSet<W>& b) {
    MEL_W_FOR(k) a.w[k] &= b.w[k];
    return a;
}
template <int W> __device__ __forceinline__ bool operator==(const NodeSet<W>& a, const NodeSet<W>& b) {
    bool e = true;
    MEL_W_FOR(k) e = e && a.w[k] == b.w[k];
    return e;
}
template <int W> __device__ __forceinline__ bool operator!=(const NodeSet<W>& a, const NodeSet<W>& b) { return !(a == b); }
template <int W> __device__ __forceinline__ bool ns_any(const NodeSet<W>& a) {
    uint64_t o = 0;
    MEL_W_FOR(k) o |= a.w[k];
    return o != 0ull;
}
template <int W> __device__ __forceinline__ int ns_count(const NodeSet<W>& a) {
    int c = 0;
    MEL_W_FOR(k) c += __popcll(a.w[k]);
    return c;
}
// lowest / highest member (the set must not be empty)
template <int W> __device__ __forceinline__ int ns_lowest(const NodeSet<W>& a) {
    if constexpr (W == 1) return lowest_bit(a.w[0]);
    else return a.w[0] ? lowest_bit(a.w[0]) : 64 + lowest_bit(a.w[1]);
}
template <int W> __device__ __forceinline__ int ns_highest(const NodeSet<W>& a) {
    if constexpr (W == 1) return 63 - __clzll((long long)a.w[0]);
    else return a.w[1] ? 127 - __clzll((long long)a.w[1]) : 63 - __clzll((long long)a.w[0]);
}
template <int W> __device__ __forceinline__ void ns_clear_lowest(NodeSet<W>& a) {      // an empty set stays empty
    if constexpr (W == 1) a.w[0] &= a.w[0] - 1ull;
    else {
        if (a.w[0]) a.w[0] &= a.w[0] - 1ull;
        else a.w[1] &= a.w[1] - 1ull;
    }
}
template <int W> __device__ __forceinline__ bool ns_test(const NodeSet<W>& a, int i) {
    if constexpr (W == 1) return (a.w[0] >> i) & 1ull;
    else return ((i < 64 ? a.w[0] : a.w[1]) >> (i & 63)) & 1ull;
}
// is node lane + 64 h a member?
template <int W> __device__ __forceinline__ bool ns_mine(const NodeSet<W>& a, int lane, int h) { return (a.w[h] >> lane) & 1ull; }
// number of members below node i: the position of node i in a set-ordered row list
template <int W> __device__ __forceinline__ int ns_rank_below(const NodeSet<W>& a, int i) {
    if constexpr (W == 1) return rank_below(a.w[0], i);
    else return i < 64 ? rank_below(a.w[0], i) : __popcll(a.w[0]) + rank_below(a.w[1], i - 64);
}
template <int W> __device__ __forceinline__ NodeSet<W> ns_uniform(NodeSet<W> a) {      // wave-uniform value -> SGPRs
    MEL_W_FOR(k) a.w[k] = uniform_u64(a.w[k]);
    return a;
}
template <int W> __device__ __forceinline__ NodeSet<W> ns_wave_or(NodeSet<W> a) {
    MEL_W_FOR(k) a.w[k] = wave_or_u64(a.w[k]);
    return a;
}
// element `idx` of an array of sets
template <int W> __device__ __forceinline__ NodeSet<W> ns_load(const uint64_t* p, size_t idx) {
    NodeSet<W> r;
    MEL_W_FOR(k) r.w[k] = p[idx * W + k];
    return r;
}
template <int W> __device__ __forceinline__ void ns_store(uint64_t* p, size_t idx, const NodeSet<W>& a) {
    MEL_W_FOR(k) p[idx * W + k] = a.w[k];
}
// cross-lane reads of per-node values (v[h] = value of node lane + 64 h) at a WAVE-UNIFORM node i
template <int W> __device__ __forceinline__ int node_i32(const int (&v)[W], int i) {
    if constexpr (W == 1) return lane_i32(v[0], i);
    else return i < 64 ? lane_i32(v[0], i) : lane_i32(v[1], i - 64);
}
template <int W> __device__ __forceinline__ double node_f64(const double (&v)[W], int i) {
    if constexpr (W == 1) return lane_f64(v[0], i);
    else return i < 64 ? lane_f64(v[0], i) : lane_f64(v[1], i - 64);
}
template <int W> __device__ __forceinline__ NodeSet<W> node_set(const NodeSet<W> (&v)[W], int i) {
    NodeSet<W> r;
    if constexpr (W == 1) r.w[0] = lane_u64(v[0].w[0], i);
    else MEL_W_FOR(k) r.w[k] = i < 64 ? lane_u64(v[0].w[k], i) : lane_u64(v[1].w[k], i - 64);
    return r;
}
inline int set_words(int n_nodes) { return (n_nodes + 63) >> 6; }

}  // namespace mel
