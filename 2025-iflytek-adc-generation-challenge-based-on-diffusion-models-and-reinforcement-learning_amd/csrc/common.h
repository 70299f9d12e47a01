// Shared host/device helpers of the adcdiff HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/adcdiff.h"

namespace adcdiff {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Thread-local message for adcdiff_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

// Measurement switches behind adcdiff_debug_set (include/adcdiff.h): every default is the shipped behaviour, and with no
// adcdiff_debug_set call the kernel choice is a function of the arguments alone.  No environment variable is read.
enum DebugKey { DBG_TILE_HINTS, DBG_GEMM_TILE, DBG_GEMM_RING, DBG_GEMM_RING2, DBG_GEMM_RING_MAX, DBG_GEMM_W4, DBG_GEMM_W4_MIN_N, DBG_GEMM_BIG_MIN_K, DBG_GEMM_W1, DBG_GEMM_W1_ROWS,
                DBG_GEMM_CL, DBG_GEMM_SPLITK, DBG_GEMM_PF_W, DBG_GEMM_PF_NEXT, DBG_STEP_FOLD, DBG_TRAIN_TRANSPOSE, DBG_TRAIN_GROUP_DW, DBG_TRAIN_FUSE_CAST, DBG_TRAIN_FUSED_FWD, DBG_SYNC, DBG_NKEYS };
long long debug_value(int key);

// Sum over aligned groups of LPG adjacent lanes, every lane receiving its group's sum: the xor butterfly (partner lane ^ 1, ^ 2,
// ^ 4, ...) on the cross-lane data path instead of ds_bpermute.  From the third step on every lane of a half already holds the same
// partial sum, so the mirror patterns (lane ^ 7 within 8, lane ^ 15 within 16) deliver the same addend as lane ^ 4 / lane ^ 8:
// the result equals the __shfl_xor loop's bit for bit.  LPG = 0: the loop, for a count known only at run time.
#ifdef __HIPCC__
template <int CTRL>
__device__ __forceinline__ float lane_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int LPG>
__device__ __forceinline__ float lanes_allsum(float x, int lanes_runtime) {
    if constexpr (LPG == 0) {
        for (int m = 1; m < lanes_runtime; m <<= 1) x += __shfl_xor(x, m);
    } else {
        if constexpr (LPG >= 2) x += lane_dpp<0xB1>(x);    // quad_perm [1,0,3,2]
        if constexpr (LPG >= 4) x += lane_dpp<0x4E>(x);    // quad_perm [2,3,0,1]
        if constexpr (LPG >= 8) x += lane_dpp<0x141>(x);   // row_half_mirror
        if constexpr (LPG >= 16) x += lane_dpp<0x140>(x);  // row_mirror
        if constexpr (LPG >= 32) x += __shfl_xor(x, 16);
        if constexpr (LPG >= 64) x += __shfl_xor(x, 32);
    }
    return x;
}
#endif

inline hipStream_t as_stream(adcdiff_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ADCDIFF_OK;
}

// Brackets one launch with hipEvents while adcdiff_timing_begin/_end is active (see include/adcdiff.h).
struct TimingScope {
    int slot;
    hipStream_t stream;
    TimingScope(int kind, int M, int N, int K, double flops, double bytes, hipStream_t st);
    ~TimingScope();
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

// ---------------------------------------------------------------- LDS-DMA staging helpers
// LDS-DMA copy of 16 bytes per lane: the wave's 64 lanes fill 1 KiB starting at the wave-uniform `lds`.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// The same copy issued from inline asm: hipcc does not track it, so it inserts no vmcnt(0) in front of the next ds_read
// (which it does for the builtin whenever the destination may alias) and a prefetch really stays in flight under the
// MFMAs; the loops that use it count vmcnt by hand.  M0 carries the wave-uniform LDS byte address and is restored.
__device__ __forceinline__ void glds16_untracked(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `tiles` (wave-uniform, 0..7) groups of P vector-memory operations are outstanding
template <int P>
__device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
    switch (tiles) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<P>(); break;
        case 2: wait_vmcnt<2 * P>(); break;
        case 3: wait_vmcnt<3 * P>(); break;
        case 4: wait_vmcnt<(4 * P <= 63 ? 4 * P : 63)>(); break;
        case 5: wait_vmcnt<(5 * P <= 63 ? 5 * P : 63)>(); break;
        case 6: wait_vmcnt<(6 * P <= 63 ? 6 * P : 63)>(); break;
        default: wait_vmcnt<(7 * P <= 63 ? 7 * P : 63)>(); break;
    }
}

// ---------------------------------------------------------------- Philox4x32-10 noise
// counter = (col / 4, row, step, stream), key = (seed lo, seed hi); the four output words give the
// four normals of columns 4q..4q+3 by two Box-Muller pairs (see oracle/philox.py for the definition).
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                              uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void box_muller(uint32_t wa, uint32_t wb, float& z0, float& z1) {
    const float u1 = (float)((wa >> 8) + 1u) * 5.9604644775390625e-8f;   // (0, 1], exact
    const float u2 = (float)(wb >> 8) * 5.9604644775390625e-8f;          // [0, 1), exact
    // v_log_f32 is log2; v_sin/v_cos take their argument in revolutions, so 2*pi*u2 is never formed
    const float radius = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    z0 = radius * __builtin_amdgcn_cosf(u2);
    z1 = radius * __builtin_amdgcn_sinf(u2);
}

__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint32_t stream, uint32_t step, uint64_t row,
                                                uint32_t quad) {
    uint32_t c0 = quad, c1 = (uint32_t)row, c2 = step, c3 = stream;
    philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
    f32x4 z;
    float a, b;
    box_muller(c0, c1, a, b); z[0] = a; z[1] = b;
    box_muller(c2, c3, a, b); z[2] = a; z[3] = b;
    return z;
}

// Dropout keep factors (0 or 1 / (1 - p)) for columns 4q..4q+3 of a row: Philox stream 5, step field = layer index.
__device__ __forceinline__ f32x4 dropout_keep4(uint64_t seed, uint32_t layer, uint64_t row, uint32_t quad, float p) {
    uint32_t c0 = quad, c1 = (uint32_t)row, c2 = layer, c3 = 5u;
    philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float scale = 1.0f / (1.0f - p);
    const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
    f32x4 k;
    k[0] = c0 >= thr ? scale : 0.0f; k[1] = c1 >= thr ? scale : 0.0f; k[2] = c2 >= thr ? scale : 0.0f; k[3] = c3 >= thr ? scale : 0.0f;
    return k;
}

}  // namespace adcdiff
