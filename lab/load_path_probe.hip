// How fast does one CU get a 64 KB operand tile out of L2, by LDS-DMA (buffer_load_dwordx4 ... lds, the GEMM kernels' path) against
// loads into registers followed by ds_write_b128?  256 workgroups of four waves (one per CU); every wave fetches sixteen 1 KB pieces
// (8 rows x 128 B, rows 4 KB apart, as the A operand at K = 2048) per tile, 64 tiles walking along K, from a 16 MB matrix that the
// eight workgroups sharing its rows keep in L2.  hipcc --offload-arch=gfx950 -O3 -o load_path_probe load_path_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void fetch(const char* A, int ld_bytes, int tiles, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // tile row = blockIdx / 8: eight workgroups share 256 rows (as the column tiles of a GEMM tile row share the A rows)
    const unsigned long long base = (unsigned long long)A + (unsigned long long)((((blockIdx.x & 7) * 4 + (blockIdx.x >> 6)) * 256) + wave * 64) * ld_bytes;   // the eight sharers of a tile row on one XCD
    const i32x4 srd = {__builtin_amdgcn_readfirstlane((int)(unsigned)base), __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu)),
                       __builtin_amdgcn_readfirstlane(64 * ld_bytes), 0x00020000};
    const int voff = (lane >> 3) * ld_bytes + (lane & 7) * 16;   // row lane / 8 of a piece, 16 bytes of its 128
    f32x4 acc = {0, 0, 0, 0};
    // DEPTH tiles in flight: tile t + DEPTH - 1 is issued before tile t is waited for
    constexpr int DEPTH = 4;
    f32x4 v[DEPTH][8];
    auto issue = [&](int t, int slot) {
        const int so = (t % tiles) * 128;
        if constexpr (MODE == 0) {
            const int dst = __builtin_amdgcn_readfirstlane((int)(lds0 + wave * 8192 * DEPTH + slot * 8192));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(voff + j * 8 * ld_bytes), "s"(srd), "s"(so) : "memory");
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v[slot][j]) : "v"(voff + j * 8 * ld_bytes), "s"(srd), "s"(so) : "memory");
        }
    };
    const int total = reps * tiles;
#pragma unroll
    for (int k = 0; k < DEPTH - 1; ++k) issue(k, k);
    for (int t0 = 0; t0 < total; t0 += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            issue(t0 + k + DEPTH - 1, (k + DEPTH - 1) % DEPTH);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (DEPTH - 1)) : "memory");   // tile t0 + k has arrived
            if constexpr (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    asm volatile("" : "+v"(v[k][j]));
                    *reinterpret_cast<f32x4*>(smem + wave * 8192 * DEPTH + k * 8192 + j * 1024 + lane * 16) = v[k][j];
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += *reinterpret_cast<const f32x4*>(smem + wave * 8192 * DEPTH + lane * 16);
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1];
}

template <int MODE>
void run(const char* name, const char* A, int ld, int tiles, float* out) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fetch<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fetch<MODE>, dim3(256), dim3(256), 131072, 0, A, ld, tiles, out, 2);
    (void)hipEventRecord(e0, 0);
    const int reps = 20;
    hipLaunchKernelGGL(fetch<MODE>, dim3(256), dim3(256), 131072, 0, A, ld, tiles, out, reps);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per CU and tile: 4 waves x 8 pieces x 1 KB = 32 KB (half a GEMM tile: the A operand only)
    const double per_tile_us = ms * 1e3 / (reps * tiles);
    printf("%-44s %.3f us per 32 KB per CU = %.1f GB/s per CU, %.2f TB/s per XCD\n", name, per_tile_us, 32768.0 / per_tile_us * 1e-3, 32 * 32768.0 / per_tile_us * 1e-6);
}

int main() {
    const int ld = 4096, rows = 256 * 32, tiles = 32;   // 32 tile rows of 256 rows, K = 2048 bf16
    char* A; float* out;
    (void)hipMalloc(&A, (size_t)rows * ld); (void)hipMemset(A, 1, (size_t)rows * ld);
    (void)hipMalloc(&out, 256 * 256 * 4);
    for (int k = 0; k < 2; ++k) {
        run<0>("LDS-DMA (buffer_load_dwordx4 ... lds)", A, ld, tiles, out);
        run<1>("buffer_load_dwordx4 + ds_write_b128", A, ld, tiles, out);
    }
    return hipDeviceSynchronize() != hipSuccess;
}
