// Sustained matrix-core rate of v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 from registers (no memory traffic), one
// wave per SIMD, 256 accumulator registers per lane as in the four-wave GEMM: which shape does more work inside the power limit?
// hipcc --offload-arch=gfx950 -O3 -o mfma_power_probe mfma_power_probe.hip && ./mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void spin(float* out, int iters, long long* clk) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 8; ++k) {
            if (iters & 1) {   // odd iteration count: operands with random signs, exponents and mantissas (|x| in [0.25, 4))
                unsigned h = (threadIdx.x * 64 + i * 8 + k + blockIdx.x * 16384) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                unsigned short ua = (unsigned short)((h & 0x807f) | ((125 + ((h >> 8) & 3)) << 7));
                unsigned short ub = (unsigned short)(((h >> 16) & 0x807f) | ((125 + ((h >> 28) & 3)) << 7));
                a[i][k] = __builtin_bit_cast(__bf16, ua); b[i][k] = __builtin_bit_cast(__bf16, ub);
            } else {
                a[i][k] = (__bf16)(0.001f * (threadIdx.x + i + k)); b[i][k] = (__bf16)(0.002f * (threadIdx.x ^ (i * 8 + k)));
            }
        }
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[8][8] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * i + h], b[2 * j + h], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
void run(const char* name, float* out, long long* clk, int iters, int launches) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(spin<SHAPE>, dim3(256), dim3(256), 0, 0, out, iters, clk);   // settle the clocks
    hipEventRecord(e0, 0);
    for (int w = 0; w < launches; ++w) hipLaunchKernelGGL(spin<SHAPE>, dim3(256), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[512]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double mhz = 0; for (int i = 0; i < 256; ++i) mhz += 100.0 * h[2 * i] / h[2 * i + 1];
    // per iteration and wave: 64 MFMAs of 16x16x32 (2 * 16 * 16 * 32 flop) = 32 of 32x32x16 (2 * 32 * 32 * 16 flop) = 1048576 flop
    const double flop = 1048576.0 * iters * 4 * 256 * launches;
    printf("%-30s %8.1f ms  %7.1f TFLOP/s  shader clock %.0f MHz  matrix-core utilisation at that clock %.2f\n", name, ms, flop / ms * 1e-9, mhz / 256,
           flop / ms * 1e-9 * 1e12 / (256.0 * 4 * 1024 * mhz / 256 * 1e6));
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 512 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<16>("16x16x32 bf16, smooth data", out, clk, 20000, 120);
        run<32>("32x32x16 bf16, smooth data", out, clk, 20000, 120);
        run<16>("16x16x32 bf16, random data", out, clk, 20001, 120);
        run<32>("32x32x16 bf16, random data", out, clk, 20001, 120);
    }
    return hipDeviceSynchronize() != hipSuccess;
}
