// fp32 matrix-core issue rate on gfx950: v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32, 1 or 2 waves per SIMD,
// short and long launches (sustained clocks).  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ long g_clk[4];

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float a, float b) {
    const long w0 = (long)wall_clock64(), c0 = (long)clock64();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = (long)wall_clock64() - w0; g_clk[1] = (long)clock64() - c0; }
}

template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <typename F>
static float timed(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d\n", ncu);
    const int iters_list[3] = {256, 4096, 65536};
    for (int threads : {256, 512, 1024}) {
        for (int ii = 0; ii < 3; ++ii) {
            const int iters = iters_list[ii];
            const double waves = (double)ncu * threads / 64;
            {
                float us = timed([&] { hipLaunchKernelGGL(k16<4>, dim3(ncu), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f); }, ii == 2 ? 2 : 10);
                double flop = waves * iters * 16.0 * 2048.0;
                long clk[4];
                hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
                printf("16x16x4  4 acc  %4d threads/CU  iters %6d : %9.1f us  %7.1f TFLOP/s   shader clock %.0f MHz\n", threads, iters, us,
                       flop / us * 1e-6, clk[1] / (clk[0] / 100.0));
            }
            {
                float us = timed([&] { hipLaunchKernelGGL(k16<8>, dim3(ncu), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f); }, ii == 2 ? 2 : 10);
                double flop = waves * iters * 32.0 * 2048.0;
                printf("16x16x4  8 acc  %4d threads/CU  iters %6d : %9.1f us  %7.1f TFLOP/s\n", threads, iters, us, flop / us * 1e-6);
            }
            {
                float us = timed([&] { hipLaunchKernelGGL(k32<2>, dim3(ncu), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f); }, ii == 2 ? 2 : 10);
                double flop = waves * iters * 8.0 * 4096.0;
                printf("32x32x2  2 acc  %4d threads/CU  iters %6d : %9.1f us  %7.1f TFLOP/s\n", threads, iters, us, flop / us * 1e-6);
            }
        }
    }
    return 0;
}
