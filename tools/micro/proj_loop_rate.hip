// The inner loop of csrc/nm_proj.hip in isolation: 128 stationary A registers, a B fragment per step from LDS
// (ds_read_b128, D ahead), four 16x16x4 MFMAs per step -- which ingredient keeps it from the 155 TFLOP/s of a bare
// MFMA loop?      hipcc --offload-arch=gfx950 -O3 -o proj_loop_rate proj_loop_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// READS: fragment reads from LDS; BAR: a barrier per 16 steps; SGB: sched_group_barrier pinning; NACC accumulators
template <bool READS, bool BAR, bool SGB, int NACC>
__global__ __launch_bounds__(512, 2) void loop_kernel(float* out, const float* in, int blocks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 3 * 64 * 64; i += blockDim.x) lds[i] = in[i & 1023];
    float a[128];
#pragma unroll
    for (int j = 0; j < 128; ++j) a[j] = in[(tid + 7 * j) & 1023];
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int p = 0; p < NACC; ++p) acc[p] = {0.f, 0.f, 0.f, 0.f};
    const float* bl = lds + (4 * kq) * 64 + 4 * n;
    auto frag = [&](const float* bp, int s) { return *reinterpret_cast<const float4*>(bp + (16 * (s >> 2) + (s & 3)) * 64); };
    constexpr int S = 16, D = 4;
    float4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = frag(bl, d);
    int gi = 0;
    for (int blk = 0; blk < blocks; ++blk) {
#pragma unroll
        for (int c = 0; c < 8; ++c, ++gi) {
            if (BAR) __builtin_amdgcn_s_barrier();
            const float* bp = bl + (gi % 3) * 4096;
            const float* bn = bl + ((gi + 1) % 3) * 4096;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float4 b = ring[s % D];
                if (READS) ring[s % D] = (s + D < S) ? frag(bp, s + D) : frag(bn, s + D - S);
                const float av = a[4 * (4 * c + (s >> 2)) + (s & 3)];
                if (NACC == 4) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.w, acc[3], 0, 0, 0);
                } else {
                    acc[(2 * s) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.x, acc[(2 * s) % NACC], 0, 0, 0);
                    acc[(2 * s + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.y, acc[(2 * s + 1) % NACC], 0, 0, 0);
                    acc[(2 * s) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.z, acc[(2 * s) % NACC], 0, 0, 0);
                    acc[(2 * s + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.w, acc[(2 * s + 1) % NACC], 0, 0, 0);
                }
                if (SGB) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NACC; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    if (s == 12345.678f) out[tid] = s;
}

template <typename K>
static void run(const char* name, K kern, int threads, float* out, const float* in) {
    const int blocks = 10;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 3 * 64 * 64 * 4, 0, out, in, blocks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 3 * 64 * 64 * 4, 0, out, in, blocks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 20;
    const double flop = 256.0 * (threads / 64) * blocks * 8 * 16 * 4 * 2048.0;
    printf("%-44s %4d threads: %8.1f us  %6.1f TFLOP/s  (%s)\n", name, threads, us, flop / us * 1e-6,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    float *out, *in;
    hipMalloc(&out, 4096); hipMalloc(&in, 4096);
    hipMemset(in, 0, 4096);
    for (int threads : {256, 512}) {
        run("bare (no reads, no barrier)", loop_kernel<false, false, false, 4>, threads, out, in);
        run("reads", loop_kernel<true, false, false, 4>, threads, out, in);
        run("reads + sched_group_barrier", loop_kernel<true, false, true, 4>, threads, out, in);
        run("reads + sgb + barrier", loop_kernel<true, true, true, 4>, threads, out, in);
        run("no reads + sgb", loop_kernel<false, false, true, 4>, threads, out, in);
        run("reads + sgb, 2 accumulators (NACC 2)", loop_kernel<true, false, true, 2>, threads, out, in);
    }
    return 0;
}
