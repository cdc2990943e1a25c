// Micro-benchmark: does a wave64 vector instruction cost less when all of its active lanes sit in one
// 32-lane half of the wave?  Each variant runs the same chain of independent v_fma_f32 under a
// different set of active lanes; waves per SIMD is varied through the grid size.
//   hipcc --offload-arch=gfx950 -O3 -o halfwave halfwave.hip && ./halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void chain(float* out, unsigned long long mask, int iters)
{
    const unsigned lane = threadIdx.x;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0000001f, c = 0.5f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    const int iters = 4096;
    const struct { const char* name; unsigned long long mask; } variants[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 32-63", 0xffffffff00000000ull},
        {"even lanes (32 scattered)", 0x5555555555555555ull}, {"lanes 0-15", 0xffffull}, {"22 scattered lanes", 0x9249249249249249ull},
        {"lanes 0-21", 0x3fffffull}, {"lane 0 and lane 63", 0x8000000000000001ull}};
    float* out;
    hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wavesPerSimd : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * 4 * wavesPerSimd;
        printf("waves per SIMD %d (%d blocks of one wave)\n", wavesPerSimd, blocks);
        for (const auto& v : variants) {
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, out, v.mask, 16);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, out, v.mask, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)iters * 16 * 8 * wavesPerSimd;  // vector instructions per SIMD
            printf("  %-28s %8.3f ms   %.2f ns per instruction per SIMD\n", v.name, ms, ms * 1e6 / insts);
        }
    }
    return 0;
}
