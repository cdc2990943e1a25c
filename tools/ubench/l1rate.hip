// Micro-benchmark: what does the vector L1 charge for per-lane loads that hit it, by width, active lanes and
// address pattern?  Every wave issues iters x 8 independent loads into a 16 KB region (L1 resident, shared by
// the block's waves) at per-lane offsets that advance by a fixed stride (two vector instructions per load).
//   hipcc --offload-arch=gfx950 -O3 -o l1rate l1rate.hip && ./l1rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WIDTH, int LANES_PER_LINE>  // WIDTH: floats per lane and load (1, 2, 4)
__global__ __launch_bounds__(256) void loads(const float* __restrict__ buf, float* out, int iters, unsigned long long mask)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const float* base = buf + (size_t)(blockIdx.x & 255u) * 4096u;  // 16 KB = 128 lines of 32 floats
    float acc = 0.0f;
    // float offset of the lane's first load: line (lane / LANES_PER_LINE), chunk (lane % LANES_PER_LINE)
    unsigned off = ((lane / LANES_PER_LINE) * 32u + (lane % LANES_PER_LINE) * WIDTH + wave * 1056u) & 4095u;
    if ((mask >> lane) & 1ull)
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                off = (off + 2080u) & 4095u;  // 65 lines on: another line, same chunk
                if (WIDTH == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(base + off);
                    acc += v.x + v.w;
                } else if (WIDTH == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(base + off);
                    acc += v.x + v.y;
                } else {
                    acc += base[off];
                }
            }
        }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WIDTH, int LPL>
void run(const char* name, const float* buf, float* out, unsigned long long mask, int wavesPerSimd)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2048, blocks = 256 * wavesPerSimd;
    hipLaunchKernelGGL((loads<WIDTH, LPL>), dim3(blocks), dim3(256), 0, 0, buf, out, 8, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((loads<WIDTH, LPL>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instsPerCu = (double)iters * 8 * 4 * wavesPerSimd;
    const double cyc = ms * 1e6 / instsPerCu * 2.4;
    printf("  %-52s %d waves/SIMD %8.3f ms  %6.1f cycles per wave-load per CU, %5.1f B/cycle/CU\n", name, wavesPerSimd, ms, cyc,
           __builtin_popcountll(mask) * WIDTH * 4.0 / cyc);
}

int main()
{
    float *buf, *out;
    hipMalloc(&buf, 256 * 4096 * sizeof(float));
    hipMemset(buf, 0, 256 * 4096 * sizeof(float));
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    for (int w : {1, 3}) {
        run<4, 8>("16 B per lane, 8 lanes per line (contiguous)", buf, out, ~0ull, w);
        run<4, 1>("16 B per lane, a line per lane", buf, out, ~0ull, w);
        run<4, 2>("16 B per lane, 2 lanes per line (32 B contiguous)", buf, out, ~0ull, w);
        run<4, 4>("16 B per lane, 4 lanes per line (64 B contiguous)", buf, out, ~0ull, w);
        run<2, 2>("8 B per lane, 2 lanes per line (16 B contiguous)", buf, out, ~0ull, w);
        run<2, 4>("8 B per lane, 4 lanes per line (32 B contiguous)", buf, out, ~0ull, w);
        run<4, 1>("16 B per lane, a line per lane, 43 lanes", buf, out, (1ull << 43) - 1, w);
        run<4, 1>("16 B per lane, a line per lane, 22 lanes", buf, out, (1ull << 22) - 1, w);
        run<4, 1>("16 B per lane, a line per lane, 22 scattered", buf, out, 0x9249249249249249ull, w);
        run<2, 1>("8 B per lane, a line per lane", buf, out, ~0ull, w);
        run<1, 1>("4 B per lane, a line per lane", buf, out, ~0ull, w);
        run<1, 32>("4 B per lane, contiguous", buf, out, ~0ull, w);
    }
    return 0;
}
