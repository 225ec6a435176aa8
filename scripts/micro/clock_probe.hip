// clock_probe.hip -- how fast s_memtime ticks in wall-clock terms under light and heavy load (gfx950):
// every wave runs a dependent v_add chain for `iters` x 256 instructions, stamps s_memtime and s_memrealtime
// (100 MHz) around it; the host times the launch with HIP events.
//   hipcc --offload-arch=gfx950 -O2 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define BODY(ins) REP8(REP8(ins))
__global__ void k(unsigned long long *out, float seed, int iters)
{
    float a0 = seed, b0 = 1.0001f;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int i = 0; i < iters; ++i) {
        BODY(asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b0));)
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (t1 - t0) + (a0 == 12345.678f ? 1 : 0);
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 16 * 256 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int wps : {1, 2, 4, 8}) {
        for (int iters : {200, 2000, 20000}) {
            const int blocks = 256 * 4 * wps;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 1.5f, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * blocks);
            hipMemcpy(h.data(), d, 16 * blocks, hipMemcpyDeviceToHost);
            double st = 0, sr = 0; for (int i = 0; i < blocks; ++i) { st += (double)h[2 * i]; sr += (double)h[2 * i + 1]; }
            st /= blocks; sr /= blocks;
            printf("%d wave(s)/SIMD, %6d x 256 instr: launch %.3f ms; per wave %.0f s_memtime ticks = %.3f ms of s_memrealtime (100 MHz) -> s_memtime at %.0f MHz; %.2f ticks, %.2f ns per instruction\n",
                   wps, iters, ms, st, sr / 1e5, st / (sr / 100.0), st / (iters * 256.0), sr * 10.0 / (iters * 256.0));
        }
    }
    return 0;
}
