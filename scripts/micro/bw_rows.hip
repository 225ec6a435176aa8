// Micro-benchmark: HBM read bandwidth for "16 rows per wave, SEG bytes per row per step"
// access patterns over a [51200][7680 B] array (the IQ layout), vs a linear stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int LPR, int AHEAD>   // lanes per row (each lane 16 B per load); rows per wave = 64/LPR
__global__ __launch_bounds__(256) void k_rows(const uint4* __restrict__ src, unsigned* sink, int total_rows)
{
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    constexpr int RPW = 64 / LPR;
    const int row = wave * RPW + lane / LPR;
    if (row >= total_rows) return;
    const uint4* p = src + (size_t)row * 480 + (lane % LPR);
    unsigned acc = 0;
    uint4 ring[AHEAD];
    constexpr int STEPS = 480 / LPR;
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) ring[k] = p[k * LPR];
    for (int s = 0; s < STEPS; s += AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            uint4 v = ring[k];
            int nx = s + k + AHEAD; if (nx >= STEPS) nx = STEPS - 1;
            ring[k] = p[nx * LPR];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_linear(const uint4* __restrict__ src, unsigned* sink, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * 256) { uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int LPR, int AHEAD> float run(const uint4* d, unsigned* sink, int rows)
{
    constexpr int RPW = 64 / LPR;
    int waves = (rows + RPW - 1) / RPW, blocks = (waves + 3) / 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_rows<LPR, AHEAD>), dim3(blocks), dim3(256), 0, 0, d, sink, rows);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_rows<LPR, AHEAD>), dim3(blocks), dim3(256), 0, 0, d, sink, rows);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}

int main()
{
    const int rows = 51200; const size_t bytes = (size_t)rows * 7680;
    uint4* d; unsigned* sink; CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(d, 1, bytes));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int g : {1024, 2048, 4096, 8192}) {
        hipLaunchKernelGGL(k_linear, dim3(g), dim3(256), 0, 0, d, sink, bytes / 16);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_linear, dim3(g), dim3(256), 0, 0, d, sink, bytes / 16);
        hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
        printf("linear grid %5d: %.1f us  %.2f TB/s\n", g, ms * 1e3, bytes / ms / 1e9);
    }
    float t;
    t = run<4, 2>(d, sink, rows);  printf("16 rows x  64 B, 2 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<4, 4>(d, sink, rows);  printf("16 rows x  64 B, 4 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<4, 8>(d, sink, rows);  printf("16 rows x  64 B, 8 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<8, 4>(d, sink, rows);  printf(" 8 rows x 128 B, 4 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<8, 8>(d, sink, rows);  printf(" 8 rows x 128 B, 8 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<16, 4>(d, sink, rows); printf(" 4 rows x 256 B, 4 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<16, 8>(d, sink, rows); printf(" 4 rows x 256 B, 8 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<32, 4>(d, sink, rows); printf(" 2 rows x 512 B, 4 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<32, 8>(d, sink, rows); printf(" 2 rows x 512 B, 8 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<2, 4>(d, sink, rows);  printf("32 rows x  32 B, 4 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    t = run<1, 4>(d, sink, rows);  printf("64 rows x  16 B, 4 ahead: %.1f us %.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    return 0;
}
