// fused_rates.hip -- what one chunk of the fused kernel's front-end phase costs a wave (gfx950): the DPP chain of the DC sum
// and the convert / limit / discriminate part, each alone, at 1 / 2 / 4 / 5 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fused_rates.bin fused_rates.hip && ./fused_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define FU_STEP "s_nop 1\n\tv_add_f32_dpp %0, %0, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
                "v_add_f32 %0, %0, %7\n\tv_add_f32 %0, %0, %8\n\tv_add_f32 %0, %0, %9\n\t"
__device__ __forceinline__ void fu_chain(float &carry, float u0, float u1, float u2, float u3, float a0, float a1, float a2, float a3)
{
    float T;
    asm volatile("v_add_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %0, %0, %5\n\t"
                 FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP
                 "s_nop 1\n\tv_mov_b32_dpp %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(T), "+v"(carry) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}
__device__ __forceinline__ float s16_to_float(int x) { const float xf = (float)x; return __builtin_fmaf(xf, 0x1.f75104p-16f, xf * 0x1.aaa3aep-41f); }
__device__ __forceinline__ void limit(float &re, float &im)
{
    const float a = re * re + im * im;
    const float q = __builtin_amdgcn_rsqf(a), y0 = a * q, r = __builtin_fmaf(-y0, y0, a), m = __builtin_fmaf(r, q * 0.5f, y0);
    const float r0 = __builtin_amdgcn_rcpf(m), e = __builtin_fmaf(-m, r0, 1.0f), g = __builtin_fmaf(e, r0, r0);
    re = re * g; im = im * g;
}
template <int KIND>
__global__ void k(unsigned long long *out, const uint4 *in, float seed, int iters)
{
    float carry = seed, u0 = seed * 0.5f, u1 = seed * 0.25f, u2 = seed * 0.125f, u3 = seed * 0.3f;
    uint4 v = in[threadIdx.x & 63];
    float pre = seed, pim = seed * 0.7f, acc = 0.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0 || KIND == 2) fu_chain(carry, u0, u1, u2, u3, u0, u1, u2, u3);
        if (KIND == 1 || KIND == 2) {
            const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
            float re[4], im[4];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) { re[kx] = s16_to_float((int)(short)(ww[kx] & 0xFFFF)); im[kx] = s16_to_float((int)ww[kx] >> 16); limit(re[kx], im[kx]); }
            float uu[4];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const float z0re = kx ? re[kx - 1] : pre, z0im = kx ? im[kx - 1] : pim, z1re = kx > 1 ? re[kx - 2] : pim, z1im = kx > 1 ? im[kx - 2] : pre;
                uu[kx] = (z0re * (im[kx] - z1im) - z0im * (re[kx] - z1re)) * 0.5f;
            }
            pre = re[3]; pim = im[3];
            u0 = uu[0]; u1 = uu[1]; u2 = uu[2]; u3 = uu[3];
            v.x += 0x00010001u; v.y ^= __float_as_uint(uu[0]) & 0x00030003u; v.z += 3; v.w += 0x00050007u;
            if (KIND == 1) acc += uu[0] + uu[1] + uu[2] + uu[3];
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) + ((carry + acc + u0) == 12345.678f ? 1 : 0);
}
template <int KIND> void run(const char *name, unsigned long long *d, const uint4 *in)
{
    const int iters = 400;
    for (int wps : {1, 2, 4, 5, 8}) {
        const int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, in, 1.5f, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto x : h) s += (double)x;
        printf("%-44s %d wave(s)/SIMD: %8.1f ticks per chunk per wave => %7.1f per chunk per SIMD\n", name, wps, s / blocks / iters, s / blocks / iters / wps);
    }
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 8 * 256 * 4 * 8);
    uint4 *in; hipMalloc(&in, 64 * 16);
    std::vector<uint32_t> hv(256); for (int i = 0; i < 256; ++i) hv[i] = 0x12345678u * (i + 1) | 0x00400040u;
    hipMemcpy(in, hv.data(), 1024, hipMemcpyHostToDevice);
    run<0>("DC chain of one chunk (64 adds, 16 DPP)", d, in);
    run<1>("convert + limit + discriminate, 4 samples", d, in);
    run<2>("both", d, in);
    return 0;
}
