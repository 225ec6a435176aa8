// valu_rates.hip -- cycles per wave-instruction of the VALU forms the kernels lean on (gfx950), measured with
// s_memtime around an unrolled stream of independent instructions, at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define BODY(ins) REP8(REP8(ins))
template <int KIND>
__global__ void k(unsigned long long *out, float seed, int iters)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {seed, seed}, p1 = {seed + 1, seed}, p2 = {seed + 2, seed}, p3 = {seed + 3, seed}, q = {1.0001f, 0.9999f};
    unsigned s0 = (unsigned)iters, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 * 3;
    unsigned long long sq = 0x3f8000003f800000ull + (unsigned)iters, m0 = (unsigned)iters;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { BODY(asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %5\n\tv_mul_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (KIND == 1) { BODY(asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }
        if (KIND == 2) { BODY(asm volatile("v_rsq_f32 %0, %0\n\tv_rsq_f32 %1, %1\n\tv_rsq_f32 %2, %2\n\tv_rsq_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 3) { BODY(asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 4) { BODY(asm volatile("v_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %1, %1\n\tv_cvt_f32_i32 %2, %2\n\tv_cvt_f32_i32 %3, %3" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 5) { BODY(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n\tv_pk_fma_f32 %1, %1, %4, %1\n\tv_pk_fma_f32 %2, %2, %4, %2\n\tv_pk_fma_f32 %3, %3, %4, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }
        if (KIND == 6) { BODY(asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b0));) }     // dependent chain
        if (KIND == 7) { BODY(asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(q));) }  // dependent chain
        if (KIND == 8) { BODY(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %1, vcc\n\tv_xor_b32 %3, %3, %1\n\tv_and_b32 %4, %4, %1" : "+v"(a0), "+v"(b0), "+v"(a2), "+v"(a3), "+v"(a4) :: "vcc");) }
        if (KIND == 9) { BODY(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 10) { BODY(asm volatile("s_add_u32 %0, %0, %4\n\ts_add_u32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_add_u32 %3, %3, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(s4) : "scc");) }
        if (KIND == 11) { BODY(asm volatile("v_mul_f32 %0, %0, %4\n\ts_add_u32 %2, %2, %5\n\tv_mul_f32 %1, %1, %4\n\ts_add_u32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : "v"(b0), "s"(s4) : "scc");) }
        if (KIND == 12) { BODY(asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sq));) }
        if (KIND == 13) { BODY(asm volatile("v_pk_mul_f32 %0, %0, %4\n\ts_add_u32 %2, %2, %5\n\tv_pk_mul_f32 %1, %1, %4\n\ts_add_u32 %3, %3, %5" : "+v"(p0), "+v"(p1), "+s"(s0), "+s"(s1) : "v"(q), "s"(s4) : "scc");) }
        if (KIND == 14) { BODY(asm volatile("s_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s4) : "scc");) }
        if (KIND == 15) { BODY(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_b64 %2, vcc, exec\n\tv_cmp_gt_f32 vcc, %0, %1\n\ts_or_b64 %2, %2, vcc" : "+v"(a0), "+v"(a1), "+s"(m0) :: "vcc", "scc");) }
        if (KIND == 16) { BODY(asm volatile("v_mul_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %3\n\tv_mul_f32 %2, %2, %3\n\ts_add_u32 %4, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(b0), "s"(s0), "s"(s4) : "scc");) }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    const float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + b0 + (float)(s0 + s1 + s2 + s3) + (float)m0;
    if (threadIdx.x == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) + (sink == 12345.678f ? 1 : 0);
}
template <int KIND> void run(const char *name, unsigned long long *d)
{
    const int iters = 200;                       // 64 groups x 4 instructions per iteration = 256 instructions
    for (int wps : {1, 2, 4, 8}) {               // waves per SIMD: blocks of 64 threads, 4 * wps per CU
        const int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 1.5f, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        const double per_wave = s / blocks / (iters * 256.0);
        printf("%-34s %d wave(s)/SIMD: %6.2f cycles per instruction per wave  => %5.2f per instruction per SIMD\n", name, wps, per_wave, per_wave / wps);
    }
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 8 * 256 * 4 * 8 * 2);
    run<0>("v_mul_f32 (independent)", d);
    run<1>("v_pk_mul_f32 (independent)", d);
    run<5>("v_pk_fma_f32 (4 chains)", d);
    run<2>("v_rsq_f32", d);
    run<3>("v_rcp_f32", d);
    run<4>("v_cvt_f32_i32", d);
    run<6>("v_add_f32 dependent chain", d);
    run<7>("v_pk_add_f32 dependent chain", d);
    run<8>("cndmask/xor/and mix", d);
    run<9>("v_mov_b32_dpp quad_perm", d);
    run<10>("s_add_u32 (independent)", d);
    run<14>("s_add_u32 dependent chain", d);
    run<11>("v_mul_f32 / s_add_u32 alternating", d);
    run<16>("3 v_mul_f32 + 1 s_add_u32", d);
    run<12>("v_pk_mul_f32 with SGPR-pair operand", d);
    run<13>("v_pk_mul_f32 / s_add_u32 alternating", d);
    run<15>("v_cmp->vcc / s_and vcc alternating", d);
    return 0;
}
