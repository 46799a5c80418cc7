// Issue-cost probe for the fp64-heavy solver kernels on gfx950: cycles per wave-instruction of the instructions those kernels
// are made of, with 1..4 resident waves per SIMD, independent vs dependent chains, and the cross-lane moves that the level
// scans can be built from (ds_bpermute vs DPP row_shr vs v_permlane16/32_swap). Build: hipcc --offload-arch=gfx950 -O3.
// Output: cycles per instruction per SIMD (s_memtime ticks, median block) -- throughput view, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

constexpr int ITERS = 2048;

enum Op { FMA_IND, FMA_DEP, ADD_IND, MUL_IND, RCP, RSQ, LDEXP, RNDNE, CVT_I32, CVT_F32, FMA32_IND, BPERM, DPP_MOV, PLANE16, PLANE32, MAXF, CNDMASK, SQRT_LIB, EXP_LEAN, MINMAX3, PKFMA32, PKMUL32, PKADD32, EXP32, RCP32, RSQ32, SQRT32, MAX32, CND32, FMA32_DEP, PKFMA32_DEP };

template<int OP>
__global__ void __launch_bounds__(256) probe(double* out, unsigned long long* ticks, const double seed)
{
    const int lane = threadIdx.x & 63;
    double a0 = seed + lane*1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    float f0 = float(a0), f1 = float(a1), f2 = float(a2), f3 = float(a3), f4 = float(a4), f5 = float(a5), f6 = float(a6), f7 = float(a7);
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    typedef float pf2 __attribute__((ext_vector_type(2)));
    pf2 p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = {f1, f0}, p5 = {f3, f2}, p6 = {f5, f4}, p7 = {f7, f6};
    const pf2 pm = {1.0000001f, 0.9999999f}, pc = {1e-9f, 2e-9f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll 1
    for (int it=0; it<ITERS; ++it)
    {
        if constexpr (OP == FMA_IND) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c); }
        if constexpr (OP == FMA_DEP) { a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); }
        if constexpr (OP == ADD_IND) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
        if constexpr (OP == MUL_IND) { a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m; }
        if constexpr (OP == RCP) { a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3); a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7); }
        if constexpr (OP == RSQ) { a0 = __builtin_amdgcn_rsq(a0); a1 = __builtin_amdgcn_rsq(a1); a2 = __builtin_amdgcn_rsq(a2); a3 = __builtin_amdgcn_rsq(a3); a4 = __builtin_amdgcn_rsq(a4); a5 = __builtin_amdgcn_rsq(a5); a6 = __builtin_amdgcn_rsq(a6); a7 = __builtin_amdgcn_rsq(a7); }
        if constexpr (OP == LDEXP) { a0 = __builtin_amdgcn_ldexp(a0, i0 & 1); a1 = __builtin_amdgcn_ldexp(a1, i0 & 1); a2 = __builtin_amdgcn_ldexp(a2, i0 & 1); a3 = __builtin_amdgcn_ldexp(a3, i0 & 1); a4 = __builtin_amdgcn_ldexp(a4, i0 & 1); a5 = __builtin_amdgcn_ldexp(a5, i0 & 1); a6 = __builtin_amdgcn_ldexp(a6, i0 & 1); a7 = __builtin_amdgcn_ldexp(a7, i0 & 1); }
        if constexpr (OP == RNDNE) { a0 = __builtin_rint(a0); a1 = __builtin_rint(a1); a2 = __builtin_rint(a2); a3 = __builtin_rint(a3); a4 = __builtin_rint(a4); a5 = __builtin_rint(a5); a6 = __builtin_rint(a6); a7 = __builtin_rint(a7);
                                    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if constexpr (OP == CVT_I32) { i0 += (int)a0; i1 += (int)a1; i2 += (int)a2; i3 += (int)a3; i0 += (int)a4; i1 += (int)a5; i2 += (int)a6; i3 += (int)a7;
                                      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if constexpr (OP == CVT_F32) { f0 = (float)a0; f1 = (float)a1; f2 = (float)a2; f3 = (float)a3; f4 = (float)a4; f5 = (float)a5; f6 = (float)a6; f7 = (float)a7;
                                      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); }
        if constexpr (OP == FMA32_IND) { f0 = fmaf(f0, 1.0000001f, 1e-9f); f1 = fmaf(f1, 1.0000001f, 1e-9f); f2 = fmaf(f2, 1.0000001f, 1e-9f); f3 = fmaf(f3, 1.0000001f, 1e-9f); f4 = fmaf(f4, 1.0000001f, 1e-9f); f5 = fmaf(f5, 1.0000001f, 1e-9f); f6 = fmaf(f6, 1.0000001f, 1e-9f); f7 = fmaf(f7, 1.0000001f, 1e-9f); }
        if constexpr (OP == BPERM)
        {   // 8 independent 32-bit ds_bpermute (a 64-bit shuffle is two of them)
            const int addr = ((lane + 16) & 63) << 2;
            i0 = __builtin_amdgcn_ds_bpermute(addr, i0); i1 = __builtin_amdgcn_ds_bpermute(addr, i1); i2 = __builtin_amdgcn_ds_bpermute(addr, i2); i3 = __builtin_amdgcn_ds_bpermute(addr, i3);
            i0 = __builtin_amdgcn_ds_bpermute(addr, i0); i1 = __builtin_amdgcn_ds_bpermute(addr, i1); i2 = __builtin_amdgcn_ds_bpermute(addr, i2); i3 = __builtin_amdgcn_ds_bpermute(addr, i3);
        }
        if constexpr (OP == DPP_MOV)
        {   // row_shr:1 = 0x111
            i0 = __builtin_amdgcn_update_dpp(i0, i0, 0x111, 0xf, 0xf, false); i1 = __builtin_amdgcn_update_dpp(i1, i1, 0x111, 0xf, 0xf, false);
            i2 = __builtin_amdgcn_update_dpp(i2, i2, 0x111, 0xf, 0xf, false); i3 = __builtin_amdgcn_update_dpp(i3, i3, 0x111, 0xf, 0xf, false);
            i0 = __builtin_amdgcn_update_dpp(i0, i0, 0x111, 0xf, 0xf, false); i1 = __builtin_amdgcn_update_dpp(i1, i1, 0x111, 0xf, 0xf, false);
            i2 = __builtin_amdgcn_update_dpp(i2, i2, 0x111, 0xf, 0xf, false); i3 = __builtin_amdgcn_update_dpp(i3, i3, 0x111, 0xf, 0xf, false);
        }
        if constexpr (OP == PLANE16)
        {
            auto r0 = __builtin_amdgcn_permlane16_swap(i0, i1, false, false); i0 = r0[0]; i1 = r0[1];
            auto r1 = __builtin_amdgcn_permlane16_swap(i2, i3, false, false); i2 = r1[0]; i3 = r1[1];
            auto r2 = __builtin_amdgcn_permlane16_swap(i0, i2, false, false); i0 = r2[0]; i2 = r2[1];
            auto r3 = __builtin_amdgcn_permlane16_swap(i1, i3, false, false); i1 = r3[0]; i3 = r3[1];
            auto r4 = __builtin_amdgcn_permlane16_swap(i0, i1, false, false); i0 = r4[0]; i1 = r4[1];
            auto r5 = __builtin_amdgcn_permlane16_swap(i2, i3, false, false); i2 = r5[0]; i3 = r5[1];
            auto r6 = __builtin_amdgcn_permlane16_swap(i0, i2, false, false); i0 = r6[0]; i2 = r6[1];
            auto r7 = __builtin_amdgcn_permlane16_swap(i1, i3, false, false); i1 = r7[0]; i3 = r7[1];
        }
        if constexpr (OP == PLANE32)
        {
            auto r0 = __builtin_amdgcn_permlane32_swap(i0, i1, false, false); i0 = r0[0]; i1 = r0[1];
            auto r1 = __builtin_amdgcn_permlane32_swap(i2, i3, false, false); i2 = r1[0]; i3 = r1[1];
            auto r2 = __builtin_amdgcn_permlane32_swap(i0, i2, false, false); i0 = r2[0]; i2 = r2[1];
            auto r3 = __builtin_amdgcn_permlane32_swap(i1, i3, false, false); i1 = r3[0]; i3 = r3[1];
            auto r4 = __builtin_amdgcn_permlane32_swap(i0, i1, false, false); i0 = r4[0]; i1 = r4[1];
            auto r5 = __builtin_amdgcn_permlane32_swap(i2, i3, false, false); i2 = r5[0]; i3 = r5[1];
            auto r6 = __builtin_amdgcn_permlane32_swap(i0, i2, false, false); i0 = r6[0]; i2 = r6[1];
            auto r7 = __builtin_amdgcn_permlane32_swap(i1, i3, false, false); i1 = r7[0]; i3 = r7[1];
        }
        if constexpr (OP == MAXF) { a0 = fmax(a0, c); a1 = fmax(a1, c); a2 = fmax(a2, c); a3 = fmax(a3, c); a4 = fmax(a4, c); a5 = fmax(a5, c); a6 = fmax(a6, c); a7 = fmax(a7, c);
                                   asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if constexpr (OP == CNDMASK) { a0 = (i0 & 1) ? a0 : a1; a2 = (i0 & 1) ? a2 : a3; a4 = (i0 & 1) ? a4 : a5; a6 = (i0 & 1) ? a6 : a7; a1 = (i0 & 1) ? a1 : a2; a3 = (i0 & 1) ? a3 : a4; a5 = (i0 & 1) ? a5 : a6; a7 = (i0 & 1) ? a7 : a0;
                                      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if constexpr (OP == SQRT_LIB) { a0 = sqrt(a0) + 2; a1 = sqrt(a1) + 2; a2 = sqrt(a2) + 2; a3 = sqrt(a3) + 2; a4 = sqrt(a4) + 2; a5 = sqrt(a5) + 2; a6 = sqrt(a6) + 2; a7 = sqrt(a7) + 2; }
        if constexpr (OP == EXP_LEAN) { a0 = exp(-a0) + 1; a1 = exp(-a1) + 1; a2 = exp(-a2) + 1; a3 = exp(-a3) + 1; a4 = exp(-a4) + 1; a5 = exp(-a5) + 1; a6 = exp(-a6) + 1; a7 = exp(-a7) + 1; }
        // round 4: the fp32 instructions the packed (two columns per lane) solvers are made of
        if constexpr (OP == PKFMA32) { p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc); p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc); p4 = __builtin_elementwise_fma(p4, pm, pc); p5 = __builtin_elementwise_fma(p5, pm, pc); p6 = __builtin_elementwise_fma(p6, pm, pc); p7 = __builtin_elementwise_fma(p7, pm, pc); }
        if constexpr (OP == PKFMA32_DEP) { p0 = __builtin_elementwise_fma(p0, pm, pc); p0 = __builtin_elementwise_fma(p0, pm, pc); p0 = __builtin_elementwise_fma(p0, pm, pc); p0 = __builtin_elementwise_fma(p0, pm, pc); p0 = __builtin_elementwise_fma(p0, pm, pc); p0 = __builtin_elementwise_fma(p0, pm, pc); p0 = __builtin_elementwise_fma(p0, pm, pc); p0 = __builtin_elementwise_fma(p0, pm, pc); }
        if constexpr (OP == FMA32_DEP) { f0 = fmaf(f0, 1.0000001f, 1e-9f); f0 = fmaf(f0, 1.0000001f, 1e-9f); f0 = fmaf(f0, 1.0000001f, 1e-9f); f0 = fmaf(f0, 1.0000001f, 1e-9f); f0 = fmaf(f0, 1.0000001f, 1e-9f); f0 = fmaf(f0, 1.0000001f, 1e-9f); f0 = fmaf(f0, 1.0000001f, 1e-9f); f0 = fmaf(f0, 1.0000001f, 1e-9f); }
        if constexpr (OP == PKMUL32) { p0 *= pm; p1 *= pm; p2 *= pm; p3 *= pm; p4 *= pm; p5 *= pm; p6 *= pm; p7 *= pm; }
        if constexpr (OP == PKADD32) { p0 += pc; p1 += pc; p2 += pc; p3 += pc; p4 += pc; p5 += pc; p6 += pc; p7 += pc; }
        if constexpr (OP == EXP32) { f0 = __builtin_amdgcn_exp2f(f0); f1 = __builtin_amdgcn_exp2f(f1); f2 = __builtin_amdgcn_exp2f(f2); f3 = __builtin_amdgcn_exp2f(f3); f4 = __builtin_amdgcn_exp2f(f4); f5 = __builtin_amdgcn_exp2f(f5); f6 = __builtin_amdgcn_exp2f(f6); f7 = __builtin_amdgcn_exp2f(f7); }
        if constexpr (OP == RCP32) { f0 = __builtin_amdgcn_rcpf(f0); f1 = __builtin_amdgcn_rcpf(f1); f2 = __builtin_amdgcn_rcpf(f2); f3 = __builtin_amdgcn_rcpf(f3); f4 = __builtin_amdgcn_rcpf(f4); f5 = __builtin_amdgcn_rcpf(f5); f6 = __builtin_amdgcn_rcpf(f6); f7 = __builtin_amdgcn_rcpf(f7); }
        if constexpr (OP == RSQ32) { f0 = __builtin_amdgcn_rsqf(f0); f1 = __builtin_amdgcn_rsqf(f1); f2 = __builtin_amdgcn_rsqf(f2); f3 = __builtin_amdgcn_rsqf(f3); f4 = __builtin_amdgcn_rsqf(f4); f5 = __builtin_amdgcn_rsqf(f5); f6 = __builtin_amdgcn_rsqf(f6); f7 = __builtin_amdgcn_rsqf(f7); }
        if constexpr (OP == SQRT32) { f0 = __builtin_amdgcn_sqrtf(f0); f1 = __builtin_amdgcn_sqrtf(f1); f2 = __builtin_amdgcn_sqrtf(f2); f3 = __builtin_amdgcn_sqrtf(f3); f4 = __builtin_amdgcn_sqrtf(f4); f5 = __builtin_amdgcn_sqrtf(f5); f6 = __builtin_amdgcn_sqrtf(f6); f7 = __builtin_amdgcn_sqrtf(f7); }
        if constexpr (OP == MAX32) { f0 = fmaxf(f0, 1e-9f); f1 = fmaxf(f1, 1e-9f); f2 = fmaxf(f2, 1e-9f); f3 = fmaxf(f3, 1e-9f); f4 = fmaxf(f4, 1e-9f); f5 = fmaxf(f5, 1e-9f); f6 = fmaxf(f6, 1e-9f); f7 = fmaxf(f7, 1e-9f);
                                    asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); }
        if constexpr (OP == CND32) { f0 = (i0 & 1) ? f0 : f1; f2 = (i0 & 1) ? f2 : f3; f4 = (i0 & 1) ? f4 : f5; f6 = (i0 & 1) ? f6 : f7; f1 = (i0 & 1) ? f1 : f2; f3 = (i0 & 1) ? f3 : f4; f5 = (i0 & 1) ? f5 : f6; f7 = (i0 & 1) ? f7 : f0;
                                    asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + p4[0] + p4[1] + p5[0] + p5[1] + p6[0] + p6[1] + p7[0] + p7[1] + i0 + i1 + i2 + i3;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template<int OP>
void run(const char* name, double* out, unsigned long long* ticks, const int lds_pad_blocks_per_cu)
{
    // blocks of 256 threads = one wave per SIMD; k blocks per CU -> k waves per SIMD (grid = 256 CUs * k, all resident)
    for (int k : {1, 2, 3, 4})
    {
        const int grid = 256*k;
        probe<OP><<<grid, 256>>>(out, ticks, 1.5);
        hipDeviceSynchronize();
        probe<OP><<<grid, 256>>>(out, ticks, 1.5);
        hipDeviceSynchronize();
        std::vector<unsigned long long> t(grid);
        hipMemcpy(t.data(), ticks, grid*sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(t.begin(), t.end());
        const double per_instr_wave = double(t[grid/2]) / (ITERS*8.0);          // cycles between two instructions of ONE wave
        std::printf("%-10s waves/SIMD %d : %6.2f cyc per instr per wave  -> %6.2f cyc per instr per SIMD\n", name, k, per_instr_wave, per_instr_wave/k);
    }
}

int main()
{
    double* out; unsigned long long* ticks;
    hipMalloc(&out, 1024*256*sizeof(double)); hipMalloc(&ticks, 1024*sizeof(unsigned long long));
    run<FMA_IND>("fma64_ind", out, ticks, 0);
    run<FMA_DEP>("fma64_dep", out, ticks, 0);
    run<ADD_IND>("add64", out, ticks, 0);
    run<MUL_IND>("mul64", out, ticks, 0);
    run<RCP>("rcp64", out, ticks, 0);
    run<RSQ>("rsq64", out, ticks, 0);
    run<LDEXP>("ldexp64", out, ticks, 0);
    run<RNDNE>("rndne64", out, ticks, 0);
    run<CVT_I32>("cvt_i32+add", out, ticks, 0);
    run<CVT_F32>("cvt_f32", out, ticks, 0);
    run<MAXF>("max64", out, ticks, 0);
    run<CNDMASK>("cndmask64", out, ticks, 0);
    run<FMA32_IND>("fma32_ind", out, ticks, 0);
    run<BPERM>("bpermute32", out, ticks, 0);
    run<DPP_MOV>("dpp_mov32", out, ticks, 0);
    run<PLANE16>("permlane16", out, ticks, 0);
    run<PLANE32>("permlane32", out, ticks, 0);
    run<SQRT_LIB>("sqrt_lib+add", out, ticks, 0);
    run<EXP_LEAN>("exp_lib+add", out, ticks, 0);
    run<FMA32_DEP>("fma32_dep", out, ticks, 0);
    run<PKFMA32>("pk_fma32", out, ticks, 0);
    run<PKFMA32_DEP>("pk_fma32_dep", out, ticks, 0);
    run<PKMUL32>("pk_mul32", out, ticks, 0);
    run<PKADD32>("pk_add32", out, ticks, 0);
    run<EXP32>("exp32", out, ticks, 0);
    run<RCP32>("rcp32", out, ticks, 0);
    run<RSQ32>("rsq32", out, ticks, 0);
    run<SQRT32>("sqrt32", out, ticks, 0);
    run<MAX32>("max32", out, ticks, 0);
    run<CND32>("cndmask32", out, ticks, 0);
    return 0;
}
