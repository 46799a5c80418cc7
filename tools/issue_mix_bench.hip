// Issue-cost probe, part 2 (round 4): the instruction classes the windowed gas optics' g-point loop is made of next to its fp32
// math -- 32-bit integer VALU, moves, scalar ALU, scalar branches, and VALU + SALU interleaved in ONE wave's stream (does a scalar
// instruction between two vector ones cost the wave a VALU slot?). Cycles per instruction per SIMD at 1..5 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int ITERS = 2048;
enum Op { ADDU32, MOV32, LSHLADD, ADDF32, FMA32, SALU8, VALU8_SALU8, VALU8_SALU16, VALU8_BR4, DSREAD_DEP, DSREAD_IND4, VALU8_NOP8, VALU8_WAIT8, FMA64_SALU8, FMA64 };

template<int OP>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* ticks, const float seed, const int one)
{
    __shared__ float lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed; lds[threadIdx.x + 512] = seed; lds[threadIdx.x + 768] = seed;
    float f0 = seed + lane, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3, i4 = lane + 4, i5 = lane + 5, i6 = lane + 6, i7 = lane + 7;
    int s0 = one, s1 = one + 1, s2 = one + 2, s3 = one + 3;
    const double dseed = seed;
    double d0 = dseed + lane, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll 1
    for (int it=0; it<ITERS; ++it)
    {
#define V8(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8" \
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane));
#define F8(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8" \
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(seed));
        if constexpr (OP == ADDU32) V8("v_add_u32")
        if constexpr (OP == MOV32) asm volatile("v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0"
                                                : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        if constexpr (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %8\nv_lshl_add_u32 %1, %1, 3, %8\nv_lshl_add_u32 %2, %2, 3, %8\nv_lshl_add_u32 %3, %3, 3, %8\nv_lshl_add_u32 %4, %4, 3, %8\nv_lshl_add_u32 %5, %5, 3, %8\nv_lshl_add_u32 %6, %6, 3, %8\nv_lshl_add_u32 %7, %7, 3, %8"
                                                  : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane));
        if constexpr (OP == ADDF32) F8("v_add_f32")
        if constexpr (OP == FMA32) asm volatile("v_fmac_f32 %0, %8, %8\nv_fmac_f32 %1, %8, %8\nv_fmac_f32 %2, %8, %8\nv_fmac_f32 %3, %8, %8\nv_fmac_f32 %4, %8, %8\nv_fmac_f32 %5, %8, %8\nv_fmac_f32 %6, %8, %8\nv_fmac_f32 %7, %8, %8"
                                                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(seed));
        if constexpr (OP == SALU8) asm volatile("s_add_u32 %0, %0, %1\ns_add_u32 %1, %1, %2\ns_add_u32 %2, %2, %3\ns_add_u32 %3, %3, %0\ns_add_u32 %0, %0, %1\ns_add_u32 %1, %1, %2\ns_add_u32 %2, %2, %3\ns_add_u32 %3, %3, %0"
                                                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
        if constexpr (OP == VALU8_SALU8)       // strictly alternating
            asm volatile("v_fmac_f32 %0, %12, %12\ns_add_u32 %8, %8, %9\nv_fmac_f32 %1, %12, %12\ns_add_u32 %9, %9, %10\nv_fmac_f32 %2, %12, %12\ns_add_u32 %10, %10, %11\nv_fmac_f32 %3, %12, %12\ns_add_u32 %11, %11, %8\n"
                         "v_fmac_f32 %4, %12, %12\ns_add_u32 %8, %8, %9\nv_fmac_f32 %5, %12, %12\ns_add_u32 %9, %9, %10\nv_fmac_f32 %6, %12, %12\ns_add_u32 %10, %10, %11\nv_fmac_f32 %7, %12, %12\ns_add_u32 %11, %11, %8"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(seed) : "scc");
        if constexpr (OP == VALU8_SALU16)      // two scalar instructions per vector one
            asm volatile("v_fmac_f32 %0, %12, %12\ns_add_u32 %8, %8, %9\ns_add_u32 %9, %9, %10\nv_fmac_f32 %1, %12, %12\ns_add_u32 %10, %10, %11\ns_add_u32 %11, %11, %8\nv_fmac_f32 %2, %12, %12\ns_add_u32 %8, %8, %9\ns_add_u32 %9, %9, %10\nv_fmac_f32 %3, %12, %12\ns_add_u32 %10, %10, %11\ns_add_u32 %11, %11, %8\n"
                         "v_fmac_f32 %4, %12, %12\ns_add_u32 %8, %8, %9\ns_add_u32 %9, %9, %10\nv_fmac_f32 %5, %12, %12\ns_add_u32 %10, %10, %11\ns_add_u32 %11, %11, %8\nv_fmac_f32 %6, %12, %12\ns_add_u32 %8, %8, %9\ns_add_u32 %9, %9, %10\nv_fmac_f32 %7, %12, %12\ns_add_u32 %10, %10, %11\ns_add_u32 %11, %11, %8"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(seed) : "scc");
        if constexpr (OP == VALU8_NOP8)        // s_nop 0 after every vector instruction
            asm volatile("v_fmac_f32 %0, %8, %8\ns_nop 0\nv_fmac_f32 %1, %8, %8\ns_nop 0\nv_fmac_f32 %2, %8, %8\ns_nop 0\nv_fmac_f32 %3, %8, %8\ns_nop 0\nv_fmac_f32 %4, %8, %8\ns_nop 0\nv_fmac_f32 %5, %8, %8\ns_nop 0\nv_fmac_f32 %6, %8, %8\ns_nop 0\nv_fmac_f32 %7, %8, %8\ns_nop 0"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(seed));
        if constexpr (OP == VALU8_WAIT8)       // s_waitcnt with nothing outstanding after every vector instruction
            asm volatile("v_fmac_f32 %0, %8, %8\ns_waitcnt vmcnt(0) lgkmcnt(0)\nv_fmac_f32 %1, %8, %8\ns_waitcnt vmcnt(0) lgkmcnt(0)\nv_fmac_f32 %2, %8, %8\ns_waitcnt vmcnt(0) lgkmcnt(0)\nv_fmac_f32 %3, %8, %8\ns_waitcnt vmcnt(0) lgkmcnt(0)\nv_fmac_f32 %4, %8, %8\ns_waitcnt vmcnt(0) lgkmcnt(0)\nv_fmac_f32 %5, %8, %8\ns_waitcnt vmcnt(0) lgkmcnt(0)\nv_fmac_f32 %6, %8, %8\ns_waitcnt vmcnt(0) lgkmcnt(0)\nv_fmac_f32 %7, %8, %8\ns_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(seed));
        if constexpr (OP == FMA64)
            asm volatile("v_fmac_f64 %0, %8, %8\nv_fmac_f64 %1, %8, %8\nv_fmac_f64 %2, %8, %8\nv_fmac_f64 %3, %8, %8\nv_fmac_f64 %4, %8, %8\nv_fmac_f64 %5, %8, %8\nv_fmac_f64 %6, %8, %8\nv_fmac_f64 %7, %8, %8"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dseed));
        if constexpr (OP == FMA64_SALU8)       // the solvers' case: fp64 vector instructions with a scalar one after each
            asm volatile("v_fmac_f64 %0, %12, %12\ns_add_u32 %8, %8, %9\nv_fmac_f64 %1, %12, %12\ns_add_u32 %9, %9, %10\nv_fmac_f64 %2, %12, %12\ns_add_u32 %10, %10, %11\nv_fmac_f64 %3, %12, %12\ns_add_u32 %11, %11, %8\n"
                         "v_fmac_f64 %4, %12, %12\ns_add_u32 %8, %8, %9\nv_fmac_f64 %5, %12, %12\ns_add_u32 %9, %9, %10\nv_fmac_f64 %6, %12, %12\ns_add_u32 %10, %10, %11\nv_fmac_f64 %7, %12, %12\ns_add_u32 %11, %11, %8"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(dseed) : "scc");
        if constexpr (OP == VALU8_BR4)         // eight vector instructions with four not-taken and taken scalar branches between them
            asm volatile("v_fmac_f32 %0, %9, %9\nv_fmac_f32 %1, %9, %9\ns_cmp_lg_u32 %8, 0\ns_cbranch_scc1 1f\nv_fmac_f32 %2, %9, %9\n1:\nv_fmac_f32 %2, %9, %9\nv_fmac_f32 %3, %9, %9\ns_cmp_eq_u32 %8, 0\ns_cbranch_scc1 2f\n2:\nv_fmac_f32 %4, %9, %9\nv_fmac_f32 %5, %9, %9\ns_cmp_lg_u32 %8, 0\ns_cbranch_scc1 3f\nv_fmac_f32 %6, %9, %9\n3:\nv_fmac_f32 %6, %9, %9\ns_cmp_eq_u32 %8, 0\ns_cbranch_scc1 4f\n4:\nv_fmac_f32 %7, %9, %9"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "s"(one), "v"(seed) : "scc");
        if constexpr (OP == DSREAD_DEP)        // eight dependent LDS reads (address from the value before): the LDS round trip
        {
            #pragma unroll
            for (int k=0; k<8; ++k) i0 = __float_as_int(lds[(i0 & 1023)]) & 1023;
        }
        if constexpr (OP == DSREAD_IND4)       // eight independent LDS reads issued together, one wait
        {
            const float a = lds[i0 & 1023], b = lds[(i0 + 64) & 1023], c = lds[(i0 + 128) & 1023], d = lds[(i0 + 192) & 1023];
            const float e = lds[(i0 + 256) & 1023], f = lds[(i0 + 320) & 1023], g = lds[(i0 + 384) & 1023], h = lds[(i0 + 448) & 1023];
            i0 = __float_as_int(a + b + c + d + e + f + g + h) & 1023;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x*blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + s0 + s1 + s2 + s3 + float(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template<int OP>
void run(const char* name, const double instr_per_iter, float* out, unsigned long long* ticks)
{
    std::printf("%-34s", name);
    for (int k : {1, 2, 3, 4, 5})
    {
        const int grid = 256*k;
        probe<OP><<<grid, 256>>>(out, ticks, 0.f, 0); hipDeviceSynchronize();
        probe<OP><<<grid, 256>>>(out, ticks, 0.f, 0); hipDeviceSynchronize();
        std::vector<unsigned long long> t(grid);
        hipMemcpy(t.data(), ticks, grid*sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(t.begin(), t.end());
        std::printf("  %d: %6.2f", k, double(t[grid/2]) / (ITERS*instr_per_iter) / k);
    }
    std::printf("   (cycles per instruction per SIMD at k waves per SIMD)\n");
}

int main()
{
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 2048*256*sizeof(float)); hipMalloc(&ticks, 2048*sizeof(unsigned long long));
    run<FMA32>("v_fmac_f32", 8, out, ticks);
    run<ADDF32>("v_add_f32", 8, out, ticks);
    run<ADDU32>("v_add_u32", 8, out, ticks);
    run<LSHLADD>("v_lshl_add_u32", 8, out, ticks);
    run<MOV32>("v_mov_b32", 8, out, ticks);
    run<SALU8>("s_add_u32", 8, out, ticks);
    run<VALU8_SALU8>("v_fmac + s_add alternating (per v)", 8, out, ticks);
    run<VALU8_SALU16>("v_fmac + 2 s_add (per v)", 8, out, ticks);
    run<VALU8_BR4>("10 v_fmac + 4 cmp/branch (per v)", 10, out, ticks);
    run<VALU8_NOP8>("v_fmac + s_nop 0 (per v)", 8, out, ticks);
    run<VALU8_WAIT8>("v_fmac + s_waitcnt (per v)", 8, out, ticks);
    run<FMA64>("v_fmac_f64", 8, out, ticks);
    run<FMA64_SALU8>("v_fmac_f64 + s_add alternating (per v)", 8, out, ticks);
    run<DSREAD_DEP>("ds_read dependent (per read)", 8, out, ticks);
    run<DSREAD_IND4>("ds_read 8 together (per read)", 8, out, ticks);
    return 0;
}
