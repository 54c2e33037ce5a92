// Issue cost of the instruction classes of the k_quad evaluation loop on gfx950, and the clock the chip sustains
// under each (tools only, not part of the product).
//
// Every class runs as 8 independent chains per wave, 1 and 2 waves per SIMD, 256 CUs busy.  Cycles are counted by the
// wave itself with s_memtime (shader-clock counter) and the elapsed time with s_memrealtime (constant 100 MHz
// counter): cycles per wave-instruction per SIMD = d(memtime) / instructions of the SIMD's waves, sustained clock =
// d(memtime) / d(memrealtime) x 100 MHz.  Usage: ./op_issue      (prints a table; bench.py's VALU floor weights the
// ISA histogram of the kernel with these figures)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

enum Op { FMA64, MUL64, ADD64, FMA32, MOV32, MOV_DPP, CNDMASK, ADD_U32, LSHL_B64, RCP64, DS_READ64, OPS };
static const char * NAMES[OPS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_mov_b32", "v_mov_b32 dpp quad_perm",
                                  "v_cndmask_b32", "v_add_u32", "v_lshlrev_b64", "v_rcp_f64", "ds_read_b64"};

template<int OP> __device__ __forceinline__ void step(double (&x)[8], float (&f)[8], unsigned (&u)[8], double a, double b, const double * lds)
{
    const unsigned long long mask = 0x5555555555555555ull;   // lane mask of the selects (an SGPR pair, read-only)
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        if constexpr (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        else if constexpr (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
        else if constexpr (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
        else if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"((float)a), "v"((float)b));
        else if constexpr (OP == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(u[(c + 1) & 7]));
        else if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(u[c]) : "v"(u[(c + 1) & 7]));
        else if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) & 7]), "s"(mask));
        else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
        else if constexpr (OP == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(x[c]));
        else if constexpr (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[c]));
        else if constexpr (OP == DS_READ64) asm volatile("ds_read_b64 %0, %1" : "=v"(x[c]) : "v"((unsigned)(threadIdx.x * 8 + c * 2048)));
    }
    if constexpr (OP == DS_READ64) asm volatile("s_waitcnt lgkmcnt(0)");
    (void)lds;
}

template<int OP> __global__ void __launch_bounds__(256) k_op(double * out, uint64_t * clk, int iters, double a, double b)
{
    __shared__ double lds[2048];
    lds[threadIdx.x] = a;
    __syncthreads();
    double x[8];
    float f[8];
    unsigned u[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { x[c] = 1.0 + threadIdx.x + c; f[c] = (float)x[c]; u[c] = threadIdx.x * 8 + c; }
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int r = 0; r < 8; ++r) step<OP>(x, f, u, a, b, lds);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += x[c] + f[c] + u[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0)
    {
        const unsigned w = blockIdx.x * 4 + threadIdx.x / 64;
        clk[2 * w] = t1 - t0;
        clk[2 * w + 1] = r1 - r0;
    }
}

template<int OP> void report()
{
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps)
    {
        const int blocks = 256 * wps;   // 4 waves per block = one per SIMD; wps blocks per CU
        double * out;
        uint64_t * clk;
        hipMalloc(&out, sizeof(double) * blocks * 256);
        hipMalloc(&clk, sizeof(uint64_t) * 2 * blocks * 4);
        k_op<OP><<<blocks, 256>>>(out, clk, 10, 1.0000001, 1e-9);
        hipDeviceSynchronize();
        k_op<OP><<<blocks, 256>>>(out, clk, iters, 1.0000001, 1e-9);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(2 * blocks * 4);
        hipMemcpy(h.data(), clk, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int w = 0; w < blocks * 4; ++w)
        {
            cyc.push_back((double)h[2 * w]);
            ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(ghz.begin(), ghz.end());
        const double insts = (double)iters * 64;   // per wave
        printf("%-26s waves/SIMD=%d: %6.2f cycles per wave-instruction per SIMD (median wave: %.0f cycles / %.0f instructions / %d waves), "
               "sustained clock %.3f GHz (median; min %.3f max %.3f)\n", NAMES[OP], wps, cyc[cyc.size() / 2] / insts / wps,
               cyc[cyc.size() / 2], insts, wps, ghz[ghz.size() / 2], ghz.front(), ghz.back());
        hipFree(out); hipFree(clk);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, nominal %.2f GHz\n", p.name, p.multiProcessorCount, p.clockRate * 1e-6);
    report<FMA64>(); report<MUL64>(); report<ADD64>(); report<FMA32>(); report<MOV32>(); report<MOV_DPP>();
    report<CNDMASK>(); report<ADD_U32>(); report<LSHL_B64>(); report<RCP64>(); report<DS_READ64>();
    return 0;
}
