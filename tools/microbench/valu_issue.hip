// Issue / dependency microbenchmark for gfx950 (tools only, not part of the product):
// cycles per wave64 v_fma_f64 / v_fma_f32 for C independent dependency chains per wave, at 1 and 2
// waves per SIMD.  Usage: ./valu_issue  (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template<class T, int C> __global__ void __launch_bounds__(256) k_chain(T * out, int iters, T a, T b)
{
    T x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = T(threadIdx.x + c);
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
    T s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<class T, int C> double run(int blocks, int iters)
{
    T * out;
    hipMalloc(&out, sizeof(T) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_chain<T, C><<<blocks, 256>>>(out, 10, T(1.0000001), T(1e-9));
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_chain<T, C><<<blocks, 256>>>(out, iters, T(1.0000001), T(1e-9));
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

template<class T, int C> void report(const char * name, double ghz)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps)
    {
        const int blocks = 256 * wps;   // 256 CUs x (4 waves per block = 1 per SIMD) x wps
        const double ms = run<T, C>(blocks, iters);
        const double inst_per_simd = double(iters) * 16 * C * wps;
        printf("%s chains=%d waves/SIMD=%d: %.3f ms, %.2f cycles per wave-instruction per SIMD\n", name, C, wps, ms,
               ms * 1e-3 * ghz * 1e9 / inst_per_simd);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    report<double, 1>("f64", ghz); report<double, 2>("f64", ghz); report<double, 3>("f64", ghz);
    report<double, 4>("f64", ghz); report<double, 8>("f64", ghz);
    report<float, 1>("f32", ghz); report<float, 2>("f32", ghz); report<float, 4>("f32", ghz);
    return 0;
}
