// Device math of jm_math.h against the host libm (tools only): sincos_ and tanh_ over log-spaced arguments.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../jiminy_amd/csrc/jm_math.h"

__global__ void k_eval(const double * x, double * s, double * c, double * t, int n, int div)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double ss = 0, cc = 0, tt = 0;
    // optionally inside divergent control flow, like the contact law
    if (!div || (i % 3) != 1)
    {
        jm::sincos_(x[i], &ss, &cc);
        tt = jm::tanh_(x[i]);
    }
    s[i] = ss; c[i] = cc; t[i] = tt;
}

int main()
{
    const int n = 1 << 16;
    std::vector<double> x(n), s(n), c(n), t(n);
    for (int i = 0; i < n; ++i)
    {
        const double u = (double)i / n;
        x[i] = std::pow(10.0, -9.0 + 11.0 * u) * ((i & 1) ? 1 : -1);
    }
    double *dx, *ds, *dc, *dt;
    hipMalloc(&dx, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dc, n * 8); hipMalloc(&dt, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (int div = 0; div < 2; ++div)
    {
        k_eval<<<n / 256, 256>>>(dx, ds, dc, dt, n, div);
        hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), dt, n * 8, hipMemcpyDeviceToHost);
        double es = 0, ec = 0, et = 0;
        for (int i = 0; i < n; ++i)
        {
            if (div && (i % 3) == 1) continue;
            es = std::fmax(es, std::fabs(s[i] - std::sin(x[i])));
            ec = std::fmax(ec, std::fabs(c[i] - std::cos(x[i])));
            et = std::fmax(et, std::fabs(t[i] - std::tanh(x[i])) / std::fabs(std::tanh(x[i])));
        }
        int wi = 0; double w = 0;
        for (int i = 0; i < n; ++i) { double e = std::fabs(t[i] - std::tanh(x[i])) / std::fabs(std::tanh(x[i])); if (!(div && (i % 3) == 1) && e > w) { w = e; wi = i; } }
        printf("worst tanh at x=%.17g: got %.17g want %.17g\n", x[wi], t[wi], std::tanh(x[wi]));
        for (int i = 0; i < n; i += n / 16) printf("  x=%.6g got %.17g want %.17g\n", x[i], t[i], std::tanh(x[i]));
        printf("divergent=%d: max |sin err| %.3e  |cos err| %.3e  tanh rel err %.3e\n", div, es, ec, et);
    }
    return 0;
}
