// ulp_bench.hip -- maximum error, in units in the last place of the correctly rounded result, of the device's
// fast_rcp / fast_rsqrt (fot_math.hpp: hardware estimate + one Newton step) over the argument ranges the planner
// feeds them.  Diagnostic; the number quoted in DESIGN.md section 2 comes from here.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../integrated_path_planning_amd/csrc/fot_math.hpp"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_eval(int n, const double *a, double *rcp, double *rsq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rcp[i] = fot::fast_rcp(a[i]);
    rsq[i] = fot::fast_rsqrt(a[i]);
}

static double ulp_err(double got, long double exact)
{
    int e;
    frexpl(exact, &e);                                  // exact = m 2^e, 0.5 <= m < 1: one ulp of a double there is 2^(e-53)
    return (double)(fabsl((long double)got - exact) / ldexpl(1.0L, e - 53));
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> a(n);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int i = 0; i < n; ++i) {
        const int kind = i & 3;
        if (kind == 0) a[i] = std::exp((u(rng) * 2.0 - 1.0) * 13.8);          // 1e-6 .. 1e6, log-uniform
        else if (kind == 1) a[i] = 0.05 + u(rng) * 3.0;                       // 1 - kappa d and h^2 around 1
        else if (kind == 2) a[i] = 1e-3 + u(rng) * 25.0;                      // s_dot
        else a[i] = (u(rng) < 0.5 ? -1.0 : 1.0) * std::exp((u(rng) * 2.0 - 1.0) * 6.9);   // signed (rcp only)
    }
    double *d_a, *d_r, *d_q;
    HIP_OK(hipMalloc(&d_a, sizeof(double) * n)); HIP_OK(hipMalloc(&d_r, sizeof(double) * n)); HIP_OK(hipMalloc(&d_q, sizeof(double) * n));
    HIP_OK(hipMemcpy(d_a, a.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    k_eval<<<(n + 255) / 256, 256>>>(n, d_a, d_r, d_q);
    HIP_OK(hipDeviceSynchronize());
    std::vector<double> r(n), q(n);
    HIP_OK(hipMemcpy(r.data(), d_r, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(q.data(), d_q, sizeof(double) * n, hipMemcpyDeviceToHost));
    double max_rcp = 0, max_rsq = 0, sum_rcp = 0, sum_rsq = 0;
    long n_rsq = 0;
    for (int i = 0; i < n; ++i) {
        const double er = ulp_err(r[i], 1.0L / (long double)a[i]);
        max_rcp = er > max_rcp ? er : max_rcp; sum_rcp += er;
        if (a[i] > 0.0) {
            const double eq = ulp_err(q[i], 1.0L / sqrtl((long double)a[i]));
            max_rsq = eq > max_rsq ? eq : max_rsq; sum_rsq += eq; ++n_rsq;
        }
    }
    printf("{\"n\": %d, \"fast_rcp_max_ulp\": %.3f, \"fast_rcp_mean_ulp\": %.3f, \"fast_rsqrt_max_ulp\": %.3f, \"fast_rsqrt_mean_ulp\": %.3f}\n",
           n, max_rcp, sum_rcp / n, max_rsq, sum_rsq / (double)n_rsq);
    return 0;
}
