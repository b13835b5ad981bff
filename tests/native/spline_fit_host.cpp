// Host build of the product's spline_fit.hpp for the CPU test (tests/test_spline_fit.py):
//   g++ -O2 -shared -fPIC -ffp-contract=off spline_fit_host.cpp -o libsplinefit_host.so
#include <cmath>
#include "../../m3p2i_aip_amd/csrc/spline_fit.hpp"

extern "C" int sf_fit_eval(const double* y, int m, int k, double s, int n_out, double* out) {
    return m3::spline_fit_eval<double>(y, m, k, s, n_out, out, 1);
}
extern "C" void sf_fit_eval_batch(const double* y, int n_series, int m, int k, double s, int n_out, double* out,
                                  int* knots) {
    for (int i = 0; i < n_series; ++i)
        knots[i] = m3::spline_fit_eval<double>(y + (long long)i * m, m, k, s, n_out, out + (long long)i * n_out, 1);
}
