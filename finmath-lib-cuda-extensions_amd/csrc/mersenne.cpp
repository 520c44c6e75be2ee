// mersenne.cpp — the library side of the host Mersenne-Twister Brownian motion: thin wrappers over host/mersenne.hpp.
#include "../host/mersenne.hpp"

namespace fm {

double inverse_normal_cdf(double p) { return fmhost::inverseNormalCdf(p); }

void mersenne_increments(int32_t seed, int n_steps, int n_factors, int64_t n_paths, const double* dt, double* out) {
    fmhost::mersenneIncrements(seed, n_steps, n_factors, n_paths, dt, out);
}

} // namespace fm
