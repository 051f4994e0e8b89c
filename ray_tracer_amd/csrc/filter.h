// Bilateral filter kernel (see filter.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mi355rt {

// d_rgba, d_out: width * height * 4 doubles in HBM (may not alias)
hipError_t launch_bilateral(const double *d_rgba, int width, int height, int window, double sigma_d, double sigma_r, double *d_out, hipStream_t stream);

} // namespace mi355rt
