// TEST INFRASTRUCTURE ONLY.  C entry point over the reference's own bilateral filter (filter.cpp, compiled unmodified
// from the reference checkout) so that tests can pin the oracle's restatement and the device kernel against it.
#include "filter.h"
#include <cstring>

extern "C" void ref_bilateral(const double *img, int width, int height, int window, double sigma_d, double sigma_r, double *result)
{
	std::vector<double> in(img, img + (size_t)width * height * 4);
	BiliteralFilter f(in, width, height); // filter.cpp:9-11
	f.sigma_d = sigma_d;
	f.sigma_r = sigma_r;
	f.window = window;
	f.FilterImage(); // filter.cpp:13-51
	std::memcpy(result, f.result.data(), sizeof(double) * f.result.size());
}
