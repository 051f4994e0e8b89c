// Host-side drop-in for the reference's Filter / BiliteralFilter (filter.h:6-27): same members, FilterImage runs on the
// MI355X through rt_filter_bilateral.
#pragma once
#include <vector>
#include "mi355rt.h"

namespace mi355host {

class Filter {
public:
	Filter(const std::vector<double> &img, int width, int height) : orig_img(img), img_width(width), img_height(height) {}
	virtual ~Filter() {}
	virtual void FilterImage(void) = 0;
	std::vector<double> result;

protected:
	std::vector<double> orig_img;
	int img_width;
	int img_height;
};

class BiliteralFilter : public Filter { // spelling as in the reference
public:
	// gpu: the context of the Renderer that owns the image (rt_create); the reference's class needs none
	BiliteralFilter(rt_context *gpu, const std::vector<double> &img, int width, int height) : Filter(img, width, height), gpu(gpu) {}
	void FilterImage(void) override
	{
		result.assign((size_t)img_width * img_height * 4, 0.0);
		ok = gpu && rt_filter_bilateral(gpu, orig_img.data(), img_width, img_height, window, sigma_d, sigma_r, result.data()) == RT_OK;
	}
	double sigma_d = 1.0;
	double sigma_r = 1.0;
	int window = 0;
	bool ok = false; // addition: false when the device call failed (rt_last_error has the text)

private:
	rt_context *gpu;
};

} // namespace mi355host
