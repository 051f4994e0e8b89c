// LensCamera (reference camera.h:101-209), host side: SetLens / set_Camera derive the fields the device's
// ray-generation kernel reads.  Field names follow the reference; the private ones are public here because
// the renderer copies them into rt_camera.
#pragma once
#include <cmath>
#include <vector>
#include "mi355rt.h"
#include "obj.h"

namespace mi355host {

class LensCamera {
public:
	void SetLens(double vfov, double focal_length) // camera.h:108-125
	{
		double theta = vfov * M_PI / 180;
		d = (0.5 * film_height) / std::tan(theta / 2.0);
		LensMakerCoefficient = 1.0 / (focal_length * (n[RT_NBINS / 2] - 1));
	}
	void set_Camera(dvec3 lookfrom, dvec3 lookat, dvec3 vup) // camera.h:128-145
	{
		origin = lookfrom;
		w = unit(sub(lookfrom, lookat));
		u = unit(cross(vup, w));
		v = cross(w, u);
		for (int k = 0; k < 3; k++) {
			film_lower_left_corner[k] = origin[k] - 0.5 * film_width * u[k] - 0.5 * film_height * v[k] + d * w[k];
			horizontal[k] = film_width * u[k];
			vertical[k] = film_height * v[k];
		}
	}
	double LensMakersFormula(double wavelength) const // camera.h:197-200
	{
		return 1.0 / ((n[(size_t)std::floor((wavelength - 400) / 10)] - 1) * LensMakerCoefficient);
	}
	rt_camera Pod() const
	{
		rt_camera c{};
		c.film_sensitivity = film_sensitivity; c.aperture = aperture; c.film_width = film_width; c.film_height = film_height;
		for (int k = 0; k < RT_NBINS; k++) c.n[k] = n[k];
		c.d = d; c.lens_maker_coefficient = LensMakerCoefficient;
		for (int k = 0; k < 3; k++) {
			c.origin[k] = origin[k]; c.film_lower_left_corner[k] = film_lower_left_corner[k];
			c.horizontal[k] = horizontal[k]; c.vertical[k] = vertical[k]; c.w[k] = w[k];
		}
		return c;
	}
	double film_sensitivity = 1.0, aperture = 0.5, film_width = 0.024, film_height = 0.024; // camera.h:189-192
	std::vector<double> n = std::vector<double>(RT_NBINS, 2.0);                                // camera.h:193
	double d = 0.0, LensMakerCoefficient = 0.0;
	dvec3 origin, film_lower_left_corner, horizontal, vertical, w, u, v;

private:
	static dvec3 sub(dvec3 a, dvec3 b) { return dvec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
	static dvec3 cross(dvec3 a, dvec3 b) { return dvec3(a[1] * b[2] - a[2] * b[1], -(a[0] * b[2] - a[2] * b[0]), a[0] * b[1] - a[1] * b[0]); }
	static dvec3 unit(dvec3 a) { double l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); return dvec3(a[0] / l, a[1] / l, a[2] / l); }
};

} // namespace mi355host
