// image.hpp -- 8-bit RGBA raster standing in for QImage at the boundary of the Qt-free classes.
// The C-ABI takes images ALREADY scaled (QImage::scaledToWidth is Qt-version-specific
// arithmetic and stays in the Qt adapter, SURVEY.md section 7 "Third-party resampling").
#pragma once

#include <cstdint>
#include <vector>

struct Image {
	int w = 0, h = 0;
	std::vector<uint8_t> rgba;                       // w*h*4, byte order R,G,B,A
	Image() { }
	Image(int w_, int h_, uint8_t r = 255, uint8_t g = 255, uint8_t b = 255, uint8_t a = 255)
		: w(w_), h(h_), rgba(static_cast<size_t>(w_)*h_*4)
	{
		for (size_t i = 0; i < rgba.size(); i += 4) { rgba[i] = r; rgba[i+1] = g; rgba[i+2] = b; rgba[i+3] = a; }
	}
	bool isNull() const { return w <= 0 || h <= 0 || rgba.empty(); }
	int width() const { return w; }
	int height() const { return h; }
	uint8_t *pixel(int x, int y) { return &rgba[(static_cast<size_t>(y)*w + x)*4]; }
	const uint8_t *pixel(int x, int y) const { return &rgba[(static_cast<size_t>(y)*w + x)*4]; }
};

// mask.pixel(x,y) == WHITE  <=>  r = g = b = a = 255 (util/vectorimage.hpp:64-69)
inline std::vector<uint8_t> whiteMask(const Image &m, int w, int h) {
	std::vector<uint8_t> out(static_cast<size_t>(w)*h, 1);
	if (m.isNull()) return out;                      // null mask => VectorImage(w, h, WHITE)
	for (int y = 0; y < h && y < m.h; ++y)
		for (int x = 0; x < w && x < m.w; ++x) {
			const uint8_t *p = m.pixel(x, y);
			out[static_cast<size_t>(y)*w + x] = (p[0] == 255 && p[1] == 255 && p[2] == 255 && p[3] == 255) ? 1 : 0;
		}
	return out;
}
