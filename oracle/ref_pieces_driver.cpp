/*
 * ref_pieces_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * extern "C" entry points over the reference's OWN, UNMODIFIED sources
 *   util/vectorimage.cpp, util/lineiter.cpp,
 *   stereo/adaptiveweight.cpp, stereo/geodesicweight.cpp
 * which oracle/Makefile compiles where they lie under /root/reference into
 * oracle/_ref/libref_pieces.so (git-ignored, never copied into the repo).
 * These four files need only the C++ standard library and Qt5Gui's QImage, both
 * present in this image (/opt/conda), so no stand-in header or library is
 * written: the std headers the reference gets from its precompiled header are
 * named with -include on the command line.
 *
 * Used to (a) pin oracle/sr_oracle.c rows 1,2,3,7 of SURVEY.md 8(a) and
 * (b) generate tests/golden/ref_pieces_*.npz (tests/golden/make_ref_pieces.py).
 * The rest of the path (twoviewstereo.cpp, multiviewstereo.cpp, camera.cpp,
 * ray.cpp) needs Eigen/GSL, which this image lacks: unbuildable here.
 */
#include <QImage>
#include <cstring>

#include "util/vectorimage.hpp"
#include "util/lineiter.hpp"
#include "stereo/adaptiveweight.hpp"
#include "stereo/geodesicweight.hpp"

extern "C" {

/* rgba: w*h*4 bytes R,G,B,A -> VectorImage through the reference's fromQImage */
void *refp_image_create(const unsigned char *rgba, int w, int h) {
	QImage img(w, h, QImage::Format_ARGB32);
	for (int y = 0; y < h; ++y) {
		QRgb *line = reinterpret_cast<QRgb *>(img.scanLine(y));
		for (int x = 0; x < w; ++x) {
			const unsigned char *p = rgba + (static_cast<size_t>(y)*w + x)*4;
			line[x] = qRgba(p[0], p[1], p[2], p[3]);
		}
	}
	return new VectorImage(VectorImage::fromQImage(img));
}

void refp_image_free(void *h) { delete static_cast<VectorImage *>(h); }

/* out[4] = r,g,b,a ; returns isValid() */
int refp_image_pixel(void *h, int x, int y, double *out) {
	const RGBA &p = static_cast<VectorImage *>(h)->pixel(x, y);
	out[0] = p.r; out[1] = p.g; out[2] = p.b; out[3] = p.a;
	return p.isValid() ? 1 : 0;
}

int refp_image_sample(void *h, double x, double y, double *out) {
	const RGBA p = static_cast<VectorImage *>(h)->sample(x, y);
	out[0] = p.r; out[1] = p.g; out[2] = p.b; out[3] = p.a;
	return p.isValid() ? 1 : 0;
}

double refp_to_gray(double r, double g, double b) { return RGBA(r, g, b).toGray(); }

/* mask.pixel(x,y) == WHITE */
int refp_pixel_is_white(void *h, int x, int y) {
	return static_cast<VectorImage *>(h)->pixel(x, y) == WHITE ? 1 : 0;
}

/* doubles are converted to the int ctor parameters exactly as at the reference
 * call sites (twoviewstereo.cpp:1028, multiviewstereo.cpp:783) */
int refp_line_points(double x0, double y0, double x1, double y1, int clip, int w, int h,
                     int *out_xy, int max_pts)
{
	int n = 0;
	if (clip) {
		LineIterator iter(x0, y0, x1, y1, w, h);
		while (iter.hasNext()) {
			int tx, ty;
			iter.current(tx, ty);
			if (n < max_pts) { out_xy[2*n] = tx; out_xy[2*n + 1] = ty; }
			++n; ++iter;
		}
	} else {
		LineIterator iter(x0, y0, x1, y1);
		while (iter.hasNext()) {
			int tx, ty;
			iter.current(tx, ty);
			if (n < max_pts) { out_xy[2*n] = tx; out_xy[2*n + 1] = ty; }
			++n; ++iter;
		}
	}
	return n;
}

/* kind 0 = AdaptiveWeight, 1 = GeodesicWeight; out[(row+r)*(2r+1)+(col+r)] = w(row,col) */
void refp_weights(void *h, int cx, int cy, int radius, int kind, double *out) {
	const VectorImage &img = *static_cast<VectorImage *>(h);
	const int ws = 2*radius + 1;
	if (kind == 0) {
		AdaptiveWeight wf(radius);
		wf.init_weights(img, cx, cy);
		for (int row = -radius; row <= radius; ++row)
			for (int col = -radius; col <= radius; ++col)
				out[(row + radius)*ws + (col + radius)] = wf(row, col);
	} else {
		GeodesicWeight wf(radius);
		wf.init_weights(img, cx, cy);
		for (int row = -radius; row <= radius; ++row)
			for (int col = -radius; col <= radius; ++col)
				out[(row + radius)*ws + (col + radius)] = wf(row, col);
	}
}

} /* extern "C" */

/* ---- example-project ingest, exactly as the reference does it (needs only Qt):
 * image  = QImage(file).scaledToWidth(w*scale, Qt::SmoothTransformation) -> VectorImage::fromQImage
 *          (stereo/twoviewstereo.cpp:97, stereo/multiviewstereo.cpp:220-222)
 * mask   = alpha == 255 on a FastTransformation copy (multiviewstereo.cpp:225-234)
 * out_rgba: w*h*4 doubles->bytes R,G,B,A as fromQImage stores them; returns 0 on failure. */
extern "C" int refp_load_scaled(const char *path, double scale, int max_w, int max_h,
                                unsigned char *out_rgba, unsigned char *out_mask, int *out_w, int *out_h)
{
	QImage base(path);
	if (base.isNull()) return 0;
	QImage image = base.scaledToWidth(base.width() * scale, Qt::SmoothTransformation);
	VectorImage vi = VectorImage::fromQImage(image);
	const int w = vi.width(), h = vi.height();
	*out_w = w; *out_h = h;
	if (w > max_w || h > max_h) return 0;
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			const RGBA &p = vi.pixel(x, y);
			unsigned char *o = out_rgba + (static_cast<size_t>(y)*w + x)*4;
			o[0] = static_cast<unsigned char>(p.r); o[1] = static_cast<unsigned char>(p.g);
			o[2] = static_cast<unsigned char>(p.b); o[3] = static_cast<unsigned char>(p.a);
		}
	if (base.hasAlphaChannel()) {
		QImage mask = base.scaledToWidth(image.width(), Qt::FastTransformation);
		for (int y = 0; y < h; ++y)
			for (int x = 0; x < w; ++x)
				out_mask[static_cast<size_t>(y)*w + x] = (qAlpha(mask.pixel(x, y)) == 255) ? 1 : 0;
	} else {
		memset(out_mask, 1, static_cast<size_t>(w)*h);
	}
	return 1;
}
