// stereo_qt.cpp -- see stereo_qt.hpp.  No arithmetic of the matching path happens here: images are decoded and
// scaled by Qt exactly where the reference does it, everything else is the C-ABI.
#include "stereo_qt.hpp"

#include <QtCore/QFileInfo>
#include <QtGui/QColor>

#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>

namespace srq {

Raster rasterFromQImage(const QImage &in) {
	Raster r;
	if (in.isNull()) return r;
	const QImage img = in.depth() == 32 ? in : in.convertToFormat(QImage::Format_ARGB32);
	r.w = img.width(); r.h = img.height();
	r.rgba.resize(static_cast<size_t>(r.w)*r.h*4);
	for (int y = 0; y < r.h; ++y) {
		const QRgb *scanline = reinterpret_cast<const QRgb *>(img.constScanLine(y));
		unsigned char *out = &r.rgba[static_cast<size_t>(y)*r.w*4];
		for (int x = 0; x < r.w; ++x, out += 4) {
			const QRgb rgb = scanline[x];
			out[0] = static_cast<unsigned char>(qRed(rgb)); out[1] = static_cast<unsigned char>(qGreen(rgb));
			out[2] = static_cast<unsigned char>(qBlue(rgb)); out[3] = static_cast<unsigned char>(qAlpha(rgb));
		}
	}
	return r;
}

void writePLY(const std::string &path, size_t npoints, const double *xyz, const unsigned char *rgb) {
	std::ofstream lout(path.c_str());
	lout << "ply\n" << "format ascii 1.0\n" << "element vertex " << npoints << "\n"
	     << "property float x\n" << "property float y\n" << "property float z\n"
	     << "property uchar diffuse_red\n" << "property uchar diffuse_green\n" << "property uchar diffuse_blue\n"
	     << "end_header\n";
	for (size_t i = 0; i < npoints; ++i)
		lout << xyz[i*3] << ' ' << xyz[i*3 + 1] << ' ' << xyz[i*3 + 2] << ' ' << static_cast<int>(rgb[i*3]) << ' '
		     << static_cast<int>(rgb[i*3 + 1]) << ' ' << static_cast<int>(rgb[i*3 + 2]) << '\n';
}

void writePLY(const std::string &path, size_t npoints, const double *xyz, const int *rgb) {
	std::ofstream lout(path.c_str());
	lout << "ply\n" << "format ascii 1.0\n" << "element vertex " << npoints << "\n"
	     << "property float x\n" << "property float y\n" << "property float z\n"
	     << "property uchar diffuse_red\n" << "property uchar diffuse_green\n" << "property uchar diffuse_blue\n"
	     << "end_header\n";
	for (size_t i = 0; i < npoints; ++i)
		lout << xyz[i*3] << ' ' << xyz[i*3 + 1] << ' ' << xyz[i*3 + 2] << ' ' << rgb[i*3] << ' ' << rgb[i*3 + 1] << ' ' << rgb[i*3 + 2] << '\n';
}

std::vector<unsigned char> whiteMask(const Raster &m) {
	std::vector<unsigned char> out(static_cast<size_t>(m.w)*m.h);
	for (size_t i = 0; i < out.size(); ++i) {
		const unsigned char *p = &m.rgba[i*4];
		out[i] = (p[0] == 255 && p[1] == 255 && p[2] == 255 && p[3] == 255) ? 1 : 0;
	}
	return out;
}

bool ingestViewFile(const QString &file, double imageScale, Raster &image, std::vector<unsigned char> &mask) {
	if (!QFileInfo(file).exists()) return false;
	QImage baseImage(file);
	if (baseImage.isNull()) return false;
	const QImage scaled = baseImage.scaledToWidth(static_cast<int>(baseImage.width() * imageScale), Qt::SmoothTransformation);
	image = rasterFromQImage(scaled);
	mask.assign(static_cast<size_t>(image.w)*image.h, 1);
	if (baseImage.hasAlphaChannel()) {
		const QImage m = baseImage.scaledToWidth(scaled.width(), Qt::FastTransformation);
		for (int y = 0; y < m.height() && y < image.h; ++y)
			for (int x = 0; x < m.width() && x < image.w; ++x)
				if (qAlpha(m.pixel(x, y)) != 255) mask[static_cast<size_t>(y)*image.w + x] = 0;   // not fully opaque: ignored
	}
	return true;
}

} // namespace srq

// ------------------------------------------------------------------ TwoViewStereo
static int g_device = 0;
void TwoViewStereo::setDevice(int ordinal) { g_device = ordinal; }
void MultiViewStereo::setDevice(int ordinal) { g_device = ordinal; }

TwoViewStereo::TwoViewStereo(CameraPtr leftView_, QImage left_, QImage leftMask_,
                             CameraPtr rightView_, QImage right_, QImage rightMask_,
                             double minDepth_, double maxDepth_, int numDepthLevels_, double imageScale_)
	: leftView(leftView_), rightView(rightView_)
	, minDepth(minDepth_), maxDepth(maxDepth_), numDepthLevels(numDepthLevels_), imageScale(imageScale_), ctx_(nullptr)
{
	const int deviceOrdinal = g_device;
	memset(&leftCam, 0, sizeof(leftCam)); memset(&rightCam, 0, sizeof(rightCam));
	if (leftView) leftCam = srq::cameraInfo(*leftView).camera;         // the host application's glue (stereo_qt.hpp)
	if (rightView) rightCam = srq::cameraInfo(*rightView).camera;
	// twoviewstereo.cpp:89-124: images and masks are smooth-scaled to width*imageScale; a null mask is all WHITE
	left = srq::rasterFromQImage(left_.scaledToWidth(static_cast<int>(left_.width() * imageScale), Qt::SmoothTransformation));
	right = srq::rasterFromQImage(right_.scaledToWidth(static_cast<int>(right_.width() * imageScale), Qt::SmoothTransformation));
	if (!leftMask_.isNull())
		leftMask = srq::whiteMask(srq::rasterFromQImage(leftMask_.scaledToWidth(static_cast<int>(leftMask_.width() * imageScale), Qt::SmoothTransformation)));
	else leftMask.assign(static_cast<size_t>(left.w)*left.h, 1);
	if (!rightMask_.isNull())
		rightMask = srq::whiteMask(srq::rasterFromQImage(rightMask_.scaledToWidth(static_cast<int>(rightMask_.width() * imageScale), Qt::SmoothTransformation)));
	else rightMask.assign(static_cast<size_t>(right.w)*right.h, 1);
	const double NaN = std::numeric_limits<double>::quiet_NaN();
	computedDepthLeft.assign(static_cast<size_t>(left.w)*left.h, NaN);
	computedDepthRight.assign(static_cast<size_t>(left.w)*left.h, NaN);      // sized from the LEFT image, as :119
	srh_params_twoview_defaults(&params_);
	if (srh_create(deviceOrdinal, &ctx_) != SRH_OK) { error_ = srh_last_error(); ctx_ = nullptr; }
}

TwoViewStereo::~TwoViewStereo() { if (ctx_) srh_destroy(ctx_); }

QImage TwoViewStereo::colorize(const DepthMap &d, int w, int h) const {
	// twoviewstereo.cpp:128-146 + VectorImage::toQImage(.., false)
	QImage out(w, h, QImage::Format_ARGB32);
	for (int y = 0; y < h; ++y) {
		QRgb *scanline = reinterpret_cast<QRgb *>(out.scanLine(y));
		for (int x = 0; x < w; ++x) {
			const double depth = d[static_cast<size_t>(y)*w + x];
			QRgb px = qRgba(0, 0, 0, 255);
			if (!std::isnan(depth) && !std::isinf(depth)) {
				const double t = (depth - minDepth) / (maxDepth - minDepth);
				if (t > 1.1) px = qRgba(255, 255, 255, 255);
				else if (!(t < 1e-5)) { const QColor c = QColor::fromHsvF(2.0 * t / 3.0, 1.0, 1.0); px = qRgba(c.red(), c.green(), c.blue(), 255); }
			}
			scanline[x] = px;
		}
	}
	return out;
}

bool TwoViewStereo::uploadViews() const {
	if (uploaded_) return true;
	if (srh_view_upload(ctx_, 0, left.w, left.h, left.rgba.data(), leftMask.data(), &leftCam) != SRH_OK ||
	    srh_view_upload(ctx_, 1, right.w, right.h, right.rgba.data(), rightMask.data(), &rightCam) != SRH_OK) {
		error_ = srh_last_error(); return false;
	}
	uploaded_ = true;
	return true;
}

std::vector<std::array<double, 3> > TwoViewStereo::curveOfPixel(int x, int y, bool fromLeft) const {
	std::vector<std::array<double, 3> > curve;
	if (!ctx_ || left.w <= 0 || right.w <= 0 || !uploadViews()) return curve;
	srh_params p = params_;
	p.min_depth = minDepth; p.max_depth = maxDepth; p.num_depth_levels = numDepthLevels; p.image_scale = imageScale;
	const int32_t xy[2] = { x, y };
	int32_t count = 0;
	std::vector<int32_t> pts(2*256);
	for (int pass = 0; pass < 2; ++pass) {                  // the curve's length is only known afterwards
		if (srh_epipolar_curves(ctx_, fromLeft ? 0 : 1, fromLeft ? 1 : 0, &p, 0, 1, xy, pts.data(),
		                        static_cast<int>(pts.size()/2), &count) != SRH_OK) { error_ = srh_last_error(); return curve; }
		if (static_cast<size_t>(count) <= pts.size()/2) break;
		pts.resize(2*static_cast<size_t>(count));
	}
	curve.resize(static_cast<size_t>(count));
	for (int k = 0; k < count; ++k) curve[k] = { static_cast<double>(pts[2*k]), static_cast<double>(pts[2*k + 1]), 1.0 };
	return curve;
}

void TwoViewStereo::computeDepthMaps() {
	// twoviewstereo.cpp:150-227: cost volumes (steps 1, 3), cross-check (5), colourise, finished (8).
	// Errors are silent, as in the reference; lastError() keeps the library's message.
	if (!ctx_ || left.w <= 0 || right.w <= 0) return;
	if (left.w != right.w || left.h != right.h) { error_ = "views differ in size"; return; }
	params_.min_depth = minDepth; params_.max_depth = maxDepth;
	params_.num_depth_levels = numDepthLevels; params_.image_scale = imageScale;
	if (!uploadViews()) return;
	emit progressUpdate(1);
	emit stageUpdate("Computing cost volume for left image...");
	if (srh_twoview_wta(ctx_, 0, 1, &params_, 0, 0) != SRH_OK) { error_ = srh_last_error(); return; }
	if (isCancelled()) return;
	emit progressUpdate(3);
	emit stageUpdate("Computing cost volume for right image...");
	if (srh_twoview_wta(ctx_, 1, 0, &params_, 0, 0) != SRH_OK) { error_ = srh_last_error(); return; }
	if (isCancelled()) return;
	emit progressUpdate(5);
	emit stageUpdate("Detecting inconsistencies...");
	if (srh_twoview_cross_check(ctx_, 0, 1, &params_) != SRH_OK ||
	    srh_view_depth_download(ctx_, 0, computedDepthLeft.data()) != SRH_OK ||
	    srh_view_depth_download(ctx_, 1, computedDepthRight.data()) != SRH_OK) { error_ = srh_last_error(); return; }
	if (isCancelled()) return;
	resultLeft = colorize(computedDepthLeft, left.w, left.h);
	resultRight = colorize(computedDepthRight, right.w, right.h);
	emit progressUpdate(8);
	emit stageUpdate("Finished!");
}

// ------------------------------------------------------------------ MultiViewStereo
MultiViewStereo::MultiViewStereo()
	: minDepth(0), maxDepth(0), crossCheckThreshold(0), imageScale(1), numDepthLevels(0), ctx_(nullptr)
{
	const int deviceOrdinal = g_device;
	srh_params_mvs_defaults(&params_);
	srh_mrf_params_defaults(&mrfParams_);
	if (srh_create(deviceOrdinal, &ctx_) != SRH_OK) { error_ = srh_last_error(); ctx_ = nullptr; }
}

MultiViewStereo::~MultiViewStereo() { if (ctx_) srh_destroy(ctx_); }

void MultiViewStereo::initialize(ProjectPtr project_, ImageSetPtr imageSet, const std::vector<CameraPtr> &views,
                                 double minDepth_, double maxDepth_, int numDepthLevels_, double crossCheckThreshold_, double imageScale_)
{
	// multiviewstereo.cpp:193-247: every non-null view with a default image in the set becomes a record
	std::vector<View> records;
	for (size_t index = 0; index < views.size(); ++index) if (views[index] && imageSet) {
		const srq::CameraInfo info = srq::cameraInfo(*views[index]);
		View v;
		v.id = info.id; v.name = info.name; v.camera = info.camera; v.ptr = views[index];
		v.file = srq::defaultImageFile(*imageSet, views[index]);
		if (!v.file.isEmpty()) records.push_back(v);
	}
	initialize(records, minDepth_, maxDepth_, numDepthLevels_, crossCheckThreshold_, imageScale_);
	project = project_;
	imageSet_ = imageSet;
}

void MultiViewStereo::initialize(const std::vector<View> &views, double minDepth_, double maxDepth_, int numDepthLevels_,
                                 double crossCheckThreshold_, double imageScale_)
{
	project.reset(); imageSet_.reset();
	minDepth = minDepth_; maxDepth = maxDepth_; numDepthLevels = numDepthLevels_;
	crossCheckThreshold = crossCheckThreshold_; imageScale = imageScale_;
	views_.clear(); images.clear(); masks.clear(); results.clear(); computedDepths.clear();
	const double NaN = std::numeric_limits<double>::quiet_NaN();
	for (size_t i = 0; i < views.size() && views_.size() < static_cast<size_t>(SRH_MAX_VIEWS); ++i) {
		srq::Raster img; std::vector<unsigned char> m;
		if (!srq::ingestViewFile(views[i].file, imageScale, img, m)) continue;       // no file: the view is skipped (:214)
		images.push_back(img); masks.push_back(m);
		results.push_back(QImage());
		computedDepths.push_back(std::vector<double>(static_cast<size_t>(img.w)*img.h, NaN));
		views_.push_back(views[i]);
	}
}

void MultiViewStereo::colorize(int v) {
	// multiviewstereo.cpp:252-276, 382-395: black = close, white = far; NaN / INF / unknown (-1) and masked-out: white
	const int w = images[v].w, h = images[v].h;
	QImage out(w, h, QImage::Format_ARGB32);
	for (int y = 0; y < h; ++y) {
		QRgb *scanline = reinterpret_cast<QRgb *>(out.scanLine(y));
		for (int x = 0; x < w; ++x) {
			int gray = 255;
			if (masks[v][static_cast<size_t>(y)*w + x] == 1) {
				const double d = computedDepths[v][static_cast<size_t>(y)*w + x];
				if (!std::isnan(d) && !std::isinf(d) && !(d + 1e-5 < minDepth)) {
					const double t = std::min(1.0, std::max(0.0, (d - minDepth) / (maxDepth - minDepth)));
					gray = static_cast<int>(255*t);          // RGBA(255*t) stored as double, truncated by qRgba's int parameters
				}
			}
			scanline[x] = qRgba(gray, gray, gray, 255);
		}
	}
	results[v] = out;
}

void MultiViewStereo::runTask() {
	// multiviewstereo.cpp:325-475: neighbours, initial estimates (steps 0..V-1), colourise, cross-checks in view
	// order (steps V..2V-1), colourise.  Invalid state: silent return (:326-327).
	const int V = static_cast<int>(views_.size());
	if (!ctx_ || V == 0) return;
	params_.min_depth = minDepth; params_.max_depth = maxDepth; params_.num_depth_levels = numDepthLevels;
	params_.image_scale = imageScale; params_.cross_check_threshold = crossCheckThreshold;
	std::vector<srh_camera> cams(V);
	std::vector<int32_t> slotIds(V);
	for (int v = 0; v < V; ++v) {
		cams[v] = views_[v].camera; slotIds[v] = v;
		if (srh_view_upload(ctx_, v, images[v].w, images[v].h, images[v].rgba.data(), masks[v].data(), &cams[v]) != SRH_OK) {
			error_ = srh_last_error(); return;
		}
	}
	const int nn = params_.num_neighbours > 0 ? params_.num_neighbours : 1;
	std::vector<int32_t> neigh(static_cast<size_t>(V)*nn), count(V);
	if (srh_mvs_neighbours(V, cams.data(), &params_, neigh.data(), count.data()) != SRH_OK) { error_ = srh_last_error(); return; }
	int step = 0;
	for (int v = 0; v < V; ++v) {
		if (isCancelled()) return;
		emit progressUpdate(step++);
		emit stageUpdate(tr("Computing cost volume for camera %1").arg(views_[v].name));
		const int32_t *nb = &neigh[static_cast<size_t>(v)*nn];
		if ((useMrf_ ? srh_mvs_initial_estimate_peaks(ctx_, v, nb, count[v], &params_)                        // #ifdef USE_MRF: peaks kept,
		             : srh_mvs_initial_estimate(ctx_, v, nb, count[v], &params_, 0, 0, nullptr)) != SRH_OK) { error_ = srh_last_error(); return; }
	}
	if (useMrf_) {                                                                                            // ... the MRF stage of all views side by side
		std::vector<int32_t> all(V);
		for (int v = 0; v < V; ++v) all[v] = v;
		if (srh_mvs_mrf_estimate_views(ctx_, all.data(), V, &mrfParams_, nullptr) != SRH_OK) { error_ = srh_last_error(); return; }
	}
	for (int v = 0; v < V; ++v)
		if (srh_view_depth_download(ctx_, v, computedDepths[v].data()) != SRH_OK) { error_ = srh_last_error(); return; }
	emit stageUpdate(tr("Constructing depth maps"));
	for (int v = 0; v < V; ++v) colorize(v);
	emit stageUpdate(tr("Cross-checking"));
	for (int v = 0; v < V; ++v) {
		if (isCancelled()) return;
		emit progressUpdate(step++);
		if (srh_mvs_cross_check(ctx_, slotIds.data(), V, v, &params_) != SRH_OK) { error_ = srh_last_error(); return; }
	}
	for (int v = 0; v < V; ++v)
		if (srh_view_depth_download(ctx_, v, computedDepths[v].data()) != SRH_OK) { error_ = srh_last_error(); return; }
	emit stageUpdate(tr("Constructing depth maps"));
	for (int v = 0; v < V; ++v) colorize(v);
}

QImage MultiViewStereo::depthMap(CameraPtr view) const {
	for (size_t v = 0; v < views_.size(); ++v)
		if (views_[v].ptr && views_[v].ptr == view) return results[v];
	return QImage();
}

QImage MultiViewStereo::depthMap(const QString &viewId) const {
	for (size_t v = 0; v < views_.size(); ++v)
		if (views_[v].id == viewId) return results[v];
	return QImage();
}
