// twoviewstereo.cpp -- host side of TwoViewStereo above the C-ABI (reference:
// stereo/twoviewstereo.cpp:89-227).  No arithmetic of the matching path happens here.
#include "twoviewstereo.hpp"

#include <cmath>
#include <limits>

namespace {
	void onProgress(int step, const char *stage, void *user);
}

struct TwoViewHooks {
	TwoViewStereo *self;
	std::function<void(int)> *progress;
	std::function<void(const std::string &)> *stage;
};

namespace {
	void onProgress(int step, const char *stage, void *user) {
		TwoViewHooks *h = static_cast<TwoViewHooks *>(user);
		if (*h->progress) (*h->progress)(step);
		if (*h->stage) (*h->stage)(stage);
	}
}

TwoViewStereo::TwoViewStereo(CameraPtr leftView_, Image left_, Image leftMask_,
                             CameraPtr rightView_, Image right_, Image rightMask_,
                             double minDepth_, double maxDepth_, int numDepthLevels_,
                             double imageScale_, int deviceOrdinal)
	: leftView(leftView_), rightView(rightView_)
	, left(left_), right(right_)
	, minDepth(minDepth_), maxDepth(maxDepth_), numDepthLevels(numDepthLevels_), imageScale(imageScale_)
	, ctx_(nullptr)
{
	// masks: null => all WHITE (twoviewstereo.cpp:105-113)
	leftMask = whiteMask(leftMask_, left.w, left.h);
	rightMask = whiteMask(rightMask_, right.w, right.h);
	resultLeft = Image(left.w, left.h);
	resultRight = Image(right.w, right.h);
	const double NaN = std::numeric_limits<double>::quiet_NaN();
	computedDepthLeft.assign(static_cast<size_t>(left.w)*left.h, NaN);
	computedDepthRight.assign(static_cast<size_t>(left.w)*left.h, NaN);   // sized from the LEFT image, as :119
	srh_params_twoview_defaults(&params_);
	params_.min_depth = minDepth; params_.max_depth = maxDepth;
	params_.num_depth_levels = numDepthLevels; params_.image_scale = imageScale;
	if (srh_create(deviceOrdinal, &ctx_) != SRH_OK) { error_ = srh_last_error(); ctx_ = nullptr; }
}

TwoViewStereo::~TwoViewStereo() { if (ctx_) srh_destroy(ctx_); }

void TwoViewStereo::colorFromDepth(double depth, uint8_t rgb[3]) const {
	// twoviewstereo.cpp:128-146 (QColor::fromHsvF(2t/3, 1, 1) restated; outside the numeric contract)
	rgb[0] = rgb[1] = rgb[2] = 0;
	if (std::isnan(depth) || std::isinf(depth)) return;
	const double t = (depth - minDepth) / (maxDepth - minDepth);
	if (t < 1e-5) return;
	if (t > 1.1) { rgb[0] = rgb[1] = rgb[2] = 255; return; }
	double h6 = (2.0*t/3.0)*6.0;
	h6 -= 6.0*std::floor(h6/6.0);
	const int sector = static_cast<int>(h6);
	const double f = h6 - sector, q = 1.0 - f;
	double r = 0, g = 0, b = 0;
	switch (sector) {
	case 0: r = 1; g = f; b = 0; break;
	case 1: r = q; g = 1; b = 0; break;
	case 2: r = 0; g = 1; b = f; break;
	case 3: r = 0; g = q; b = 1; break;
	case 4: r = f; g = 0; b = 1; break;
	default: r = 1; g = 0; b = q; break;
	}
	rgb[0] = static_cast<uint8_t>(std::lround(r*255)); rgb[1] = static_cast<uint8_t>(std::lround(g*255));
	rgb[2] = static_cast<uint8_t>(std::lround(b*255));
}

void TwoViewStereo::colorize(const DepthMap &d, Image &out) const {
	for (int y = 0; y < out.h; ++y)
		for (int x = 0; x < out.w; ++x) {
			uint8_t *p = out.pixel(x, y);
			colorFromDepth(d[static_cast<size_t>(y)*out.w + x], p);
			p[3] = 255;
		}
}

bool TwoViewStereo::uploadViews() {
	params_.min_depth = minDepth; params_.max_depth = maxDepth;
	params_.num_depth_levels = numDepthLevels; params_.image_scale = imageScale;
	const srh_camera lc = leftView->snapshot(), rc = rightView->snapshot();   // snapshot: the GUI may mutate cameras
	if (srh_view_upload(ctx_, 0, left.w, left.h, left.rgba.data(), leftMask.data(), &lc) != SRH_OK ||
	    srh_view_upload(ctx_, 1, right.w, right.h, right.rgba.data(), rightMask.data(), &rc) != SRH_OK) {
		error_ = srh_last_error();
		return false;
	}
	return true;
}

std::vector<std::pair<int, int> > TwoViewStereo::epipolarCurve(int x, int y, bool fromLeft) {
	std::vector<std::pair<int, int> > curve;
	if (!ctx_ || !leftView || !rightView || left.isNull() || right.isNull() || !uploadViews()) return curve;
	const int32_t xy[2] = { x, y };
	int32_t n = 0;
	std::vector<int32_t> pts(2*64);
	for (int attempt = 0; attempt < 2; ++attempt) {
		const int cap = static_cast<int>(pts.size()/2);
		if (srh_epipolar_curves(ctx_, fromLeft ? 0 : 1, fromLeft ? 1 : 0, &params_, 0, 1, xy, pts.data(), cap, &n) != SRH_OK) {
			error_ = srh_last_error();
			return curve;
		}
		if (n <= cap) break;
		pts.resize(2*static_cast<size_t>(n));
	}
	for (int i = 0; i < n; ++i) curve.push_back(std::make_pair(pts[2*i], pts[2*i + 1]));
	return curve;
}

void TwoViewStereo::computeCostVolumes(CameraPtr leftView_, CameraPtr rightView_) {
	if (!ctx_ || !leftView_ || !rightView_) return;
	leftView = leftView_; rightView = rightView_;
	if (!uploadViews()) return;
	if (srh_twoview_wta(ctx_, 0, 1, &params_, 0, 0) != SRH_OK || srh_twoview_wta(ctx_, 1, 0, &params_, 0, 0) != SRH_OK ||
	    srh_view_depth_download(ctx_, 0, computedDepthLeft.data()) != SRH_OK ||
	    srh_view_depth_download(ctx_, 1, computedDepthRight.data()) != SRH_OK)
		error_ = srh_last_error();
}

void TwoViewStereo::crossCheck(CameraPtr leftView_, CameraPtr rightView_) {
	if (!ctx_ || !leftView_ || !rightView_) return;
	leftView = leftView_; rightView = rightView_;
	if (!uploadViews()) return;
	if (srh_view_depth_upload(ctx_, 0, computedDepthLeft.data()) != SRH_OK ||
	    srh_view_depth_upload(ctx_, 1, computedDepthRight.data()) != SRH_OK ||
	    srh_twoview_cross_check(ctx_, 0, 1, &params_) != SRH_OK ||
	    srh_view_depth_download(ctx_, 0, computedDepthLeft.data()) != SRH_OK ||
	    srh_view_depth_download(ctx_, 1, computedDepthRight.data()) != SRH_OK)
		error_ = srh_last_error();
}

double TwoViewStereo::depthFromLabel(int label) const {
	double t = static_cast<double>(label) / (numDepthLevels - 1);
	t /= (5.0 - 4.0*t);
	return minDepth*(1.0 - t) + maxDepth*t;
}

void TwoViewStereo::computeDepthMaps() {
	// twoviewstereo.cpp:150-227: cost volumes (steps 1,3), cross-check (5), colourise, finished (8)
	if (!ctx_) { if (error_.empty()) error_ = "no device context"; return; }
	if (!leftView || !rightView || left.isNull() || right.isNull()) { error_ = "missing view"; return; }
	if (!uploadViews()) return;
	TwoViewHooks hooks = { this, &progressUpdate, &stageUpdate };
	srh_set_hooks(ctx_, cancelFlag(), onProgress, &hooks);
	const int rc_ = srh_twoview_compute(ctx_, 0, 1, &params_, computedDepthLeft.data(), computedDepthRight.data());
	srh_set_hooks(ctx_, nullptr, nullptr, nullptr);
	if (rc_ == SRH_E_CANCELLED) return;              // reference: silent return on cancel
	if (rc_ != SRH_OK) { error_ = srh_last_error(); return; }
	colorize(computedDepthLeft, resultLeft);
	colorize(computedDepthRight, resultRight);
}
