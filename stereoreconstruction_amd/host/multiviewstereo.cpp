// multiviewstereo.cpp -- host side of MultiViewStereo above the C-ABI (reference:
// stereo/multiviewstereo.cpp:193-475).  Sequencing and bookkeeping only.
#include "multiviewstereo.hpp"

#include <cmath>
#include <fstream>
#include <limits>

void outputPLYFile(const std::string &path, const std::vector<PLYPoint> &points) {
	std::ofstream lout(path.c_str());
	lout << "ply\n" << "format ascii 1.0\n" << "element vertex " << points.size() << "\n"
	     << "property float x\n" << "property float y\n" << "property float z\n"
	     << "property uchar diffuse_red\n" << "property uchar diffuse_green\n" << "property uchar diffuse_blue\n"
	     << "end_header\n";
	for (size_t i = 0; i < points.size(); ++i) {
		const PLYPoint &pp = points[i];
		lout << pp.p[0] << ' ' << pp.p[1] << ' ' << pp.p[2] << ' '
		     << static_cast<int>(pp.rgb[0]) << ' ' << static_cast<int>(pp.rgb[1]) << ' ' << static_cast<int>(pp.rgb[2]) << '\n';
	}
}

MultiViewStereo::MultiViewStereo(int deviceOrdinal)
	: minDepth(0), maxDepth(0), crossCheckThreshold(0), imageScale(1), numDepthLevels(0), ctx_(nullptr)
{
	srh_params_mvs_defaults(&params_);
	srh_mrf_params_defaults(&mrfParams_);
	if (srh_create(deviceOrdinal, &ctx_) != SRH_OK) { error_ = srh_last_error(); ctx_ = nullptr; }
}

MultiViewStereo::~MultiViewStereo() { if (ctx_) srh_destroy(ctx_); }

void MultiViewStereo::initialize(const std::vector<CameraPtr> &views_, const std::vector<Image> &images_,
                                 double minDepth_, double maxDepth_, int numDepthLevels_,
                                 double crossCheckThreshold_, double imageScale_)
{
	// multiviewstereo.cpp:193-247
	minDepth = minDepth_; maxDepth = maxDepth_; numDepthLevels = numDepthLevels_;
	crossCheckThreshold = crossCheckThreshold_; imageScale = imageScale_;
	views.clear(); images.clear(); masks.clear(); results.clear(); computedDepths.clear();
	const double NaN = std::numeric_limits<double>::quiet_NaN();
	for (size_t i = 0; i < views_.size() && i < images_.size(); ++i) {
		if (!views_[i] || images_[i].isNull()) continue;
		const Image &im = images_[i];
		images.push_back(im);
		std::vector<uint8_t> m(static_cast<size_t>(im.w)*im.h);
		for (size_t k = 0; k < m.size(); ++k) m[k] = im.rgba[k*4 + 3] == 255 ? 1 : 0;   // not fully opaque => ignored
		masks.push_back(m);
		results.push_back(Image(im.w, im.h));
		computedDepths.push_back(DepthMap(static_cast<size_t>(im.w)*im.h, NaN));
		views.push_back(views_[i]);
	}
	neighbours.assign(views.size(), std::vector<int>());
}

void MultiViewStereo::initialize(ProjectPtr project_, ImageSetPtr imageSet__, const std::vector<CameraPtr> &views_,
                                 double minDepth_, double maxDepth_, int numDepthLevels_,
                                 double crossCheckThreshold_, double imageScale_, const ImageLoader &load)
{
	// multiviewstereo.cpp:193-247 with the file access handed to `load`
	std::vector<CameraPtr> kept;
	std::vector<Image> imgs, maskSrc;
	for (size_t i = 0; i < views_.size(); ++i) if (views_[i] && imageSet__) {
		const ProjectImagePtr pi = imageSet__->defaultImageForCamera(views_[i]);
		Image im, ms;
		if (!pi || !load || !load(pi->file(), imageScale_, im, ms) || im.isNull()) continue;
		kept.push_back(views_[i]); imgs.push_back(im); maskSrc.push_back(ms);
	}
	initialize(kept, imgs, minDepth_, maxDepth_, numDepthLevels_, crossCheckThreshold_, imageScale_);
	project = project_;
	imageSet_ = imageSet__;
	for (size_t v = 0; v < views.size(); ++v) {
		// mask: alpha == 255 on the fast-scaled copy; no alpha channel => WHITE everywhere (:225-237)
		if (maskSrc[v].isNull()) masks[v].assign(masks[v].size(), 1);
		else if (maskSrc[v].w == images[v].w && maskSrc[v].h == images[v].h) masks[v] = maskFromAlpha(maskSrc[v]);
	}
}

int MultiViewStereo::numSteps() const { return 2*static_cast<int>(views.size()); }

void MultiViewStereo::colorize(size_t v) {
	// runTask "Constructing depth maps" (multiviewstereo.cpp:382-395) + colorFromDepth (:257-276)
	Image &out = results[v];
	for (int y = 0; y < out.h; ++y)
		for (int x = 0; x < out.w; ++x) {
			const size_t i = static_cast<size_t>(y)*out.w + x;
			uint8_t g = 255;
			const double d = computedDepths[v][i];
			if (masks[v][i] == 1 && !std::isnan(d) && !std::isinf(d) && !(d + 1e-5 < minDepth)) {
				double t = (d - minDepth) / (maxDepth - minDepth);
				t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
				g = static_cast<uint8_t>(255*t);
			}
			uint8_t *p = out.pixel(x, y);
			p[0] = p[1] = p[2] = g; p[3] = 255;
		}
}

void MultiViewStereo::runTask() {
	// multiviewstereo.cpp:325-475
	if (!ctx_ || views.empty()) return;               // reference: silent return (":326-327 TODO error")
	const int nv = static_cast<int>(views.size());
	if (nv > SRH_MAX_VIEWS) { error_ = "too many views"; return; }
	params_.min_depth = minDepth; params_.max_depth = maxDepth; params_.num_depth_levels = numDepthLevels;
	params_.image_scale = imageScale; params_.cross_check_threshold = crossCheckThreshold;
	int current_step = 0;

	std::vector<srh_camera> cams(nv);
	for (int v = 0; v < nv; ++v) cams[v] = views[v]->snapshot();
	std::vector<int32_t> neigh(static_cast<size_t>(nv)*params_.num_neighbours, -1), count(nv, 0);
	if (srh_mvs_neighbours(nv, cams.data(), &params_, neigh.data(), count.data()) != SRH_OK) { error_ = srh_last_error(); return; }
	for (int v = 0; v < nv; ++v) {
		neighbours[v].assign(neigh.begin() + static_cast<size_t>(v)*params_.num_neighbours,
		                     neigh.begin() + static_cast<size_t>(v)*params_.num_neighbours + count[v]);
		if (srh_view_upload(ctx_, v, images[v].w, images[v].h, images[v].rgba.data(), masks[v].data(), &cams[v]) != SRH_OK) {
			error_ = srh_last_error(); return;
		}
	}
	srh_set_hooks(ctx_, cancelFlag(), nullptr, nullptr);

	for (int v = 0; v < nv; ++v) {                     // initial stereo estimate, :365-376
		emitProgress(current_step++);
		emitStage("Computing cost volume for camera " + views[v]->name());
		const int32_t *nb = &neigh[static_cast<size_t>(v)*params_.num_neighbours];
		// #ifdef USE_MRF (:610-652): the peaks are kept and the MRF stage of all views runs afterwards, side by side
		// (the views' estimates do not depend on each other: the same results as one view after the other)
		const int rc = useMrf_ ? srh_mvs_initial_estimate_peaks(ctx_, v, nb, count[v], &params_)
		                       : srh_mvs_initial_estimate(ctx_, v, nb, count[v], &params_, 0, 0, nullptr);
		if (rc == SRH_E_CANCELLED || isCancelled()) { srh_set_hooks(ctx_, nullptr, nullptr, nullptr); return; }
		if (rc != SRH_OK) { error_ = srh_last_error(); srh_set_hooks(ctx_, nullptr, nullptr, nullptr); return; }
	}
	if (useMrf_) {
		std::vector<int32_t> all(nv);
		for (int v = 0; v < nv; ++v) all[v] = v;
		const int rc = srh_mvs_mrf_estimate_views(ctx_, all.data(), nv, &mrfParams_, nullptr);
		if (rc == SRH_E_CANCELLED || isCancelled()) { srh_set_hooks(ctx_, nullptr, nullptr, nullptr); return; }
		if (rc != SRH_OK) { error_ = srh_last_error(); srh_set_hooks(ctx_, nullptr, nullptr, nullptr); return; }
	}
	emitStage("Constructing depth maps");
	for (int v = 0; v < nv; ++v) {
		srh_view_depth_download(ctx_, v, computedDepths[v].data());
		colorize(v);
	}

	emitStage("Cross-checking");                       // :427-431, views in order
	std::vector<int32_t> slots(nv);
	for (int v = 0; v < nv; ++v) slots[v] = v;
	for (int v = 0; v < nv; ++v) {
		emitProgress(current_step++);
		if (srh_mvs_cross_check(ctx_, slots.data(), nv, v, &params_) != SRH_OK) { error_ = srh_last_error(); break; }
		if (isCancelled()) break;
	}
	emitStage("Constructing depth maps");
	for (int v = 0; v < nv; ++v) {
		srh_view_depth_download(ctx_, v, computedDepths[v].data());
		colorize(v);
	}
	srh_set_hooks(ctx_, nullptr, nullptr, nullptr);
}

Image MultiViewStereo::depthMap(CameraPtr view) const {
	for (size_t v = 0; v < views.size(); ++v)
		if (views[v] == view) return results[v];
	return Image();                                    // unknown view => null image (:285)
}

const MultiViewStereo::DepthMap *MultiViewStereo::depths(CameraPtr view) const {
	for (size_t v = 0; v < views.size(); ++v)
		if (views[v] == view) return &computedDepths[v];
	return nullptr;
}

std::vector<PLYPoint> MultiViewStereo::pointCloud(CameraPtr view) {
	std::vector<PLYPoint> pts;
	for (size_t v = 0; v < views.size(); ++v) if (views[v] == view && ctx_) {
		const size_t n = static_cast<size_t>(images[v].w)*images[v].h;
		std::vector<double> xyz(3*n);
		std::vector<uint8_t> rgb(3*n), valid(n);
		params_.image_scale = imageScale;
		if (srh_view_depth_upload(ctx_, static_cast<int>(v), computedDepths[v].data()) != SRH_OK ||
		    srh_view_point_cloud(ctx_, static_cast<int>(v), &params_, xyz.data(), rgb.data(), valid.data(), nullptr, nullptr, nullptr) != SRH_OK) {
			error_ = srh_last_error();
			return pts;
		}
		for (size_t i = 0; i < n; ++i) if (valid[i]) {
			PLYPoint q;
			for (int k = 0; k < 3; ++k) { q.p[k] = xyz[3*i + k]; q.rgb[k] = rgb[3*i + k]; }
			pts.push_back(q);
		}
	}
	return pts;
}

double MultiViewStereo::coverage(CameraPtr view) const {
	for (size_t v = 0; v < views.size(); ++v) if (views[v] == view) {
		long total = 0, have = 0;
		for (size_t i = 0; i < masks[v].size(); ++i) if (masks[v][i] == 1) { ++total; if (std::isfinite(computedDepths[v][i])) ++have; }
		return total ? (100.0*have)/total : 0.0;
	}
	return 0.0;
}
