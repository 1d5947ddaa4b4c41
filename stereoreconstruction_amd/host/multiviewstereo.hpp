// multiviewstereo.hpp -- MultiViewStereo with the reference's public interface
// (stereo/multiviewstereo.hpp:36-113) on libstereo_recon_hip.  initialize() takes the views'
// already-loaded, already-scaled images instead of (ProjectPtr, ImageSetPtr): the project /
// image-set data model is out of scope and only its outputs --
// camera parameters and pixels -- feed this path.
#pragma once

#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "camera.hpp"
#include "image.hpp"
#include "project.hpp"
#include "task.hpp"

struct PLYPoint { double p[3]; uint8_t rgb[3]; };
void outputPLYFile(const std::string &path, const std::vector<PLYPoint> &points);   // multiviewstereo.cpp:291-315

class MultiViewStereo : public Task {
public:
	typedef std::vector<double> DepthMap;

	explicit MultiViewStereo(int deviceOrdinal = 0);
	~MultiViewStereo();

	// images[i] is the default image of views[i] scaled by imageScale; its alpha channel is the
	// mask (alpha == 255 <=> WHITE, multiviewstereo.cpp:225-234).  Views with a null image are skipped.
	void initialize(const std::vector<CameraPtr> &views, const std::vector<Image> &images,
	                double minDepth, double maxDepth,
	                int numDepthLevels,
	                double crossCheckThreshold,
	                double imageScale = 1.0);

	// The reference's own signature (multiviewstereo.hpp:46-52): views are looked up in `imageSet` (default image
	// per camera; cameras without one, or whose file `load` cannot produce, are skipped, :218-220).  Decoding and
	// the two Qt scalings stay with the caller: load(file, imageScale, image, maskSource) returns the
	// smooth-scaled image and -- when the file has an alpha channel -- the fast-scaled copy the mask is taken from
	// (maskSource left null: all pixels WHITE, :225-237).
	typedef std::function<bool(const std::string &file, double imageScale, Image &image, Image &maskSource)> ImageLoader;
	void initialize(ProjectPtr project, ImageSetPtr imageSet, const std::vector<CameraPtr> &views,
	                double minDepth, double maxDepth,
	                int numDepthLevels,
	                double crossCheckThreshold,
	                double imageScale,
	                const ImageLoader &load);
	ImageSetPtr imageSet() const { return imageSet_; }   // the image set of the last initialize (multiviewstereo.hpp:62)

	std::string title() const { return "Multi-View Stereo"; }
	int numSteps() const;                              // 2 * views (multiviewstereo.cpp:319-321)

	Image depthMap(CameraPtr view) const;              // grayscale, NaN/INF/unknown white (:252-276)
	const DepthMap *depths(CameraPtr view) const;      // computedDepths[view]
	// The view's current depth map as coloured 3-D points (pixels with a WHITE mask and a finite depth; the point
	// is the cross-checks' construction, multiviewstereo.cpp:688-692) -- what outputPLYFile takes.
	std::vector<PLYPoint> pointCloud(CameraPtr view);
	// "percent of pixels have depth hypotheses": finite depths among the masked-in pixels (:402-421)
	double coverage(CameraPtr view) const;
	const std::vector<std::vector<int> > &neighbourViews() const { return neighbours; }

	srh_params &params() { return params_; }
	// The reference picks the MRF branch of computeInitialEstimate at build time (CONFIG+=mrf -> USE_MRF,
	// StereoReconstruction.pro:100-103); here it is a run-time switch, off by default like the reference's default build.
	void setUseMRF(bool on) { useMrf_ = on; }
	bool useMRF() const { return useMrf_; }
	srh_mrf_params &mrfParams() { return mrfParams_; }
	const std::string &lastError() const { return error_; }

protected:
	void runTask();

private:
	void colorize(size_t v);

	ProjectPtr project;
	ImageSetPtr imageSet_;
	std::vector<CameraPtr> views;
	std::vector<Image> images;
	std::vector<std::vector<uint8_t> > masks;
	std::vector<Image> results;
	std::vector<DepthMap> computedDepths;
	std::vector<std::vector<int> > neighbours;
	double minDepth, maxDepth, crossCheckThreshold, imageScale;
	int numDepthLevels;
	srh_params params_;
	srh_mrf_params mrfParams_;
	bool useMrf_ = false;
	srh_context *ctx_;
	std::string error_;
};
