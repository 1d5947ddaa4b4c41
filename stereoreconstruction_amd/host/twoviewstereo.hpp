// twoviewstereo.hpp -- TwoViewStereo with the reference's public interface
// (stereo/twoviewstereo.hpp:39-126), running on libstereo_recon_hip (MI355X).
// Differences forced by dropping Qt: QImage -> Image (already scaled by imageScale),
// QString -> std::string, signals -> std::function members of Task.
#pragma once

#include <string>
#include <utility>
#include <vector>

#include "camera.hpp"
#include "image.hpp"
#include "task.hpp"

class TwoViewStereo : public Task {
public:
	typedef std::vector<double> DepthMap;

	TwoViewStereo(CameraPtr leftView, Image left, Image leftMask,
	              CameraPtr rightView, Image right, Image rightMask,
	              double minDepth, double maxDepth,
	              int numDepthLevels,
	              double imageScale = 1.0,
	              int deviceOrdinal = 0);
	~TwoViewStereo();

	std::string title() const { return "Two-View Stereo"; }
	int numSteps() const { return 8; }

	void computeDepthMaps();

	// Candidate pixels, in visiting order, of pixel (x,y) of the left (fromLeft) or right view in the
	// other view.  The reference's public epipolarCurve (twoviewstereo.hpp:66-70) takes the unprojected
	// ray, camera offset, plane normal, mask and view; all of them follow from the pixel and the
	// direction, which is what StereoWidget has in hand (stereowidget.cpp:621-672).
	std::vector<std::pair<int, int> > epipolarCurve(int x, int y, bool fromLeft = true);

	Image leftDepthMap() const { return resultLeft; }
	Image rightDepthMap() const { return resultRight; }

	// raw results (computedDepthLeft / computedDepthRight of the reference, twoviewstereo.hpp:113-114)
	const DepthMap &leftDepths() const { return computedDepthLeft; }
	const DepthMap &rightDepths() const { return computedDepthRight; }

	srh_params &params() { return params_; }          // every hard-coded constant, with the reference's defaults
	const std::string &lastError() const { return error_; }

protected:
	void runTask() { computeDepthMaps(); }

	// the reference's protected stages (twoviewstereo.hpp:72-84), for subclasses that re-sequence them:
	// WTA of both directions into computedDepthLeft/Right (twoviewstereo.cpp:233-501, non-MRF body) ...
	void computeCostVolumes(CameraPtr leftView, CameraPtr rightView);
	// ... and the mutual consistency filter on them (:596-672)
	void crossCheck(CameraPtr leftView, CameraPtr rightView);
	// label -> depth, non-uniform (:981-985)
	double depthFromLabel(int label) const;
	// cost_sad, filterInvalidPixels and weightedMedian are unreachable in the reference (never called /
	// call site under `#if 0`, twoviewstereo.cpp:200) and have no counterpart here.

private:
	bool uploadViews();
	void colorize(const DepthMap &d, Image &out) const;
	void colorFromDepth(double depth, uint8_t rgb[3]) const;

	CameraPtr leftView, rightView;
	Image left, right;
	std::vector<uint8_t> leftMask, rightMask;
	double minDepth, maxDepth;
	int numDepthLevels;
	double imageScale;
	Image resultLeft, resultRight;
	DepthMap computedDepthLeft, computedDepthRight;
	srh_params params_;
	srh_context *ctx_;
	std::string error_;
};
