// stereo_qt.hpp -- the Qt binding of libstereo_recon_hip: TwoViewStereo and MultiViewStereo as the reference's GUI
// sees them (stereo/twoviewstereo.hpp:39-126, stereo/multiviewstereo.hpp:44-113), derived from the reference's own
// Task (gui/task.hpp:57-105; compiled from /root/reference where it lies, with its moc output), QImage in and out,
// the reference's signals, everything between "scaled images + cameras" and "depth maps" behind the C-ABI.
//
// What differs from the reference's signatures, and why: cameras arrive as srh_camera snapshots instead of
// CameraPtr, and the project as the loader's plain records (host/project.hpp) instead of ProjectPtr / ImageSetPtr --
// project/camera.hpp and project/project.hpp need Eigen and OpenCV, which this image does not have.  A maintainer
// with those headers replaces `const srh_camera &` by `CameraPtr` + the snapshot() helper of INTEGRATION.md.
#pragma once

#include <QtCore/QString>
#include <QtGui/QImage>

#include <array>
#include <string>
#include <vector>

#include "gui/task.hpp"                 // the reference's Task (QObject with run/cancel slots and progress signals)
#include "stereo_recon_hip.h"

namespace srq {

// VectorImage::fromQImage (util/vectorimage.cpp:48-64): the raw 32-bit scanline words as R,G,B,A bytes.  (A smooth-
// scaled ARGB32 image is ARGB32_Premultiplied; the reference reads it raw, and so does this.)  Images that are not
// 32 bits deep, which the reference would misread, are converted to ARGB32 first.
struct Raster { int w = 0, h = 0; std::vector<unsigned char> rgba; };
Raster rasterFromQImage(const QImage &img);
// mask.pixel(x,y) == WHITE for every pixel of a mask image (util/vectorimage.hpp:64-69): 1 where r=g=b=a=255
std::vector<unsigned char> whiteMask(const Raster &mask);
// the image + mask MultiViewStereo::initialize builds from a file (multiviewstereo.cpp:216-241): smooth-scaled image;
// mask = alpha == 255 on a FAST-scaled copy when the file has an alpha channel, all WHITE otherwise
bool ingestViewFile(const QString &file, double imageScale, Raster &image, std::vector<unsigned char> &mask);

} // namespace srq

class TwoViewStereo : public Task {
public:
	typedef std::vector<double> DepthMap;

	TwoViewStereo(const srh_camera &leftView, QImage left, QImage leftMask,
	              const srh_camera &rightView, QImage right, QImage rightMask,
	              double minDepth, double maxDepth, int numDepthLevels, double imageScale = 1.0,
	              int deviceOrdinal = 0);
	~TwoViewStereo();

	QString title() const { return "Two-View Stereo"; }
	int numSteps() const { return 8; }

	void computeDepthMaps();
	// TwoViewStereo::epipolarCurve (public member of the reference, stereo/twoviewstereo.hpp:66-70, .cpp:999-1054; the
	// GUI's curve preview is its user): the candidate pixels (tx, ty, 1) of reference pixel (x, y), in the order the
	// reference visits them, joint duplicates included.  The reference takes (ray, cameraOffset, depthPlaneNormal,
	// mask, view) -- Ray3d / Eigen types this image lacks -- and every caller builds them from a pixel the same way
	// (twoviewstereo.cpp:275-283, 445-453: ray = unproject((x + 0.5)/scale, (y + 0.5)/scale), offset and normal from the
	// same camera, mask and view of the other one), so the pixel and the direction are the arguments here.
	std::vector<std::array<double, 3> > epipolarCurve(int x, int y, bool fromLeft = true) const;
	QImage leftDepthMap() const { return resultLeft; }
	QImage rightDepthMap() const { return resultRight; }
	const DepthMap &leftDepths() const { return computedDepthLeft; }
	const DepthMap &rightDepths() const { return computedDepthRight; }
	srh_params &params() { return params_; }
	QString lastError() const { return error_; }

protected:
	void runTask() { computeDepthMaps(); }

private:
	QImage colorize(const DepthMap &d, int w, int h) const;
	srh_camera leftView, rightView;
	srq::Raster left, right;
	std::vector<unsigned char> leftMask, rightMask;
	double minDepth, maxDepth;
	int numDepthLevels;
	double imageScale;
	QImage resultLeft, resultRight;
	DepthMap computedDepthLeft, computedDepthRight;
	srh_params params_;
	srh_context *ctx_;
	mutable bool uploaded_ = false;                        // views resident on the device (epipolarCurve before computeDepthMaps)
	bool uploadViews() const;
	mutable QString error_;
};

class MultiViewStereo : public Task {
public:
	struct View { QString id, name; srh_camera camera; QString file; };   // a camera of the project and its image file

	explicit MultiViewStereo(int deviceOrdinal = 0);
	~MultiViewStereo();

	// MultiViewStereo::initialize (multiviewstereo.cpp:193-247): loads, scales and masks every view's image;
	// views without an existing image file are skipped
	void initialize(const std::vector<View> &views, double minDepth, double maxDepth, int numDepthLevels,
	                double crossCheckThreshold, double imageScale = 1.0);

	QString title() const { return "Multi-view Stereo"; }
	int numSteps() const { return 2*static_cast<int>(views_.size()); }

	QImage depthMap(const QString &viewId) const;                 // null image for an unknown view (:279-286)
	const std::vector<double> &depths(int viewIndex) const { return computedDepths[viewIndex]; }
	int numViews() const { return static_cast<int>(views_.size()); }
	const srq::Raster &image(int viewIndex) const { return images[viewIndex]; }
	const std::vector<unsigned char> &mask(int viewIndex) const { return masks[viewIndex]; }
	srh_params &params() { return params_; }
	// CONFIG+=mrf of the reference (USE_MRF, StereoReconstruction.pro:100-103) as a run-time switch; off by default
	void setUseMRF(bool on) { useMrf_ = on; }
	srh_mrf_params &mrfParams() { return mrfParams_; }
	QString lastError() const { return error_; }

protected:
	void runTask();

private:
	void colorize(int viewIndex);
	std::vector<View> views_;
	std::vector<srq::Raster> images;
	std::vector<std::vector<unsigned char> > masks;
	std::vector<QImage> results;
	std::vector<std::vector<double> > computedDepths;
	double minDepth, maxDepth, crossCheckThreshold, imageScale;
	int numDepthLevels;
	srh_params params_;
	srh_mrf_params mrfParams_;
	bool useMrf_ = false;
	srh_context *ctx_;
	QString error_;
};
