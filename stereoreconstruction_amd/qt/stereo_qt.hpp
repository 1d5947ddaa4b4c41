// stereo_qt.hpp -- the Qt binding of libstereo_recon_hip: TwoViewStereo and MultiViewStereo with the reference's OWN
// public signatures (stereo/twoviewstereo.hpp:39-70, stereo/multiviewstereo.hpp:44-63), derived from the reference's own
// Task (gui/task.hpp:57-105; compiled from /root/reference where it lies, with its moc output), QImage in and out, the
// reference's signals, everything between "scaled images + cameras" and "depth maps" behind the C-ABI.
//
// Camera / Project / ImageSet / Ray3d / VectorImage / Eigen::Vector3d appear here as FORWARD DECLARATIONS only, exactly
// as stereo/*.hpp forward-declare the first three (FORWARD_DECLARE, util/precompiled.hpp:60-63): the binding never looks
// inside them.  What it needs from them -- a POD snapshot of a camera, its id and name, the image file an image set holds
// for a camera, the pixel a curve query is about -- comes through the glue declared below (srq::cameraInfo,
// srq::defaultImageFile, and the definition of the five-argument TwoViewStereo::epipolarCurve), defined in ONE translation
// unit of the host application:
//   qt/glue_reference.cpp   for the reference itself (includes project/camera.hpp, project/imageset.hpp, util/ray.hpp:
//                           needs Eigen and OpenCV, which this image lacks, so it is compiled where they exist;
//                           INTEGRATION.md shows it in full),
//   tests/qt_glue_test.hpp  the test driver's own small Camera / ImageSet / ... classes.
// gui/widgets/stereowidget.cpp's call sites (:160, 263, 274, 321-322, 964-966, 990-994) compile against this header
// unchanged; tests/qt_adapter_test.cpp holds them as a type-check.
#pragma once

#include <QtCore/QString>
#include <QtGui/QImage>

#include <array>
#include <string>
#include <utility>
#include <vector>

#include <memory>

#include "gui/task.hpp"                 // the reference's Task (QObject with run/cancel slots and progress signals)
#include "stereo_recon_hip.h"

FORWARD_DECLARE(Project);               // class Project; typedef std::shared_ptr<Project> ProjectPtr; ... (the reference's macro)
FORWARD_DECLARE(Camera);
FORWARD_DECLARE(ImageSet);
class Ray3d;
class VectorImage;
struct RGBA;                            // util/vectorimage.hpp:28
namespace Eigen {
template <typename Scalar, int Rows, int Cols, int Options, int MaxRows, int MaxCols> class Matrix;   // (Eigen's own forward declaration, minus its defaults)
typedef Matrix<double, 3, 1, 0, 3, 1> Vector3d;
}

// stereo/multiviewstereo.hpp:36-39, verbatim but for Ray3d::Point spelt as what it is (util/ray.hpp:32: Eigen::Vector3d).
// Both are incomplete types here; outputPLYFile is DEFINED in the glue translation unit, over srq::writePLY below.
typedef std::pair<Eigen::Vector3d, RGBA> PLYPoint;
//! Output a set of points to a PLY file
void outputPLYFile(const std::string &path, const std::vector<PLYPoint> &points);

namespace srq {

// the text outputPLYFile writes (multiviewstereo.cpp:291-315): ASCII header, then "x y z r g b" per point through
// operator<< of an ofstream; xyz = 3 doubles per point, rgb = 3 bytes per point (the reference prints static_cast<int>(rgb.r))
void writePLY(const std::string &path, size_t npoints, const double *xyz, const unsigned char *rgb);
// the same with the colours as the ints the reference prints: its RGBA holds doubles and static_cast<int>(rgb.r) goes out
// unchanged -- a component outside 0 .. 255 (or negative) must not wrap modulo 256 on the way (outputPLYFile uses this one)
void writePLY(const std::string &path, size_t npoints, const double *xyz, const int *rgb);

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

// ---- the glue: DECLARED here, DEFINED by the host application (see the header comment) ----
struct CameraInfo { srh_camera camera; QString id, name; };
// K, R, t, distortion, refractive interface of a project camera as the POD the C-ABI takes (a copy, not a live pointer)
CameraInfo cameraInfo(const Camera &cam);
// imageSet->defaultImageForCamera(cam)->file(), empty when the set has no image for the camera (multiviewstereo.cpp:214)
QString defaultImageFile(const ImageSet &set, const CameraPtr &cam);

} // namespace srq

class TwoViewStereo : public Task {
public:
	typedef std::vector<double> DepthMap;

	// stereo/twoviewstereo.hpp:44-47, verbatim.  The GPU is chosen with setDevice() (default 0).
	TwoViewStereo(CameraPtr leftView, QImage left, QImage leftMask,
	              CameraPtr rightView, QImage right, QImage rightMask,
	              double minDepth, double maxDepth, int numDepthLevels,
	              double imageScale = 1.0);
	~TwoViewStereo();
	static void setDevice(int ordinal);

	QString title() const { return "Two-View Stereo"; }
	int numSteps() const { return 8; }

	void computeDepthMaps();
	QImage leftDepthMap() const { return resultLeft; }
	QImage rightDepthMap() const { return resultRight; }
	// stereo/twoviewstereo.hpp:66-70, verbatim (twoviewstereo.cpp:999-1054): the candidate pixels (tx, ty, 1) of the
	// reference pixel whose `ray` this is, in the order the reference visits them, joint duplicates included.  Every
	// caller builds (ray, cameraOffset, depthPlaneNormal, mask, view) from a pixel of one view the same way
	// (twoviewstereo.cpp:275-283, 445-453: ray = unproject((x + 0.5)/scale, (y + 0.5)/scale), offset and normal of the same
	// camera, mask and `view` of the other one), so the glue translation unit -- which knows Ray3d and Eigen -- recovers
	// the pixel and the direction and calls curveOfPixel(); that is where this member is DEFINED.
	std::vector<Eigen::Vector3d> epipolarCurve(const Ray3d &ray,
	                                           const Eigen::Vector3d &cameraOffset,
	                                           const Eigen::Vector3d &depthPlaneNormal,
	                                           const VectorImage &mask,
	                                           CameraPtr view) const;

	// ---- beyond the reference's interface ----
	std::vector<std::array<double, 3> > curveOfPixel(int x, int y, bool fromLeft = true) const;
	CameraPtr leftCamera() const { return leftView; }
	CameraPtr rightCamera() const { return rightView; }
	double scale() const { return imageScale; }
	const DepthMap &leftDepths() const { return computedDepthLeft; }
	const DepthMap &rightDepths() const { return computedDepthRight; }
	srh_params &params() { return params_; }
	QString lastError() const { return error_; }

public: // Task implementation continued: public in the reference as well (stereo/twoviewstereo.hpp:50-52)
	void runTask() { computeDepthMaps(); }

private:
	QImage colorize(const DepthMap &d, int w, int h) const;
	CameraPtr leftView, rightView;
	srh_camera leftCam, rightCam;                          // snapshots taken by the constructor (srq::cameraInfo)
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
	// a camera of the project, its snapshot and its image file: what initialize() resolves every CameraPtr into
	struct View { QString id, name; srh_camera camera; QString file; CameraPtr ptr; };

	MultiViewStereo();
	~MultiViewStereo();
	static void setDevice(int ordinal);

	// stereo/multiviewstereo.hpp:46-52, verbatim (multiviewstereo.cpp:193-247): loads, scales and masks the image the
	// image set holds for every view; views without an existing image file are skipped
	void initialize(ProjectPtr project,
	                ImageSetPtr imageSet,
	                const std::vector<CameraPtr> &views,
	                double minDepth, double maxDepth,
	                int numDepthLevels,
	                double crossCheckThreshold,
	                double imageScale = 1.0);
	// the same from resolved records (hosts without the reference's project model)
	void initialize(const std::vector<View> &views, double minDepth, double maxDepth, int numDepthLevels,
	                double crossCheckThreshold, double imageScale = 1.0);

	QString title() const { return "Multi-view Stereo"; }
	int numSteps() const { return 2*static_cast<int>(views_.size()); }

	QImage depthMap(CameraPtr view) const;                        // stereo/multiviewstereo.hpp:60; null image for an unknown view (:279-286)
	ImageSetPtr imageSet() const { return imageSet_; }            // stereo/multiviewstereo.hpp:63
	QImage depthMap(const QString &viewId) const;                 // (by id: record-based hosts)
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
	ProjectPtr project;
	ImageSetPtr imageSet_;
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
