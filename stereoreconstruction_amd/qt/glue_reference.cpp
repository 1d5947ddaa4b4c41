// glue_reference.cpp -- the ONE translation unit of the Qt binding that looks inside the reference's project model:
// the definitions of the glue functions stereo_qt.hpp declares (srq::cameraInfo, srq::defaultImageFile) and of the
// reference-signature member TwoViewStereo::epipolarCurve(ray, cameraOffset, depthPlaneNormal, mask, view).
//
// It includes project/camera.hpp, project/imageset.hpp, project/projectimage.hpp and util/ray.hpp, which need Eigen (and,
// through the project model, OpenCV): compile it where the reference builds -- add it to StereoReconstruction.pro's
// SOURCES next to stereo_qt.cpp, in place of stereo/twoviewstereo.cpp and stereo/multiviewstereo.cpp.  It is NOT built
// in this repository's image (no Eigen here); tests/qt_glue_test.hpp plays its part for the test driver.
#include "stereo_qt.hpp"

#include "project/camera.hpp"           // Camera: K(), R(), t(), lensDistortion(), plane(), refractiveIndex() (project/camera.hpp:56-82)
#include "project/imageset.hpp"         // ImageSet::defaultImageForCamera (project/imageset.hpp:83)
#include "project/projectimage.hpp"     // ProjectImage::file (project/projectimage.hpp:41)
#include "util/ray.hpp"                 // Ray3d::point
#include "util/vectorimage.hpp"         // RGBA

#include <cstring>

namespace srq {

CameraInfo cameraInfo(const Camera &cam) {
	CameraInfo info;
	info.id = cam.id();
	info.name = cam.name();
	// Eigen matrices are column-major: the C-ABI takes row-major K, R and the vectors t, distortion, plane normal
	double K[9], R[9], t[3], dist[5], normal[3];
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j) { K[i*3 + j] = cam.K()(i, j); R[i*3 + j] = cam.R()(i, j); }
		t[i] = cam.t()[i];
		normal[i] = cam.plane().normal()[i];
	}
	for (int i = 0; i < 5; ++i) dist[i] = cam.lensDistortion()[i];
	// srh_camera_from_krt = Camera::set + setLensDistortion + setPlane / setRefractiveIndex + updatePrincipleRay
	// (project/camera.cpp:225-240, 292-344): the derived members (Kinv, Rinv, C, principal ray, flags) are rebuilt there
	memset(&info.camera, 0, sizeof(info.camera));
	srh_camera_from_krt(K, R, t, dist, normal, cam.plane().distance(), cam.refractiveIndex(), &info.camera);
	return info;
}

QString defaultImageFile(const ImageSet &set, const CameraPtr &cam) {
	const ProjectImagePtr img = set.defaultImageForCamera(cam);
	return img ? img->file() : QString();
}

} // namespace srq

// stereo/twoviewstereo.hpp:66-70.  Callers build `ray` as unproject((x + 0.5)/scale, (y + 0.5)/scale) of the camera that is
// NOT `view` (twoviewstereo.cpp:275-283, 445-453): projecting a point of the ray back gives the pixel.
std::vector<Eigen::Vector3d> TwoViewStereo::epipolarCurve(const Ray3d &ray, const Eigen::Vector3d &, const Eigen::Vector3d &,
                                                          const VectorImage &, CameraPtr view) const
{
	std::vector<Eigen::Vector3d> curve;
	const bool fromLeft = view == rightCamera();
	const CameraPtr ref = fromLeft ? leftCamera() : rightCamera();
	if (!ref) return curve;
	Eigen::Vector3d p = ray.point(1.0);
	if (!ref->project(p)) return curve;
	const std::vector<std::array<double, 3> > pts = curveOfPixel(static_cast<int>(p[0]*scale()), static_cast<int>(p[1]*scale()), fromLeft);
	curve.reserve(pts.size());
	for (size_t k = 0; k < pts.size(); ++k) curve.push_back(Eigen::Vector3d(pts[k][0], pts[k][1], pts[k][2]));
	return curve;
}

// stereo/multiviewstereo.hpp:36-39 / multiviewstereo.cpp:291-315 with the reference's own types
void outputPLYFile(const std::string &path, const std::vector<PLYPoint> &points) {
	std::vector<double> xyz(points.size()*3);
	std::vector<int> rgb(points.size()*3);
	for (size_t i = 0; i < points.size(); ++i) {
		for (int k = 0; k < 3; ++k) xyz[i*3 + k] = points[i].first[k];
		// (the reference prints static_cast<int>(rgb.r) of a double: the int goes out as it is, also outside 0 .. 255)
		rgb[i*3 + 0] = static_cast<int>(points[i].second.r);
		rgb[i*3 + 1] = static_cast<int>(points[i].second.g);
		rgb[i*3 + 2] = static_cast<int>(points[i].second.b);
	}
	srq::writePLY(path, points.size(), xyz.data(), rgb.data());
}
