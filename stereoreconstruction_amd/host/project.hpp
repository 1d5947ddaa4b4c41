// project.hpp -- the part of the reference's project model the stereo path is fed from
// (project/project.cpp:74-227, project/imageset.cpp, project/projectimage.hpp), without Qt:
// cameras (projection matrix -> Camera::setP, lens distortion, refractive interface) and image sets
// (which file belongs to which camera) read from a project XML file.  Features, responses and
// correspondences of the file format are not on the stereo path and are skipped.
#pragma once

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "camera.hpp"
#include "image.hpp"

class ProjectImage {
public:
	explicit ProjectImage(const std::string &file) : file_(file), exposure_(-1.0) { }
	const std::string &file() const { return file_; }
	double exposure() const { return exposure_; }
	void setExposure(double e) { exposure_ = e; }
	CameraPtr camera() const { return camera_; }
	void setCamera(CameraPtr c) { camera_ = c; }
private:
	std::string file_;
	double exposure_;
	CameraPtr camera_;
};
typedef std::shared_ptr<ProjectImage> ProjectImagePtr;

class ImageSet {
public:
	explicit ImageSet(const std::string &id, const std::string &name = std::string())
		: id_(id), name_(name.empty() ? "<no name>" : name) { }
	const std::string &id() const { return id_; }
	const std::string &name() const { return name_; }
	void setName(const std::string &n) { name_ = n; }
	const std::string &root() const { return root_; }
	void setRoot(const std::string &r) { root_ = r; }
	const std::vector<ProjectImagePtr> &images() const { return images_; }
	// the first image added for a camera is its default (imageset.cpp:56-71)
	void addImageForCamera(CameraPtr cam, ProjectImagePtr image);
	ProjectImagePtr defaultImageForCamera(CameraPtr cam) const;   // null when the camera has none (:75-79)
private:
	std::string id_, name_, root_;
	std::vector<ProjectImagePtr> images_;
	std::map<CameraPtr, ProjectImagePtr> defaults_;
};
typedef std::shared_ptr<ImageSet> ImageSetPtr;

class Project {
public:
	// throws std::runtime_error("Failed to open file" / "Failed to set XML content" / "Failed to validate"),
	// like the reference (project.cpp:91,97,101); an empty path gives an empty project (:78-79)
	explicit Project(const std::string &projectPath = std::string());

	const std::string &projectPath() const { return projectPath_; }
	const std::map<std::string, CameraPtr> &cameras() const { return cameras_; }
	const std::map<std::string, ImageSetPtr> &imageSets() const { return imageSets_; }
	CameraPtr camera(const std::string &id) const;
	ImageSetPtr imageSet(const std::string &id) const;

private:
	std::string projectPath_;
	std::map<std::string, CameraPtr> cameras_;
	std::map<std::string, ImageSetPtr> imageSets_;
};
typedef std::shared_ptr<Project> ProjectPtr;

// MultiViewStereo::initialize's mask rule (multiviewstereo.cpp:225-234): a pixel takes part only if the
// (fast-scaled) image is fully opaque there.  `rgba` is 8-bit R,G,B,A; returns 1 = WHITE, 0 = BLACK.
std::vector<uint8_t> maskFromAlpha(const Image &scaledForMask);
