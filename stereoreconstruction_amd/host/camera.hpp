// camera.hpp -- the part of the reference's Camera (project/camera.hpp:38-188) the stereo path
// reads, without Qt/Eigen: calibration in, an srh_camera snapshot out.  All matrix math is done
// by the C-ABI helper srh_camera_from_krt (Camera::set path, project/camera.cpp:225-240).
#pragma once

#include <array>
#include <memory>
#include <string>

#include "stereo_recon_hip.h"

typedef std::array<double, 5> LensDistortions;      // k1,k2,p1,p2,k3 (project/camera.hpp:33)

class Camera {
public:
	explicit Camera(const std::string &id, const std::string &name = std::string())
		: id_(id), name_(name.empty() ? "<no name>" : name)
	{
		const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
		for (int i = 0; i < 9; ++i) { K_[i] = I[i]; R_[i] = I[i]; }
		for (int i = 0; i < 3; ++i) { t_[i] = z[i]; normal_[i] = i == 2; }
		dist_.fill(0.0);
		refresh();
	}

	const std::string &id() const { return id_; }
	const std::string &name() const { return name_; }

	// Camera::set(K, R, t): row-major 3x3 K and R, translation t
	void set(const double K[9], const double R[9], const double t[3]) {
		for (int i = 0; i < 9; ++i) { K_[i] = K[i]; R_[i] = R[i]; }
		for (int i = 0; i < 3; ++i) t_[i] = t[i];
		fromP_ = false;
		refresh();
	}
	// Camera::setP (project/camera.cpp:251-288): the path project files take; K, R, t come out of the
	// RQ factorisation done by srh_camera_from_p
	void setP(const double P[12]) {
		for (int i = 0; i < 12; ++i) P_[i] = P[i];
		fromP_ = true;
		refresh();
		for (int i = 0; i < 9; ++i) { K_[i] = snap_.K[i]; R_[i] = snap_.R[i]; }
		for (int i = 0; i < 3; ++i) t_[i] = snap_.t[i];
	}
	void setName(const std::string &name) { name_ = name; }
	void setLensDistortion(const LensDistortions &d) { dist_ = d; refresh(); }
	// Plane3d(normal, distance) in camera space + refractive index ratio
	void setPlane(const double normal[3], double distance) {
		for (int i = 0; i < 3; ++i) normal_[i] = normal[i];
		planeDist_ = distance; refresh();
	}
	void setRefractiveIndex(double n) { refrIndex_ = n; refresh(); }

	bool isRefractive() const { return snap_.is_refractive != 0; }
	bool isDistorted() const { return snap_.is_distorted != 0; }
	const double *C() const { return snap_.C; }
	const double *K() const { return snap_.K; }
	const double *Kinv() const { return snap_.Kinv; }
	const double *R() const { return snap_.R; }
	const double *t() const { return snap_.t; }
	const LensDistortions &lensDistortion() const { return dist_; }
	const double *principleRayDirection() const { return snap_.pdir; }

	// the POD the kernels consume; taken by the stereo classes at construction / initialize
	const srh_camera &snapshot() const { return snap_; }

private:
	void refresh() {
		// a camera set from P keeps being derived from P: running the orthonormalisation of Camera::set
		// a second time over the already orthonormal R could move its last bits
		if (fromP_) srh_camera_from_p(P_, dist_.data(), normal_, planeDist_, refrIndex_, &snap_);
		else srh_camera_from_krt(K_, R_, t_, dist_.data(), normal_, planeDist_, refrIndex_, &snap_);
	}
	std::string id_, name_;
	double K_[9], R_[9], t_[3], normal_[3];
	double P_[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
	bool fromP_ = false;
	double planeDist_ = 0.0, refrIndex_ = 1.0;
	LensDistortions dist_;
	srh_camera snap_;
};

typedef std::shared_ptr<Camera> CameraPtr;
