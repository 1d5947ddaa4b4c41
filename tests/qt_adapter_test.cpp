// qt_adapter_test.cpp -- drives the Qt binding (stereoreconstruction_amd/qt) the way the reference's GUI drives its
// stereo classes: the task is moved to a QThread, run() is invoked there (gui/mainwindow.cpp:1184-1191), progress
// arrives through the reference's signals, results are read after finished().
//
//   qt_adapter_test ingest  <image file> <scale> <out.raw>        (host only) image + mask as MultiViewStereo::initialize builds them
//   qt_adapter_test twoview <spec.txt> <out prefix>               (GPU) TwoViewStereo on raw RGBA inputs
//   qt_adapter_test mvs     <spec.txt> <out prefix>               (GPU) MultiViewStereo on image files
#include <QtCore/QFile>
#include <QtCore/QThread>
#include <QtCore/QTimer>
#include <QtWidgets/QApplication>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "qt_glue_test.hpp"          // the test's own Camera / ImageSet / ... and the glue over them (includes stereo_qt.hpp)

static bool readRaw(const std::string &path, std::vector<unsigned char> &out, size_t n) {
	std::ifstream f(path, std::ios::binary);
	out.resize(n);
	f.read(reinterpret_cast<char *>(out.data()), static_cast<std::streamsize>(n));
	return static_cast<size_t>(f.gcount()) == n;
}

static void writeRaw(const std::string &path, const void *p, size_t n) {
	std::ofstream f(path, std::ios::binary);
	f.write(static_cast<const char *>(p), static_cast<std::streamsize>(n));
}

static bool readCamera(std::istream &in, srh_camera &cam) {
	double K[9], R[9], t[3], dist[5];
	for (double &v : K) in >> v;
	for (double &v : R) in >> v;
	for (double &v : t) in >> v;
	for (double &v : dist) in >> v;
	return in.good() && srh_camera_from_krt(K, R, t, dist, nullptr, 0.0, 1.0, &cam) == SRH_OK;
}

static QImage imageFromRaw(const std::vector<unsigned char> &rgba, int w, int h) {
	QImage img(w, h, QImage::Format_ARGB32);
	for (int y = 0; y < h; ++y) {
		QRgb *s = reinterpret_cast<QRgb *>(img.scanLine(y));
		for (int x = 0; x < w; ++x) { const unsigned char *p = &rgba[(static_cast<size_t>(y)*w + x)*4]; s[x] = qRgba(p[0], p[1], p[2], p[3]); }
	}
	return img;
}

// The reference GUI's call sites of the two classes (gui/widgets/stereowidget.cpp:160, 263, 274, 321-322, 964-966, 990-994),
// as expressions over the same names: this function exists to be TYPE-CHECKED against stereo_qt.hpp (it also runs, on
// an uninitialised task: null maps, no image set).
static bool stereoWidgetCallSites(ProjectPtr project, CameraPtr leftView, CameraPtr rightView, ImageSetPtr imageSet,
                                  double mind, double maxd, int levels, double crossCheck, double scale)
{
	std::shared_ptr<MultiViewStereo> mvs(new MultiViewStereo);                      // :160  mvs(new MultiViewStereo)
	QImage img = mvs->depthMap(leftView);                                           // :263
	if (!img.isNull()) return false;
	img = mvs->depthMap(rightView);                                                 // :274
	const bool sameSet = mvs->imageSet() == imageSet;                               // :321-322
	std::vector<CameraPtr> views;
	views.push_back(leftView); views.push_back(rightView);
	mvs->initialize(project, imageSet, views, mind, maxd, levels, crossCheck, scale);   // :990-994
	QObject::connect(mvs.get(), SIGNAL(finished(const Task *)), mvs.get(), SLOT(cancel()));   // :996-998 (signal of the reference's Task)
	const QImage l = mvs->depthMap(leftView), r = mvs->depthMap(rightView);         // :964-966
	return !sameSet && mvs->imageSet() == imageSet && mvs->numSteps() == 2*mvs->numViews() && l.isNull() == r.isNull();
}

// run `task` on its own thread exactly as MainWindow::customEvent does; returns the progress steps seen
static std::vector<int> runOnThread(QApplication &app, Task *task, std::vector<std::string> &stages) {
	std::vector<int> steps;
	bool started = false, finished = false;
	QObject::connect(task, &Task::progressUpdate, &app, [&](int s) { steps.push_back(s); });
	QObject::connect(task, &Task::stageUpdate, &app, [&](QString s) { stages.push_back(s.toStdString()); });
	QObject::connect(task, &Task::started, &app, [&](const Task *) { started = true; });
	QObject::connect(task, &Task::finished, &app, [&](const Task *) { finished = true; app.quit(); });
	QThread thread;
	task->moveToThread(&thread);
	QObject::connect(&thread, &QThread::started, task, &Task::run);
	thread.start();
	QTimer::singleShot(600000, &app, [&]() { app.exit(2); });
	const int rc = app.exec();
	thread.quit();
	thread.wait();
	if (rc != 0 || !started || !finished) steps.push_back(-1000);
	return steps;
}

int main(int argc, char **argv) {
	qputenv("QT_QPA_PLATFORM", "offscreen");
	QApplication app(argc, argv);
	if (argc >= 5 && !strcmp(argv[1], "ingest")) {
		srq::Raster img; std::vector<unsigned char> mask;
		if (!srq::ingestViewFile(argv[2], atof(argv[3]), img, mask)) { fprintf(stderr, "cannot ingest %s\n", argv[2]); return 1; }
		std::ofstream f(argv[4], std::ios::binary);
		const int hdr[2] = { img.w, img.h };
		f.write(reinterpret_cast<const char *>(hdr), sizeof(hdr));
		f.write(reinterpret_cast<const char *>(img.rgba.data()), static_cast<std::streamsize>(img.rgba.size()));
		f.write(reinterpret_cast<const char *>(mask.data()), static_cast<std::streamsize>(mask.size()));
		printf("ingest %dx%d\n", img.w, img.h);
		return 0;
	}
	if (argc >= 3 && !strcmp(argv[1], "ply")) {
		// the reference's free function with its own signature (stereo/multiviewstereo.hpp:36-39), and runTask() called
		// from outside the class as the reference allows (stereo/twoviewstereo.hpp:52: public)
		std::vector<PLYPoint> pts;
		pts.push_back(PLYPoint(Eigen::Vector3d(1.5, -2.25, 1e-7), RGBA(255, 0, 17)));
		pts.push_back(PLYPoint(Eigen::Vector3d(123456.789, 0.1, 3.0), RGBA(1.9, 254.2, 128)));
		pts.push_back(PLYPoint(Eigen::Vector3d(0.0, 1.0, 2.0), RGBA(300.7, -4.2, 256)));   // the reference prints static_cast<int> of the doubles: no wrap
		outputPLYFile(argv[2], pts);
		void (TwoViewStereo::*rt)() = &TwoViewStereo::runTask;
		printf("ply %d\n", rt != nullptr ? 1 : 0);
		return 0;
	}
	if (argc >= 4 && !strcmp(argv[1], "twoview")) {
		// spec: w h minDepth maxDepth levels scale left.raw right.raw leftmask.raw|- rightmask.raw|-  then two cameras
		std::ifstream in(argv[2]);
		int w, h, levels; double zmin, zmax, scale; std::string lf, rf, lm, rm;
		in >> w >> h >> zmin >> zmax >> levels >> scale >> lf >> rf >> lm >> rm;
		srh_camera cl, cr;
		if (!readCamera(in, cl) || !readCamera(in, cr)) { fprintf(stderr, "bad spec\n"); return 1; }
		std::vector<unsigned char> L, R, ML, MR;
		if (!readRaw(lf, L, static_cast<size_t>(w)*h*4) || !readRaw(rf, R, static_cast<size_t>(w)*h*4)) { fprintf(stderr, "bad image\n"); return 1; }
		QImage qml, qmr;
		if (lm != "-") { if (!readRaw(lm, ML, static_cast<size_t>(w)*h*4)) return 1; qml = imageFromRaw(ML, w, h); }
		if (rm != "-") { if (!readRaw(rm, MR, static_cast<size_t>(w)*h*4)) return 1; qmr = imageFromRaw(MR, w, h); }
		CameraPtr leftView(new Camera("left", cl)), rightView(new Camera("right", cr));
		// the reference's constructor signature (stereo/twoviewstereo.hpp:44-47)
		TwoViewStereo *tv = new TwoViewStereo(leftView, imageFromRaw(L, w, h), qml, rightView, imageFromRaw(R, w, h), qmr, zmin, zmax, levels, scale);
		// the public epipolarCurve member with the reference's five arguments (twoviewstereo.hpp:66-70), before and
		// independent of computeDepthMaps; `view` is the camera the curve is drawn in
		for (int dir = 0; dir < 2; ++dir) {
			const std::vector<Eigen::Vector3d> curve = tv->epipolarCurve(Ray3d(w/2, h/2), Eigen::Vector3d(), Eigen::Vector3d(), VectorImage(),
			                                                            dir == 0 ? rightView : leftView);
			printf("curve%d", dir);
			for (const auto &pt : curve) printf(" %d,%d", static_cast<int>(pt[0]), static_cast<int>(pt[1]));
			printf("\n");
		}
		std::vector<std::string> stages;
		const std::vector<int> steps = runOnThread(app, tv, stages);
		printf("title %s\nnumSteps %d\nsteps", tv->title().toStdString().c_str(), tv->numSteps());
		for (int s : steps) printf(" %d", s);
		printf("\nthread_back %d\n", tv->thread() == app.thread() ? 1 : 0);     // Task::run moves the task back (gui/task.cpp:32)
		printf("error %s\n", tv->lastError().toStdString().c_str());
		const QImage dl = tv->leftDepthMap(), dr = tv->rightDepthMap();
		printf("maps %dx%d %dx%d\n", dl.width(), dl.height(), dr.width(), dr.height());
		writeRaw(std::string(argv[3]) + "_left.f64", tv->leftDepths().data(), tv->leftDepths().size()*sizeof(double));
		writeRaw(std::string(argv[3]) + "_right.f64", tv->rightDepths().data(), tv->rightDepths().size()*sizeof(double));
		if (!dl.isNull()) dl.save(QString::fromStdString(std::string(argv[3]) + "_left.png"));
		delete tv;
		return 0;
	}
	if (argc >= 4 && !strcmp(argv[1], "mvs")) {
		// spec: nviews minDepth maxDepth levels crossCheck scale; per view: id file, camera
		std::ifstream in(argv[2]);
		int n, levels; double zmin, zmax, cc, scale;
		in >> n >> zmin >> zmax >> levels >> cc >> scale;
		// the project side as the reference hands it over: cameras, an image set with one default image per camera
		ProjectPtr project(new Project);
		ImageSetPtr imageSet(new ImageSet);
		std::vector<CameraPtr> views;
		std::vector<std::string> ids(n);
		for (int v = 0; v < n; ++v) {
			std::string file;
			in >> ids[v] >> file;
			srh_camera cam;
			if (!readCamera(in, cam)) { fprintf(stderr, "bad spec\n"); return 1; }
			views.push_back(CameraPtr(new Camera(QString::fromStdString(ids[v]), cam)));
			imageSet->setDefaultImage(views.back(), QString::fromStdString(file));
		}
		printf("callsites %d\n", stereoWidgetCallSites(project, views[0], views[n > 1 ? 1 : 0], imageSet, zmin, zmax, levels, cc, scale) ? 1 : 0);
		MultiViewStereo *mvs = new MultiViewStereo();
		mvs->initialize(project, imageSet, views, zmin, zmax, levels, cc, scale);     // stereo/multiviewstereo.hpp:46-52
		printf("imageset %d\n", mvs->imageSet() == imageSet ? 1 : 0);
		std::vector<std::string> stages;
		const std::vector<int> steps = runOnThread(app, mvs, stages);
		printf("title %s\nnumViews %d\nnumSteps %d\nsteps", mvs->title().toStdString().c_str(), mvs->numViews(), mvs->numSteps());
		for (int s : steps) printf(" %d", s);
		printf("\nerror %s\n", mvs->lastError().toStdString().c_str());
		printf("unknown_view_null %d\n", (mvs->depthMap(CameraPtr(new Camera("stranger", srh_camera()))).isNull() && mvs->depthMap(CameraPtr()).isNull()) ? 1 : 0);
		for (int v = 0; v < n; ++v) {                                                  // (a view without an image file: null map)
			const QImage m = mvs->depthMap(views[v]);                                  // stereo/multiviewstereo.hpp:60
			printf("map_%s %dx%d\n", ids[v].c_str(), m.width(), m.height());
		}
		for (int v = 0; v < mvs->numViews(); ++v) {
			std::ostringstream o; o << argv[3] << "_" << v;
			writeRaw(o.str() + ".f64", mvs->depths(v).data(), mvs->depths(v).size()*sizeof(double));
			const int hdr[2] = { mvs->image(v).w, mvs->image(v).h };
			std::ofstream f(o.str() + ".img", std::ios::binary);
			f.write(reinterpret_cast<const char *>(hdr), sizeof(hdr));
			f.write(reinterpret_cast<const char *>(mvs->image(v).rgba.data()), static_cast<std::streamsize>(mvs->image(v).rgba.size()));
			f.write(reinterpret_cast<const char *>(mvs->mask(v).data()), static_cast<std::streamsize>(mvs->mask(v).size()));
		}
		delete mvs;
		return 0;
	}
	fprintf(stderr, "usage: see the header of tests/qt_adapter_test.cpp\n");
	return 64;
}
