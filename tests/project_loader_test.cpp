// project_loader_test.cpp -- loads a project XML with the Qt-free Project class
// (stereoreconstruction_amd/host/project.*) and prints what the stereo path would be fed, as JSON.
//   project_loader_test file.xml          -> JSON on stdout, exit 0
//   a load failure prints the exception text on stderr and exits 3
#include <cstdio>
#include <stdexcept>

#include "project.hpp"

static void arr(const char *name, const double *v, int n, bool comma = true) {
	printf("\"%s\": [", name);
	for (int i = 0; i < n; ++i) printf("%s%.17g", i ? ", " : "", v[i]);
	printf("]%s", comma ? ", " : "");
}

int main(int argc, char **argv) {
	if (argc != 2) { fprintf(stderr, "usage: %s project.xml\n", argv[0]); return 2; }
	ProjectPtr prj;
	try {
		prj.reset(new Project(argv[1]));
	} catch (const std::runtime_error &e) {
		fprintf(stderr, "%s\n", e.what());
		return 3;
	}
	printf("{\"cameras\": {");
	bool first = true;
	for (const auto &kv : prj->cameras()) {
		const Camera &c = *kv.second;
		const srh_camera &s = c.snapshot();
		printf("%s\"%s\": {\"name\": \"%s\", ", first ? "" : ", ", c.id().c_str(), c.name().c_str());
		arr("K", s.K, 9); arr("R", s.R, 9); arr("t", s.t, 3); arr("C", s.C, 3); arr("pdir", s.pdir, 3);
		arr("Kinv", s.Kinv, 9); arr("dist", s.dist, 5); arr("plane_normal", s.plane_normal, 3);
		printf("\"plane_dist\": %.17g, \"refr_index\": %.17g, \"is_distorted\": %d, \"is_refractive\": %d}",
		       s.plane_dist, s.refr_index, s.is_distorted, s.is_refractive);
		first = false;
	}
	printf("}, \"imageSets\": {");
	first = true;
	for (const auto &kv : prj->imageSets()) {
		const ImageSet &is = *kv.second;
		printf("%s\"%s\": {\"name\": \"%s\", \"root\": \"%s\", \"images\": [", first ? "" : ", ", is.id().c_str(),
		       is.name().c_str(), is.root().c_str());
		for (size_t i = 0; i < is.images().size(); ++i) {
			const ProjectImage &im = *is.images()[i];
			printf("%s{\"file\": \"%s\", \"camera\": \"%s\", \"exposure\": %.17g, \"default\": %d}", i ? ", " : "",
			       im.file().c_str(), im.camera()->id().c_str(), im.exposure(),
			       is.defaultImageForCamera(im.camera()) == is.images()[i] ? 1 : 0);
		}
		printf("]}");
		first = false;
	}
	printf("}}\n");
	// alpha -> mask rule (multiviewstereo.cpp:225-234)
	Image img(3, 1);
	img.pixel(1, 0)[3] = 254;
	const std::vector<uint8_t> m = maskFromAlpha(img);
	return (m[0] == 1 && m[1] == 0 && m[2] == 1) ? 0 : 4;
}
