// project.cpp -- project XML reader (reference: project/project.cpp:74-227).  The reference validates the
// file against project.xsd with QXmlSchemaValidator and walks a QDomDocument; here a small recursive
// descent reader builds the element tree and the same walk is done over it.  Checked on the way (the part
// of the schema the stereo path depends on): root element <project>; every <camera> has an id and a
// <projectionMatrix> with m11..m34; every <imageSet> has an id, every <image> a file.
#include "project.hpp"

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace {

struct Node {
	std::string name;
	std::vector<std::pair<std::string, std::string> > attrs;
	std::vector<Node> children;

	bool has(const std::string &a) const {
		for (size_t i = 0; i < attrs.size(); ++i) if (attrs[i].first == a) return true;
		return false;
	}
	std::string get(const std::string &a, const std::string &def = std::string()) const {   // getAttribute (:60-64)
		for (size_t i = 0; i < attrs.size(); ++i) if (attrs[i].first == a) return attrs[i].second;
		return def;
	}
	const Node *first(const std::string &n) const {                                         // firstChildElement
		for (size_t i = 0; i < children.size(); ++i) if (children[i].name == n) return &children[i];
		return nullptr;
	}
};

struct XmlError { };

class Reader {
public:
	explicit Reader(const std::string &text) : s(text), i(0) { }

	Node document() {
		prolog();
		Node root = element();
		prolog();
		if (i != s.size()) throw XmlError();
		return root;
	}

private:
	const std::string &s;
	size_t i;

	bool starts(const char *lit) const { return s.compare(i, std::string(lit).size(), lit) == 0; }
	void skipWs() { while (i < s.size() && std::isspace(static_cast<unsigned char>(s[i]))) ++i; }
	void skipPast(const char *lit) {
		const size_t k = s.find(lit, i);
		if (k == std::string::npos) throw XmlError();
		i = k + std::string(lit).size();
	}
	void prolog() {                                      // whitespace, comments, <?...?>, <!DOCTYPE ...>
		for (;;) {
			skipWs();
			if (starts("<!--")) skipPast("-->");
			else if (starts("<?")) skipPast("?>");
			else if (starts("<!")) skipPast(">");
			else return;
		}
	}
	std::string name() {
		const size_t b = i;
		while (i < s.size() && (std::isalnum(static_cast<unsigned char>(s[i])) || s[i] == '_' || s[i] == '-' || s[i] == ':' || s[i] == '.')) ++i;
		if (i == b) throw XmlError();
		return s.substr(b, i - b);
	}
	static std::string unescape(const std::string &v) {
		std::string out;
		for (size_t k = 0; k < v.size(); ++k) {
			if (v[k] != '&') { out += v[k]; continue; }
			const size_t e = v.find(';', k);
			if (e == std::string::npos) throw XmlError();
			const std::string ent = v.substr(k + 1, e - k - 1);
			if (ent == "amp") out += '&';
			else if (ent == "lt") out += '<';
			else if (ent == "gt") out += '>';
			else if (ent == "quot") out += '"';
			else if (ent == "apos") out += '\'';
			else if (!ent.empty() && ent[0] == '#') {
				const long code = ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X') ? std::strtol(ent.c_str() + 2, nullptr, 16)
				                                                                       : std::strtol(ent.c_str() + 1, nullptr, 10);
				if (code <= 0 || code > 127) throw XmlError();      // project files are ASCII
				out += static_cast<char>(code);
			} else throw XmlError();
			k = e;
		}
		return out;
	}
	Node element() {
		if (i >= s.size() || s[i] != '<') throw XmlError();
		++i;
		Node n;
		n.name = name();
		for (;;) {
			skipWs();
			if (i >= s.size()) throw XmlError();
			if (s[i] == '/') {                                   // <name ... />
				if (i + 1 >= s.size() || s[i + 1] != '>') throw XmlError();
				i += 2;
				return n;
			}
			if (s[i] == '>') { ++i; break; }
			const std::string an = name();
			skipWs();
			if (i >= s.size() || s[i] != '=') throw XmlError();
			++i;
			skipWs();
			if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) throw XmlError();
			const char q = s[i++];
			const size_t e = s.find(q, i);
			if (e == std::string::npos) throw XmlError();
			if (n.has(an)) throw XmlError();                     // duplicate attribute: not well-formed
			n.attrs.push_back(std::make_pair(an, unescape(s.substr(i, e - i))));
			i = e + 1;
		}
		for (;;) {                                               // content: text is skipped, children kept
			const size_t lt = s.find('<', i);
			if (lt == std::string::npos) throw XmlError();
			i = lt;
			if (starts("<!--")) { skipPast("-->"); continue; }
			if (starts("<![CDATA[")) { skipPast("]]>"); continue; }
			if (starts("<?")) { skipPast("?>"); continue; }
			if (starts("</")) {
				i += 2;
				if (name() != n.name) throw XmlError();
				skipWs();
				if (i >= s.size() || s[i] != '>') throw XmlError();
				++i;
				return n;
			}
			n.children.push_back(element());
		}
	}
};

// QString::toDouble: 0.0 when the text is not a number
double toDouble(const std::string &v) {
	const char *b = v.c_str();
	char *e = nullptr;
	const double d = std::strtod(b, &e);
	if (e == b) return 0.0;
	while (*e && std::isspace(static_cast<unsigned char>(*e))) ++e;
	return *e ? 0.0 : d;
}

bool isAbsolute(const std::string &p) { return !p.empty() && p[0] == '/'; }

std::string dirOf(const std::string &path) {                 // QDir(projectPath).cdUp()
	const size_t k = path.find_last_of('/');
	if (k == std::string::npos) return ".";
	return k == 0 ? "/" : path.substr(0, k);
}

std::string join(const std::string &dir, const std::string &rel) {
	if (rel.empty()) return dir;
	if (isAbsolute(rel)) return rel;
	return (dir.empty() || dir[dir.size() - 1] == '/') ? dir + rel : dir + "/" + rel;
}

} // namespace

void ImageSet::addImageForCamera(CameraPtr cam, ProjectImagePtr image) {
	image->setCamera(cam);
	images_.push_back(image);
	if (cam && defaults_.find(cam) == defaults_.end()) defaults_[cam] = image;
}

ProjectImagePtr ImageSet::defaultImageForCamera(CameraPtr cam) const {
	const std::map<CameraPtr, ProjectImagePtr>::const_iterator it = defaults_.find(cam);
	return it == defaults_.end() ? ProjectImagePtr() : it->second;
}

CameraPtr Project::camera(const std::string &id) const {
	const std::map<std::string, CameraPtr>::const_iterator it = cameras_.find(id);
	return it == cameras_.end() ? CameraPtr() : it->second;
}

ImageSetPtr Project::imageSet(const std::string &id) const {
	const std::map<std::string, ImageSetPtr>::const_iterator it = imageSets_.find(id);
	return it == imageSets_.end() ? ImageSetPtr() : it->second;
}

Project::Project(const std::string &projectPath) : projectPath_(projectPath) {
	if (projectPath.empty()) return;
	std::ifstream f(projectPath.c_str(), std::ios::binary);
	if (!f) throw std::runtime_error("Failed to open file");
	std::ostringstream buf;
	buf << f.rdbuf();
	const std::string text = buf.str();
	Node root;
	try {
		Reader rd(text);
		root = rd.document();
	} catch (const XmlError &) {
		throw std::runtime_error("Failed to set XML content");
	}
	if (root.name != "project") throw std::runtime_error("Failed to validate");

	// ---- cameras (project.cpp:112-189)
	static const char *const M[12] = { "m11", "m12", "m13", "m14", "m21", "m22", "m23", "m24", "m31", "m32", "m33", "m34" };
	if (const Node *cams = root.first("cameras")) {
		for (size_t k = 0; k < cams->children.size(); ++k) {
			const Node &cn = cams->children[k];
			if (cn.name != "camera") continue;
			if (!cn.has("id")) throw std::runtime_error("Failed to validate");
			CameraPtr cam(new Camera(cn.get("id")));
			cam->setName(cn.get("name", cam->id()));
			const Node *pm = cn.first("projectionMatrix");
			if (!pm) throw std::runtime_error("Failed to validate");   // required by the schema
			double P[12];
			for (int e = 0; e < 12; ++e) {
				if (!pm->has(M[e])) throw std::runtime_error("Failed to validate");
				P[e] = toDouble(pm->get(M[e]));
			}
			cam->setP(P);
			if (const Node *ld = cn.first("lensDistortion")) {
				LensDistortions dist;
				dist[0] = toDouble(ld->get("k1", "0"));
				dist[1] = toDouble(ld->get("k2", "0"));
				dist[2] = toDouble(ld->get("p1", "0"));
				dist[3] = toDouble(ld->get("p2", "0"));
				dist[4] = toDouble(ld->get("k3", "0"));
				cam->setLensDistortion(dist);
			}
			if (const Node *rn = cn.first("refractiveInterface")) {
				const double px = toDouble(rn->get("px", "0.0")), py = toDouble(rn->get("py", "0.0"));
				const double dist = toDouble(rn->get("dist", "0.0"));
				cam->setRefractiveIndex(toDouble(rn->get("refractiveRatio", "1.0")));
				// Plane3d(K^-1 * (px, py, 1), dist): the interface normal goes through pixel (px, py)
				const double *Ki = cam->Kinv();
				const double n[3] = { (Ki[0]*px + Ki[1]*py) + Ki[2]*1.0, (Ki[3]*px + Ki[4]*py) + Ki[5]*1.0,
				                      (Ki[6]*px + Ki[7]*py) + Ki[8]*1.0 };
				cam->setPlane(n, dist);
			}
			cameras_[cam->id()] = cam;
		}
	}

	// ---- image sets (project.cpp:191-225)
	if (const Node *sets = root.first("imageSets")) {
		for (size_t k = 0; k < sets->children.size(); ++k) {
			const Node &sn = sets->children[k];
			if (sn.name != "imageSet") continue;
			if (!sn.has("id")) throw std::runtime_error("Failed to validate");
			ImageSetPtr set(new ImageSet(sn.get("id")));
			set->setName(sn.get("name", set->id()));
			std::string rootDir = dirOf(projectPath);
			if (sn.has("root")) rootDir = join(rootDir, sn.get("root"));
			set->setRoot(rootDir);
			for (size_t m = 0; m < sn.children.size(); ++m) {
				const Node &in = sn.children[m];
				if (!in.has("file")) throw std::runtime_error("Failed to validate");
				ProjectImagePtr image(new ProjectImage(join(set->root(), in.get("file"))));
				image->setExposure(toDouble(in.get("exposure", "-1.0")));
				const CameraPtr cam = camera(in.get("for"));
				if (cam) set->addImageForCamera(cam, image);
			}
			if (!set->images().empty()) imageSets_[set->id()] = set;
		}
	}
}

std::vector<uint8_t> maskFromAlpha(const Image &img) {
	std::vector<uint8_t> mask(static_cast<size_t>(img.w)*img.h, 1);
	for (size_t k = 0; k < mask.size(); ++k)
		if (img.rgba[4*k + 3] != 255) mask[k] = 0;             // "If not fully opaque, we ignore that pixel"
	return mask;
}
