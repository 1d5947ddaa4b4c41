"""The Qt binding (stereoreconstruction_amd/qt: TwoViewStereo / MultiViewStereo derived from the reference's own
Task, QImage in and out) driven by tests/qt_adapter_test.cpp the way the reference's GUI drives its stereo classes:
the task is moved to a QThread, run() is invoked there, progress arrives through the reference's signals.

The binary is built from /root/reference/gui/task.{hpp,cpp} where they lie (stereoreconstruction_amd/qt/Makefile)
and travels to the GPU box as a built artefact; without it (a checkout that never saw the reference) the tests skip."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "stereoreconstruction_amd", "qt", "_build", "qt_adapter_test")
ENV = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib:" + os.environ.get("LD_LIBRARY_PATH", ""),
           QT_QPA_PLATFORM="offscreen")
needs_bin = pytest.mark.skipif(not os.path.exists(BIN), reason="Qt binding not built (needs /root/reference + conda Qt)")


def _run(*args):
    r = subprocess.run([BIN] + [str(a) for a in args], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return dict(line.split(" ", 1) if " " in line else (line, "") for line in r.stdout.strip().splitlines())


def _read_ingest(path):
    raw = open(path, "rb").read()
    w, h = np.frombuffer(raw[:8], np.int32)
    img = np.frombuffer(raw[8:8 + w * h * 4], np.uint8).reshape(h, w, 4)
    mask = np.frombuffer(raw[8 + w * h * 4:8 + w * h * 5], np.uint8).reshape(h, w)
    return img, mask


@needs_bin
@pytest.mark.skipif(not os.path.isdir("/root/reference/example/images/bunny"), reason="example images live in /root/reference")
def test_ingest_of_the_example_images_equals_the_fixture():
    """MultiViewStereo::initialize's image path (multiviewstereo.cpp:216-241: decode, smooth-scale, alpha -> mask on
    a fast-scaled copy) through the binding == tests/golden/bunny_pair.npz (made by tests/golden/make_fixtures.py)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "bunny_pair.npz"))
    with tempfile.TemporaryDirectory() as td:
        for cam, side in (("7310085", "left"), ("7310087", "right")):
            out = os.path.join(td, side + ".raw")
            _run("ingest", "/root/reference/example/images/bunny/%s.png" % cam, float(g["scale"][0]), out)
            img, mask = _read_ingest(out)
            assert np.array_equal(img, g[side + "_rgba"]) and np.array_equal(mask, g[side + "_mask"])


@needs_bin
def test_output_ply_file_with_the_reference_signature():
    """outputPLYFile(const std::string &, const std::vector<PLYPoint> &) with PLYPoint = std::pair<Ray3d::Point, RGBA>
    (stereo/multiviewstereo.hpp:36-39) through the Qt binding's header: the text of multiviewstereo.cpp:291-315
    (operator<< of an ofstream: six significant digits; colours as static_cast<int>)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "p.ply")
        assert _run("ply", out)["ply"] == "1"
        want = ("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar diffuse_red\nproperty uchar diffuse_green\nproperty uchar diffuse_blue\nend_header\n"
                "1.5 -2.25 1e-07 255 0 17\n123457 0.1 3 1 254 128\n0 1 2 300 -4 256\n")   # (components outside 0 .. 255 go out as the ints they are)
        assert open(out).read() == want


@needs_bin
def test_ingest_alpha_rule_and_opaque_images():
    """alpha == 255 <=> mask WHITE, decided on a fast-scaled copy; an image without alpha channel gets an all-WHITE mask."""
    from PIL import Image
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (40, 64, 4), dtype=np.uint8)
    a[..., 3] = np.where(rng.random((40, 64)) < 0.6, 255, rng.integers(0, 255, (40, 64)))
    with tempfile.TemporaryDirectory() as td:
        Image.fromarray(a, "RGBA").save(os.path.join(td, "a.png"))
        Image.fromarray(a[..., :3].copy(), "RGB").save(os.path.join(td, "b.png"))
        _run("ingest", os.path.join(td, "a.png"), 1.0, os.path.join(td, "a.raw"))
        img, mask = _read_ingest(os.path.join(td, "a.raw"))
        assert img.shape == (40, 64, 4) and np.array_equal(mask, (a[..., 3] == 255).astype(np.uint8))
        assert np.array_equal(img[mask == 1], a[mask == 1])        # opaque pixels pass through a scale-1 "smooth" scaling
        _run("ingest", os.path.join(td, "b.png"), 0.5, os.path.join(td, "b.raw"))
        img, mask = _read_ingest(os.path.join(td, "b.raw"))
        assert img.shape == (20, 32, 4) and (mask == 1).all()


def _camera_lines(K, R, t, dist):
    d = np.zeros(5) if dist is None else np.asarray(dist, dtype=np.float64)
    return " ".join(repr(float(v)) for v in list(np.asarray(K).reshape(9)) + list(np.asarray(R).reshape(9)) + list(np.asarray(t).reshape(3)) + list(d))


@needs_bin
@pytest.mark.gpu
def test_twoview_through_the_qt_binding(hip_ctx):
    """TwoViewStereo(camera, QImage, QImage mask, ...) on a QThread: steps 1,3,5,8, task moved back to the GUI
    thread, depth maps bit-identical to the C-ABI called directly."""
    case = cases.get_twoview("geodesic_masks", w=64, h=40, D=16)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    want_l, want_r = hip_ctx.twoview_compute(0, 1, p)
    with tempfile.TemporaryDirectory() as td:
        names = []
        for i, (rgba, mask, (K, R, t), dist, plane) in enumerate(case["views"]):
            rgba.tofile(os.path.join(td, "img%d.raw" % i))
            m = np.zeros(rgba.shape, np.uint8)
            m[..., 3] = 255
            m[mask == 1] = 255                                   # WHITE where the mask is set, opaque black elsewhere
            m.tofile(os.path.join(td, "mask%d.raw" % i))
            names.append((os.path.join(td, "img%d.raw" % i), os.path.join(td, "mask%d.raw" % i)))
        h, w = case["views"][0][0].shape[:2]
        pr = case["params"]
        spec = os.path.join(td, "spec.txt")
        with open(spec, "w") as f:
            f.write("%d %d %r %r %d %r %s %s %s %s\n" % (w, h, pr["min_depth"], pr["max_depth"], pr["num_depth_levels"],
                                                       pr["image_scale"], names[0][0], names[1][0], names[0][1], names[1][1]))
            for (rgba, mask, (K, R, t), dist, plane) in case["views"]:
                f.write(_camera_lines(K, R, t, dist) + "\n")
        out = _run("twoview", spec, os.path.join(td, "out"))
        assert out["title"] == "Two-View Stereo" and out["numSteps"] == "8"
        assert out["steps"].split() == ["1", "3", "5", "8"] and out["thread_back"] == "1" and out["error"].strip() == ""
        assert out["maps"] == "%dx%d %dx%d" % (w, h, w, h)
        got_l = np.fromfile(os.path.join(td, "out_left.f64")).reshape(h, w)
        got_r = np.fromfile(os.path.join(td, "out_right.f64")).reshape(h, w)
        assert np.array_equal(got_l.view(np.uint64), want_l.view(np.uint64))
        assert np.array_equal(got_r.view(np.uint64), want_r.view(np.uint64))
        assert os.path.getsize(os.path.join(td, "out_left.png")) > 100
        # the public epipolarCurve member equals the C-ABI's curve query (TwoViewStereo::epipolarCurve, twoviewstereo.hpp:66-70)
        for d, (a, b) in enumerate(((0, 1), (1, 0))):
            want = hip_ctx.epipolar_curves(a, b, p, [(w // 2, h // 2)])[0]
            got = [tuple(int(v) for v in tok.split(",")) for tok in out["curve%d" % d].split()]
            assert got == [tuple(int(v) for v in q) for q in want] and len(got) > 4


@needs_bin
@pytest.mark.gpu
def test_multiview_through_the_qt_binding(hip_ctx):
    """MultiViewStereo::initialize from image FILES (RGBA PNGs whose alpha is the object mask, one view without a file
    is skipped) + runTask on a QThread: steps 0..2V-1, depth maps bit-identical to the C-ABI driven directly on the
    images and masks the binding ingested; unknown view -> null QImage."""
    from PIL import Image
    from stereoreconstruction_amd import capi
    case = cases.get_mvs("mvs_geodesic", w=48, h=36, D=12, nviews=4)
    pr = case["params"]
    with tempfile.TemporaryDirectory() as td:
        spec = os.path.join(td, "spec.txt")
        with open(spec, "w") as f:
            f.write("%d %r %r %d %r %r\n" % (len(case["views"]) + 1, pr["min_depth"], pr["max_depth"], pr["num_depth_levels"],
                                             pr["cross_check_threshold"], 1.0))
            for v, (rgba, mask, (K, R, t), dist, plane) in enumerate(case["views"]):
                a = rgba.copy()
                a[..., 3] = np.where(mask == 1, 255, 60)
                Image.fromarray(a, "RGBA").save(os.path.join(td, "v%d.png" % v))
                f.write("cam%d %s\n%s\n" % (v, os.path.join(td, "v%d.png" % v), _camera_lines(K, R, t, dist)))
                if v == 1:                                       # a camera whose image file does not exist
                    f.write("ghost %s\n%s\n" % (os.path.join(td, "missing.png"), _camera_lines(K, R, t, dist)))
        out = _run("mvs", spec, os.path.join(td, "out"))
        V = len(case["views"])
        assert out["title"] == "Multi-view Stereo" and out["numViews"] == str(V) and out["numSteps"] == str(2 * V)
        assert out["steps"].split() == [str(s) for s in range(2 * V)] and out["error"].strip() == ""
        assert out["unknown_view_null"] == "1"
        # the reference's signatures: initialize(ProjectPtr, ImageSetPtr, views, ...), imageSet(), depthMap(CameraPtr);
        # gui/widgets/stereowidget.cpp's call sites, compiled against the binding's header and run
        assert out["callsites"] == "1" and out["imageset"] == "1"
        h0, w0 = case["views"][0][0].shape[:2]
        assert out["map_cam0"] == "%dx%d" % (w0, h0) and out["map_ghost"] == "0x0"
        cams, p = cases.hip_inputs(case)
        for v in range(V):
            img, mask = _read_ingest(os.path.join(td, "out_%d.img" % v))
            assert np.array_equal(mask, case["views"][v][1])
            hip_ctx.upload_view(v, img, mask, cams[v])
        neigh = capi.mvs_neighbours(cams, p)
        for v in range(V):
            hip_ctx.mvs_initial_estimate(v, neigh[v], p)
        for v in range(V):
            hip_ctx.mvs_cross_check(list(range(V)), v, p)
        for v in range(V):
            h, w = case["views"][v][0].shape[:2]
            got = np.fromfile(os.path.join(td, "out_%d.f64" % v)).reshape(h, w)
            assert np.array_equal(got.view(np.uint64), hip_ctx.download_depth(v).view(np.uint64)), v
