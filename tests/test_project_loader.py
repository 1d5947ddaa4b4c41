"""project XML -> cameras (Camera::setP RQ factorisation, lens distortion, refractive
interface) and image sets, through the Qt-free host classes; checked against the oracle's restatement of
Camera::setP (bit for bit: same algorithm written twice) and against the LAPACK-based decomposition the bunny
fixture was made with (two independent implementations; Eigen itself is absent, so parity with the reference's
own binary is unpinned at the last-ulp level)."""
import json
import os
import subprocess
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import oracle_ffi as O
from stereoreconstruction_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "stereoreconstruction_amd", "host")
LIBDIR = os.path.join(ROOT, "stereoreconstruction_amd")
FIXTURE = os.path.join(ROOT, "tests", "golden", "project_fixture.xml")


@pytest.fixture(scope="module")
def loader(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("loader"))
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    exe = os.path.join(tmp, "project_loader_test")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
                           os.path.join(ROOT, "tests", "project_loader_test.cpp"),
                           os.path.join(HOST, "libstereo_recon_host.a"),
                           "-L" + LIBDIR, "-lstereo_recon_hip", "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def _fixture_P(cam_id):
    root = ET.parse(FIXTURE).getroot()
    cam = [c for c in root.find("cameras") if c.get("id") == cam_id][0]
    pm = cam.find("projectionMatrix")
    return np.array([[float(pm.get("m%d%d" % (i, j))) for j in (1, 2, 3, 4)] for i in (1, 2, 3)])


def test_project_file_gives_the_oracles_cameras(loader):
    out = json.loads(subprocess.check_output([loader, FIXTURE]))
    cams = out["cameras"]
    assert sorted(cams) == ["7310085", "7310087", "housing"]
    assert cams["7310085"]["name"] == "left & centre"            # entity in an attribute value
    assert cams["7310087"]["name"] == "7310087"                  # name defaults to the id (project.cpp:117)
    for cid in ("7310085", "7310087", "housing"):
        P = _fixture_P(cid)
        got = cams[cid]
        if cid == "housing":
            Kinv = np.array(O.camera_set_p(P).Kinv).reshape(3, 3)
            n = Kinv @ np.array([512.5, 380.25, 1.0])
            want = O.camera_set_p(P, None, n, 0.1, 1.333)
            assert got["is_refractive"] == 1 and got["is_distorted"] == 0
        else:
            dist = got["dist"]
            want = O.camera_set_p(P, np.array(dist))
            assert got["is_distorted"] == 1 and got["is_refractive"] == 0
        for f in ("K", "R", "t", "C", "pdir", "Kinv", "dist", "plane_normal"):
            a, b = np.array(got[f]), np.array(getattr(want, f))
            if f == "plane_normal" and cid == "housing":        # numpy's matrix product vs the loader's left-to-right sum
                assert np.allclose(a, b, rtol=0, atol=1e-15), (cid, f)
            else:
                assert np.array_equal(a, b), (cid, f, a, b)
        assert got["plane_dist"] == want.plane_dist and got["refr_index"] == want.refr_index
    # lens distortion attributes are optional and default to 0 (7310087 has no p1)
    assert cams["7310087"]["dist"][2] == 0.0 and cams["7310087"]["dist"][3] == 0.006
    # K is a calibration matrix after the sign fix; R is a rotation
    for cid in cams:
        K = np.array(cams[cid]["K"]).reshape(3, 3)
        R = np.array(cams[cid]["R"]).reshape(3, 3)
        assert K[0, 0] > 0 and K[1, 1] > 0 and K[2, 2] > 0 and K[1, 0] == 0 and K[2, 0] == 0 and K[2, 1] == 0
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
        P = _fixture_P(cid)
        P = P / np.sum(P[2, :3] ** 2)
        assert np.allclose(K @ np.column_stack([R, np.array(cams[cid]["t"])]), P, rtol=1e-12, atol=1e-9) or \
            np.allclose(np.abs(K @ np.column_stack([R, np.array(cams[cid]["t"])])), np.abs(P), rtol=1e-9, atol=1e-9)


def test_decomposition_agrees_with_the_lapack_fixture(loader):
    """tests/golden/bunny_pair.npz holds K, R, t of the two bunny cameras from numpy's (LAPACK) QR."""
    out = json.loads(subprocess.check_output([loader, FIXTURE]))["cameras"]
    fx = np.load(os.path.join(ROOT, "tests", "golden", "bunny_pair.npz"))
    for tag, cid in (("left", "7310085"), ("right", "7310087")):
        for f in ("K", "t"):
            want = fx[tag + "_" + f].reshape(-1)
            assert np.allclose(np.array(out[cid][f]), want, rtol=1e-12, atol=1e-12 * np.abs(want).max()), (cid, f)
        assert np.allclose(np.array(out[cid]["R"]), fx[tag + "_R"].reshape(-1), rtol=0, atol=1e-12)
        assert np.array_equal(np.array(out[cid]["dist"]), fx[tag + "_dist"])
        # and the C-ABI helper the loader calls is the same function
        lib = capi.camera_from_p(_fixture_P(cid), fx[tag + "_dist"])
        assert np.array_equal(np.array(lib.K), np.array(out[cid]["K"]))


def test_image_sets(loader):
    sets = json.loads(subprocess.check_output([loader, FIXTURE]))["imageSets"]
    assert sorted(sets) == ["000000", "bunny"]                   # a set without images is dropped (project.cpp:222-223)
    first = sets["000000"]
    base = os.path.dirname(FIXTURE)
    assert first["name"] == "first" and first["root"] == os.path.join(base, "images")
    files = [(os.path.basename(i["file"]), i["camera"], i["default"]) for i in first["images"]]
    # the first image of a camera is its default; images of unknown cameras are not added
    assert files == [("7310085_1.jpg", "7310085", 1), ("7310087_1.jpg", "7310087", 1), ("7310085_1b.jpg", "7310085", 0)]
    assert first["images"][1]["exposure"] == 0.25 and first["images"][0]["exposure"] == -1.0
    assert first["images"][0]["file"] == os.path.join(base, "images", "7310085_1.jpg")
    assert sets["bunny"]["root"] == "/abs/bunny" and sets["bunny"]["name"] == "bunny"


@pytest.mark.parametrize("text,message", [
    (None, "Failed to open file"),
    ("<project><cameras><camera id='a'>", "Failed to set XML content"),
    ("<other/>", "Failed to validate"),
    ("<project><cameras><camera id='a'/></cameras></project>", "Failed to validate"),
    ("<project><cameras><camera id='a'><projectionMatrix m11='1'/></camera></cameras></project>", "Failed to validate"),
])
def test_load_errors_are_the_references(loader, tmp_path, text, message):
    path = str(tmp_path / "p.xml")
    if text is not None:
        with open(path, "w") as f:
            f.write(text)
    r = subprocess.run([loader, path], capture_output=True, text=True)
    assert r.returncode == 3 and r.stderr.strip() == message
