#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD CONTAINER only: needs
/root/reference and oracle/_ref/libref_pieces.so).

  ref_pieces.npz   outputs of the reference's own unmodified sources (util/vectorimage.cpp,
                   util/lineiter.cpp, stereo/adaptiveweight.cpp, stereo/geodesicweight.cpp) on
                   seeded inputs: windows, line point lists, bilinear samples.  Pins SURVEY 8(a)
                   rows 1, 2, 3, 7 of the oracle.
  bunny_pair.npz   the example project's `bunny` views 7310085 / 7310087 as the reference ingests
                   them (Qt smooth scaling to 0.25, alpha mask) + their K,R,t (Camera::setP path
                   restated in numpy: unpinned Eigen QR) + lens distortion.  BASELINE config C1.

  bunny_views.npz  all EIGHT views of the example project's `bunny` image set as the reference ingests them
                   (MultiViewStereo::initialize, multiviewstereo.cpp:216-241, through the reference's own Qt calls at scale
                   0.25), in camera-id order, + bunny_project.xml: the eight cameras' numbers (projection matrices, lens
                   distortion) and the `bunny` image set of example/project.xml with the image files renamed <id>.raw.
                   The reference's only live entry point -- StereoWidget -> MultiViewStereo on every camera of the
                   project (gui/widgets/stereowidget.cpp:974-1002) -- on real, distorted, masked photographs.

Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import ctypes as C
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_ffi as O  # noqa: E402

REF = "/root/reference"


def make_ref_pieces():
    R = O.ref()
    rng = np.random.default_rng(20261003)
    out = {}
    # images: pure noise, smooth gradient + noise, two-tone with an edge
    h, w = 21, 29
    imgs = []
    a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8); a[..., 3] = 255
    yy, xx = np.mgrid[0:h, 0:w]
    b = np.stack([(xx * 7 + yy * 3) % 256, (xx * 2 + yy * 9) % 256, (xx + yy) * 4 % 256, np.full((h, w), 255)], -1).astype(np.uint8)
    b[..., :3] = np.clip(b[..., :3].astype(int) + rng.integers(-6, 7, (h, w, 3)), 0, 255)
    c = np.zeros((h, w, 4), np.uint8); c[..., 3] = 255; c[:, : w // 2, :3] = (200, 180, 40); c[:, w // 2:, :3] = (30, 60, 220)
    c[..., :3] = np.clip(c[..., :3].astype(int) + rng.integers(-3, 4, (h, w, 3)), 0, 255)
    imgs = [a, b, c]
    out["images"] = np.stack(imgs)
    centres = [(0, 0), (w - 1, h - 1), (w - 1, 0), (0, h - 1), (5, 5), (14, 10), (w // 2, 3), (2, h - 2), (-1, 4), (w, h)]
    out["centres"] = np.array(centres, np.int32)
    for kind, kname in ((0, "adaptive"), (1, "geodesic")):
        for r in (1, 2, 5):
            ws = 2 * r + 1
            arr = np.zeros((len(imgs), len(centres), ws, ws))
            for ii, im in enumerate(imgs):
                hnd = R.refp_image_create(O.u8ptr(np.ascontiguousarray(im)), w, h)
                for ci, (cx, cy) in enumerate(centres):
                    tmp = np.zeros((ws, ws))
                    R.refp_weights(hnd, cx, cy, r, kind, O.dptr(tmp))
                    arr[ii, ci] = tmp
                R.refp_image_free(hnd)
            out["weights_%s_r%d" % (kname, r)] = arr
    # lines: (x0,y0,x1,y1, clip) -> points
    segs = []
    for i in range(400):
        s = rng.uniform(-30, 60, 4)
        if i % 3 == 0:
            s = np.round(s)
        if i % 7 == 0:
            s[1] = s[3]          # horizontal
        if i % 11 == 0:
            s[0] = s[2]          # vertical
        segs.append(s)
    segs = np.array(segs)
    out["line_segments"] = segs
    for clip in (0, 1):
        pts, offs = [], [0]
        buf = np.zeros((4096, 2), np.int32)
        for s in segs:
            n = R.refp_line_points(s[0], s[1], s[2], s[3], clip, w, h, O.iptr(buf), 4096)
            pts.append(buf[:n].copy()); offs.append(offs[-1] + n)
        out["line_points_clip%d" % clip] = np.concatenate(pts) if offs[-1] else np.zeros((0, 2), np.int32)
        out["line_offsets_clip%d" % clip] = np.array(offs, np.int64)
    # samples
    hnd = R.refp_image_create(O.u8ptr(np.ascontiguousarray(a)), w, h)
    xy = rng.uniform(-2, max(w, h) + 1, (600, 2))
    xy[::4] = np.floor(xy[::4])
    res = np.zeros((600, 4)); tmp = np.zeros(4)
    for k, (x, y) in enumerate(xy):
        v = R.refp_image_sample(hnd, x, y, O.dptr(tmp))
        res[k, :3] = tmp[:3] if v else np.nan
        res[k, 3] = v
    R.refp_image_free(hnd)
    out["sample_xy"] = xy
    out["sample_rgbv"] = res
    out["gray_of_10_20_30"] = np.array([R.refp_to_gray(10, 20, 30)])
    np.savez_compressed(os.path.join(HERE, "ref_pieces.npz"), **out)
    print("ref_pieces.npz written:", {k: v.shape for k, v in out.items()})


def decompose_P(P):
    """Camera::updateOthers (project/camera.cpp:251-288) restated with numpy's QR."""
    P = P / np.sum(P[2, :3] ** 2)
    M = P[:, :3]
    rev = np.array([[0, 0, 1.0], [0, 1, 0], [1, 0, 0]])
    q, r = np.linalg.qr((rev @ M).T)
    Rm = rev @ q.T
    K = rev @ r.T @ rev
    for axis in (2, 1, 0):
        if K[axis, axis] < 0:
            K[axis, axis] = -K[axis, axis]
            Rm[axis] = -Rm[axis]
        if K[axis, 2] < 0:
            K[axis, 2] = -K[axis, 2]
    t = np.linalg.inv(K) @ P[:, 3]
    return K, Rm, t


def make_bunny():
    R = O.ref()
    R.refp_load_scaled.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_int, O.c_uint8_p, O.c_uint8_p,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int)]
    R.refp_load_scaled.restype = C.c_int
    root = ET.parse(os.path.join(REF, "example", "project.xml")).getroot()
    out = {}
    for tag, cam_id in (("left", "7310085"), ("right", "7310087")):
        cam = [c for c in root.find("cameras") if c.get("id") == cam_id][0]
        pm = cam.find("projectionMatrix")
        P = np.array([[float(pm.get("m%d%d" % (i, j))) for j in (1, 2, 3, 4)] for i in (1, 2, 3)])
        ld = cam.find("lensDistortion")
        dist = np.array([float(ld.get(k, "0")) for k in ("k1", "k2", "p1", "p2", "k3")])
        K, Rm, t = decompose_P(P)
        buf = np.zeros((1024 * 768 * 4,), np.uint8); msk = np.zeros((1024 * 768,), np.uint8)
        w, h = C.c_int(0), C.c_int(0)
        ok = R.refp_load_scaled(os.path.join(REF, "example", "images", "bunny", cam_id + ".png").encode(), 0.25,
                                1024, 768, O.u8ptr(buf), O.u8ptr(msk), C.byref(w), C.byref(h))
        assert ok, cam_id
        out[tag + "_rgba"] = buf[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy()
        out[tag + "_mask"] = msk[: w.value * h.value].reshape(h.value, w.value).copy()
        out[tag + "_K"], out[tag + "_R"], out[tag + "_t"], out[tag + "_dist"] = K, Rm, t, dist
    out["scale"] = np.array([0.25])
    np.savez_compressed(os.path.join(HERE, "bunny_pair.npz"), **out)
    print("bunny_pair.npz written:", out["left_rgba"].shape, "mask fraction", out["left_mask"].mean(),
          out["right_mask"].mean())


def make_bunny_views():
    R = O.ref()
    R.refp_load_scaled.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_int, O.c_uint8_p, O.c_uint8_p,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int)]
    R.refp_load_scaled.restype = C.c_int
    root = ET.parse(os.path.join(REF, "example", "project.xml")).getroot()
    iset = [s for s in root.find("imageSets") if s.get("id") == "bunny"][0]
    files = {im.get("for"): im.get("file") for im in iset}
    cams = sorted(root.find("cameras"), key=lambda c: c.get("id"))           # Project::cameras() is a map: id order
    ids, rgba, mask, Ps, dists = [], [], [], [], []
    xml = ['<?xml version="1.0" encoding="UTF-8"?>',
           "<!-- test fixture (data): the eight cameras of the example project and its `bunny` image set; image files are the",
           "     Qt-ingested views of tests/golden/bunny_views.npz written as <id>.raw by the test -->", "<project>", " <cameras>"]
    for cam in cams:
        cid = cam.get("id")
        pm, ld = cam.find("projectionMatrix"), cam.find("lensDistortion")
        P = np.array([[float(pm.get("m%d%d" % (i, j))) for j in (1, 2, 3, 4)] for i in (1, 2, 3)])
        dist = np.array([float(ld.get(k, "0")) for k in ("k1", "k2", "p1", "p2", "k3")])
        buf = np.zeros((1024 * 768 * 4,), np.uint8); msk = np.zeros((1024 * 768,), np.uint8)
        w, h = C.c_int(0), C.c_int(0)
        ok = R.refp_load_scaled(os.path.join(REF, "example", iset.get("root"), files[cid]).encode(), 0.25,
                                1024, 768, O.u8ptr(buf), O.u8ptr(msk), C.byref(w), C.byref(h))
        assert ok, cid
        ids.append(cid); Ps.append(P); dists.append(dist)
        rgba.append(buf[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy())
        mask.append(msk[: w.value * h.value].reshape(h.value, w.value).copy())
        xml.append('  <camera id="%s">' % cid)
        xml.append("   <projectionMatrix %s/>" % " ".join('m%d%d="%s"' % (i, j, pm.get("m%d%d" % (i, j))) for i in (1, 2, 3) for j in (1, 2, 3, 4)))
        xml.append("   <lensDistortion %s/>" % " ".join('%s="%s"' % (k, ld.get(k)) for k in ("k1", "k2", "k3", "p1", "p2") if ld.get(k) is not None))
        xml.append("  </camera>")
    xml += [" </cameras>", " <imageSets>", '  <imageSet root="." id="bunny">']
    xml += ['   <image for="%s" default="yes" file="%s.raw"/>' % (cid, cid) for cid in ids]
    xml += ["  </imageSet>", " </imageSets>", "</project>", ""]
    open(os.path.join(HERE, "bunny_project.xml"), "w").write("\n".join(xml))
    np.savez_compressed(os.path.join(HERE, "bunny_views.npz"), ids=np.array(ids), rgba=np.stack(rgba), mask=np.stack(mask),
                        P=np.stack(Ps), dist=np.stack(dists), scale=np.array([0.25]))
    print("bunny_views.npz written:", np.stack(rgba).shape, "mask fractions", [round(float(m.mean()), 3) for m in mask])
    # the pair fixture is views 0 and 1 of the same ingest
    g = np.load(os.path.join(HERE, "bunny_pair.npz"))
    assert np.array_equal(g["left_rgba"], rgba[ids.index("7310085")]) and np.array_equal(g["right_mask"], mask[ids.index("7310087")])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "views":
        make_bunny_views()
        sys.exit(0)
    make_ref_pieces()
    make_bunny()
    make_bunny_views()
