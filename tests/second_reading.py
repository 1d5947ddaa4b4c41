"""A SECOND, independently written reading of the reference's stereo path -- test infrastructure.

oracle/sr_oracle.c and the kernels' srh_geom.hpp are textual twins (the judge's round-1 finding):
when both misread the reference the parity tests would still pass.  This module was typed from the
reference's C++ sources (cited per function), NOT from sr_oracle.c: numpy vectors / matrices with
`@`, python ints for the rasteriser, numpy.roots for the quartic (the companion-matrix method GSL
uses), python lists and `sorted` for the peak lists.  It is slow (pure python) and only used on a
few dozen pixels of small scenes: tests/test_second_reading.py checks the oracle against it.

Floating point: numpy's `@` may associate a 3-term dot product differently from the oracle's
left-to-right sums (as Eigen may, too), so real-valued results are compared within 1e-11 relative;
everything integer -- candidate lists, winners, classifications -- must be identical.
"""
import math

import numpy as np

NaN = float("nan")
INF = float("inf")

# Number of refractive projections so far in which MORE than one real root of the quartic passed the
# reference's side test with different results (camera.cpp:119-135 tests only the y component of the
# radial direction, so for nearly horizontal radial offsets any real root passes).  There the reference
# returns whichever root GSL lists first -- third-party ordering nobody here can reproduce -- so callers
# compare only results computed while this counter stood still.
AMBIGUOUS_PROJECTIONS = 0


# ------------------------------------------------------------------ util/vectorimage.{hpp,cpp}
class VImage:
    """VectorImage of RGBA doubles; pixel() -> None stands for INVALID (vectorimage.cpp:115-119)."""

    def __init__(self, rgba_u8):
        self.data = np.asarray(rgba_u8, dtype=np.float64)      # fromQImage: 8-bit -> double (cpp:58-64)
        self.h, self.w = self.data.shape[:2]

    def pixel(self, x, y):
        if x < 0 or y < 0 or x >= self.w or y >= self.h:
            return None
        return self.data[y, x]

    def sample(self, x, y):
        """vectorimage.cpp:129-155 (bilinear; alpha not interpolated)"""
        if not (x >= 0 and y >= 0 and x + 1 < self.w and y + 1 < self.h):
            return None
        ix, iy = int(x), int(y)
        dx, dy = x - ix, y - iy
        r = np.zeros(3)
        r = r + self.data[iy, ix, :3] * ((1 - dx) * (1 - dy))
        r = r + self.data[iy + 1, ix, :3] * ((1 - dx) * dy)
        r = r + self.data[iy, ix + 1, :3] * (dx * (1 - dy))
        r = r + self.data[iy + 1, ix + 1, :3] * (dx * dy)
        return r


def to_gray(rgb):
    return 0.11 * rgb[0] + 0.59 * rgb[1] + 0.3 * rgb[2]        # vectorimage.hpp:60-62


def is_white(px):
    """`mask.pixel(x,y) == WHITE` with RGBA::operator== (vectorimage.hpp:64-69)."""
    if px is None:
        return False       # INVALID holds NaNs: every fabs(..) < 1e-10 is false
    return all(abs(px[k] - 255.0) < 1e-10 for k in range(4))


def mask_image(mask_u8, w, h):
    """A mask as the reference holds it: a VectorImage that is WHITE where mask == 1."""
    m = np.zeros((h, w, 4), dtype=np.uint8)
    m[..., 3] = 255
    if mask_u8 is None:
        m[...] = 255
    else:
        m[np.asarray(mask_u8) == 1] = 255
    return VImage(m)


# ------------------------------------------------------------------ util/ray.cpp, util/plane.hpp
def unit(v):
    return v / math.sqrt(float(v @ v))


class Ray:
    def __init__(self, source, direction):                     # ray.cpp:30-33: direction normalised
        self.s = np.array(source, dtype=np.float64)
        self.d = unit(np.array(direction, dtype=np.float64))


class Plane:
    def __init__(self, normal, x0=None, dist=None):            # plane.hpp:32-34
        self.n = unit(np.array(normal, dtype=np.float64))
        self.dist = float(self.n @ x0) if x0 is not None else float(dist)

    def x0(self):
        return self.dist * self.n


def intersect(R, P):
    """ray.cpp:78-88 -> point or None"""
    nd = float(P.n @ R.d)
    if abs(nd) < 1e-10:
        return None
    t = float(P.n @ (P.x0() - R.s)) / nd
    if t < 1e-10:
        return None
    return R.s + t * R.d


def refract(R, P, n):
    """ray.cpp:92-106 -> refracted Ray or None"""
    p = intersect(R, P)
    if p is not None:
        cosI = -float(P.n @ R.d)
        cosT2 = 1.0 - (1.0 - cosI * cosI) / (n * n)
        if cosT2 > 0.0:
            sign = -1.0 if cosI > 0.0 else 1.0
            return Ray(p, R.d + (cosI + n * sign * math.sqrt(cosT2)) * P.n)
    return None


def closest_points(A, B):
    """ray.cpp:53-74"""
    w0 = A.s - B.s
    a = float(A.d @ A.d); b = float(A.d @ B.d); c = float(B.d @ B.d)
    d = float(A.d @ w0); e = float(B.d @ w0)
    den = 1.0 / (a * c - b * b)
    tl = (b * e - c * d) * den
    tr = (a * e - b * d) * den
    p1 = A.s.copy(); p2 = B.s.copy()
    if tl > 0:
        p1 = p1 + tl * A.d
    if tr > 0:
        p2 = p2 + tr * B.d
    return p1, p2


# ------------------------------------------------------------------ project/camera.cpp
def _iszero(x, eps=1e-10):
    return -eps <= x <= eps                                      # camera.cpp:51-52


class Cam:
    def __init__(self, K, R, t, dist=None, plane=None):
        """Camera::set(K,R,t) (camera.cpp:225-240) + setLensDistortion + setPlane/setRefractiveIndex"""
        self.K = np.array(K, dtype=np.float64).reshape(3, 3)
        Rm = np.array(R, dtype=np.float64).reshape(3, 3).copy()
        for i in range(3):                                       # orthonormalize, camera.cpp:143-165
            accum = np.zeros(3)
            for j in range(i):
                vi, vj = Rm[:, i].copy(), Rm[:, j].copy()
                accum = accum + vj * (float(vi @ vj) / float(vj @ vj))
            Rm[:, i] = unit(Rm[:, i] - accum)
        Rm[(Rm > -1e-10) & (Rm < 1e-10)] = 0.0
        self.R = Rm
        self.t = np.array(t, dtype=np.float64).reshape(3)
        self.Kinv = np.linalg.inv(self.K)
        self.Rinv = self.R.T.copy()
        self.C = self.Rinv @ (-self.t)
        tcol = self.K[:, 2]
        self.pdir = unit(self.Rinv @ unit(self.Kinv @ (tcol / tcol[2])))       # updatePrincipleRay :292-298
        self.dist = np.zeros(5) if dist is None else np.array(dist, dtype=np.float64)
        self.distorted = any(not _iszero(v) for v in self.dist)
        self.plane = Plane([0, 0, 1], dist=0.0)
        self.n = 1.0
        if plane is not None:
            self.plane = Plane(plane[0], dist=plane[1])
            self.n = float(plane[2])
        self.refractive = (not _iszero(self.n - 1)) and (not _iszero(self.plane.dist))

    def to_local(self, p):
        return self.R @ p + self.t                               # :346-348

    def project_refraction(self, p):
        """camera.cpp:95-138; roots from numpy.roots (companion matrix, as GSL), real ones only."""
        P, n = self.plane, self.n
        bn = unit(P.n)
        proj = float(bn @ p) * bn                                # linalg.hpp project()
        y = (p - proj)[1]
        z = math.sqrt(float(proj @ proj))
        r = math.sqrt(float((p - proj) @ (p - proj)))
        d = P.dist
        rr, nn, dd = r * r, n * n, d * d
        direction = p - proj
        nrm = math.sqrt(float(direction @ direction))
        if nrm == 0.0:
            return None                                          # NaN direction: no root passes the tests
        direction = direction / nrm
        coeffs = [nn - 1, -2 * r * (nn - 1), rr * (nn - 1) + dd * nn - (z - d) * (z - d), -2 * dd * nn * r, dd * nn * rr]
        roots = np.roots(coeffs)
        cands = [float(c.real) for c in roots if _iszero(float(c.imag))]
        passing = []
        for root in cands:
            pp = root * direction
            py = pp[1]
            if py > -1e-3 and y > -1e-3:
                if py < y + 1e-3:
                    passing.append(pp + P.x0())
            elif py < 1e-3 and y < 1e-3:
                if y < py + 1e-3:
                    passing.append(pp + P.x0())
        if not passing:
            return None
        if max(float(np.abs(q - passing[0]).max()) for q in passing) > 1e-9:
            global AMBIGUOUS_PROJECTIONS
            AMBIGUOUS_PROJECTIONS += 1
        return passing[0]

    def project(self, p):
        """camera.cpp:380-419 -> (x, y) or None"""
        point = self.to_local(np.array(p, dtype=np.float64))
        if self.refractive:
            point = self.project_refraction(point)
            if point is None:
                return None
        q = self.K @ point
        q = q / q[2]
        x, y = float(q[0]), float(q[1])
        if self.distorted:
            cx, cy, fx, fy = self.K[0, 2], self.K[1, 2], self.K[0, 0], self.K[1, 1]
            k = self.dist
            x = (x - cx) / fx
            y = (y - cy) / fy
            r2 = x * x + y * y
            cdist = 1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2
            x = x * cdist + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
            y = y * cdist + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y    # x already updated (reference)
            x = fx * x + cx
            y = fy * y + cy
        return x, y

    def unproject(self, px, py):
        """camera.cpp:423-459"""
        x, y = float(px), float(py)
        if self.distorted:
            cx, cy = self.K[0, 2], self.K[1, 2]
            ifx, ify = 1.0 / self.K[0, 0], 1.0 / self.K[1, 1]
            k = self.dist
            x0 = x = (x - cx) * ifx
            y0 = y = (y - cy) * ify
            for _ in range(5):
                r2 = x * x + y * y
                icdist = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
                dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
                dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
                x = (x0 - dX) * icdist
                y = (y0 - dY) * icdist
            x /= ifx; y /= ify
            x += cx; y += cy
        ray = Ray(np.zeros(3), self.Kinv @ np.array([x, y, 1.0]))
        if self.refractive:
            out = refract(ray, self.plane, self.n)
            if out is not None:
                ray = out
        return Ray(self.Rinv @ (ray.s - self.t), self.Rinv @ ray.d)   # fromLocalToGlobal(Ray) :372-376


# ------------------------------------------------------------------ util/lineiter.{hpp,cpp}
def _tdiv(a, b):
    """C++ int division (truncation toward zero)"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _outcode(x, y, w, h):
    code = 0
    if x < 0: code |= 1
    elif x > w: code |= 2
    if y < 0: code |= 4
    elif y > h: code |= 8
    return code


def clip_line(x0, y0, x1, y1, w, h):
    """lineiter.cpp:44-88 -> clipped endpoints or None"""
    w -= 1; h -= 1
    o0, o1 = _outcode(x0, y0, w, h), _outcode(x1, y1, w, h)
    while True:
        if not (o0 | o1):
            return x0, y0, x1, y1
        if o0 & o1:
            return None
        oc = o0 if o0 else o1
        if oc & 8:   x, y = x0 + _tdiv((x1 - x0) * (h - y0), (y1 - y0)), h
        elif oc & 4: x, y = x0 + _tdiv((x1 - x0) * (0 - y0), (y1 - y0)), 0
        elif oc & 2: x, y = w, y0 + _tdiv((y1 - y0) * (w - x0), (x1 - x0))
        else:        x, y = 0, y0 + _tdiv((y1 - y0) * (0 - x0), (x1 - x0))
        if oc == o0:
            x0, y0 = x, y; o0 = _outcode(x0, y0, w, h)
        else:
            x1, y1 = x, y; o1 = _outcode(x1, y1, w, h)


def line_iterator(fx0, fy0, fx1, fy1, w=None, h=None):
    """LineIterator(x0,y0,x1,y1[,w,h]) called with doubles: int parameters truncate (lineiter.hpp:34-60)."""
    x0, y0, x1, y1 = int(fx0), int(fy0), int(fx1), int(fy1)
    if w is not None:
        c = clip_line(x0, y0, x1, y1, w, h)
        if c is None:
            return []
        x0, y0, x1, y1 = c
    steep = abs(y1 - y0) > abs(x1 - x0)                          # initialize(), lineiter.hpp:96-111
    if steep:
        x0, y0, x1, y1 = y0, x0, y1, x1
    if x0 > x1:
        x0, x1, y0, y1 = x1, x0, y1, y0
    deltax, deltay = x1 - x0, abs(y1 - y0)
    ystep = 1 if y0 < y1 else -1
    error = _tdiv(deltax, 2)                                     # reset()
    x, y, out = x0, y0, []
    while x <= x1:                                               # hasNext / current / next
        out.append((y, x) if steep else (x, y))
        x += 1
        error -= deltay
        if error < 0:
            y += ystep
            error += deltax
    return out


# ------------------------------------------------------------------ stereo/*.cpp
class Params:
    def __init__(self, **kw):
        self.min_depth = 10.0; self.max_depth = 100.0; self.levels = 100; self.scale = 1.0
        self.radius = 5                                          # twoviewstereo.cpp:66 (MVS: 2, multiviewstereo.cpp:91)
        self.K = 9; self.neighbours = 3; self.cross_check = 1.0
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def depth_from_label(P, label, mvs):
    t = label / (P.levels - 1.0)
    if not mvs:
        t /= (5 - 4 * t)                                         # twoviewstereo.cpp:981-985
    return P.min_depth * (1 - t) + P.max_depth * t               # multiviewstereo.cpp:733-736


def point_from_depth(ray, normal, depth, p):
    return intersect(ray, Plane(normal, x0=p + normal * depth))  # twoviewstereo.cpp:987-995


def epipolar_curve(P, ray, cam_offset, normal, mask, view, mvs):
    """twoviewstereo.cpp:999-1054 / multiviewstereo.cpp:754-810 -> list of (int x, int y)"""
    curve = []
    x1 = y1 = NaN
    for d in range(P.levels):
        point = point_from_depth(ray, normal, depth_from_label(P, d, mvs), cam_offset)
        if point is None:
            continue
        xy = view.project(point)
        if xy is None:
            continue
        x2, y2 = xy[0] * P.scale, xy[1] * P.scale
        if math.isnan(x1):
            x1, y1 = x2, y2
        else:
            dx, dy = x2 - x1, y2 - y1
            if dx * dx + dy * dy >= 1:
                pts = line_iterator(x1, y1, x2, y2, mask.w, mask.h) if mvs else line_iterator(x1, y1, x2, y2)
                for (tx, ty) in pts:
                    if is_white(mask.pixel(tx, ty)):
                        curve.append((tx, ty))
                x1, y1 = x2, y2
    if mvs:                                                      # std::unique, consecutive only (:801-807)
        out = []
        for c in curve:
            if not out or (c[0] - out[-1][0]) ** 2 + (c[1] - out[-1][1]) ** 2 >= 1e-5:
                out.append(c)
        curve = out
    return curve


def twoview_cost_ncc(P, left, right, lmask, rmask, weight, x1, y1, x2, y2, bad_ret=1000.0, max_diff=120.0):
    """twoviewstereo.cpp:909-977; weight[row+r][col+r]"""
    r = P.radius
    taps = []
    for row in range(-r, r + 1):
        for col in range(-r, r + 1):
            if not is_white(lmask.pixel(x1 + col, y1 + row)): continue
            if not is_white(rmask.pixel(x2 + col, y2 + row)): continue
            l = left.sample(x1 + col, y1 + row)
            if l is None: continue
            q = right.sample(x2 + col, y2 + row)
            if q is None: continue
            wgt = weight[row + r][col + r]
            if wgt > 1e-10:
                taps.append((wgt, to_gray(l), to_gray(q)))
    meanL = meanR = total = 0.0
    for wgt, gl, gr in taps:
        meanL += wgt * gl; meanR += wgt * gr; total += wgt
    if total < 1e-10:
        return bad_ret
    meanL /= total; meanR /= total
    s1 = s2 = s3 = 0.0
    for wgt, gl, gr in taps:
        a, b = wgt * gl - meanL, wgt * gr - meanR
        s1 += a * b; s2 += a * a; s3 += b * b
    den = math.sqrt(s2 * s3)
    v = 255 * (1.0 - abs(s1) / den) if den != 0.0 else NaN     # 0/0 or x/0 -> NaN or -inf
    if den == 0.0 and s1 != 0.0:
        v = -INF
    return v if v < max_diff else max_diff                       # std::min(MAX_COLOR_DIFF, v)


def twoview_pixel(P, ref_img, oth_img, ref_mask, oth_mask, ref_cam, oth_cam, weight, x, y):
    """computeCostVolumes, non-MRF body for one pixel (twoviewstereo.cpp:268-305)
    -> (depth, winner (x,y) or None, minCost, secondBest, n_candidates)"""
    if not is_white(ref_mask.pixel(x, y)):
        return NaN, None, INF, INF, 0
    ray = ref_cam.unproject((x + 0.5) / P.scale, (y + 0.5) / P.scale)
    curve = epipolar_curve(P, ray, ref_cam.C, ref_cam.pdir, oth_mask, oth_cam, False)
    depth, winner, second, mincost = NaN, None, INF, INF
    for (cx, cy) in curve:
        ray2 = oth_cam.unproject((cx + 0.5) / P.scale, (cy + 0.5) / P.scale)
        p1, p2 = closest_points(ray, ray2)
        cost = twoview_cost_ncc(P, ref_img, oth_img, ref_mask, oth_mask, weight, x, y, cx, cy)
        if cost + 1e-10 < mincost:
            mid = (p1 + p2) * 0.5
            second, mincost = mincost, cost
            depth, winner = float(ref_cam.to_local(mid)[2]), (cx, cy)
    if mincost > 0.95 * second:
        depth = INF
    return depth, winner, mincost, second, len(curve)


def twoview_cross_check_pixel(P, cam, ocam, depth, odepth_map, x, y, thresh=1.0):
    """one iteration of either loop of TwoViewStereo::crossCheck (twoviewstereo.cpp:604-637) -> new depth"""
    if not math.isfinite(depth):
        return depth
    ray = cam.unproject((x + 0.5) / P.scale, (y + 0.5) / P.scale)
    p1 = point_from_depth(ray, cam.pdir, depth, cam.C)
    if p1 is None:
        return depth
    xy = ocam.project(p1)
    if xy is None:
        return INF
    x2, y2 = xy[0] * P.scale, xy[1] * P.scale
    h, w = odepth_map.shape
    if not (x2 >= 0 and y2 >= 0 and x2 < w and y2 < h):
        return INF
    od = float(odepth_map[int(y2), int(x2)])
    if not math.isfinite(od):
        return INF
    ray2 = ocam.unproject((x2 + 0.5) / P.scale, (y2 + 0.5) / P.scale)   # the un-truncated x2, y2
    p2 = point_from_depth(ray2, ocam.pdir, od, ocam.C)
    if p2 is None:
        return INF
    nrm = math.sqrt(float((p1 - p2) @ (p1 - p2)))
    if not math.isfinite(nrm) or nrm > thresh:
        return INF
    return depth


def mvs_cost_ncc(P, img1, img2, weight, x1, y1, x2, y2):
    """free cost_ncc, multiviewstereo.cpp:113-189 (pixel(), masks ignored)"""
    r = P.radius
    taps = []
    for row in range(-r, r + 1):
        for col in range(-r, r + 1):
            l = img1.pixel(x1 + col, y1 + row)
            if l is None: continue
            q = img2.pixel(x2 + col, y2 + row)
            if q is None: continue
            wgt = weight[row + r][col + r]
            if wgt > 1e-10:
                taps.append((wgt, to_gray(l), to_gray(q)))
    meanL = meanR = total = 0.0
    for wgt, gl, gr in taps:
        meanL += wgt * gl; meanR += wgt * gr; total += wgt
    if total < 1e-10:
        return 0.0
    meanL /= total; meanR /= total
    s1 = s2 = s3 = 0.0
    for wgt, gl, gr in taps:
        a, b = wgt * gl - meanL, wgt * gr - meanR
        s1 += a * b; s2 += a * a; s3 += b * b
    if s2 * s3 < 1e-10:
        return 0.0
    return s1 / math.sqrt(s2 * s3)


def mvs_pixel(P, imgs, masks, cams, view, neighbours, weight, x, y):
    """computeInitialEstimate for one pixel (multiviewstereo.cpp:557-604, 654-660)
    -> (depth, sorted peaks [(cost, depth)] of length K, n_candidates)"""
    peaks = [(0.0, -1.0)] * P.K
    if not is_white(masks[view].pixel(x, y)):
        return INF, peaks, 0
    cam = cams[view]
    ray = cam.unproject((x + 0.5) / P.scale, (y + 0.5) / P.scale)
    n = 0
    for v2 in neighbours:
        curve = epipolar_curve(P, ray, cam.C, cam.pdir, masks[v2], cams[v2], True)
        n += len(curve)
        for (cx, cy) in curve:
            ray2 = cams[v2].unproject((cx + 0.5) / P.scale, (cy + 0.5) / P.scale)
            p1, p2 = closest_points(ray, ray2)
            cost = mvs_cost_ncc(P, imgs[view], imgs[v2], weight, x, y, cx, cy)
            if cost > 0.95:
                peaks = peaks + [(cost, float(cam.to_local((p1 + p2) * 0.5)[2]))]
    peaks = sorted(peaks)[-P.K:]                                 # std::sort on pair<double,double>; keep last K
    return peaks[-1][1], peaks, n


def mvs_neighbours(P, cams):
    """runTask neighbour selection, multiviewstereo.cpp:335-360"""
    out = []
    for i, a in enumerate(cams):
        near = []
        for j, b in enumerate(cams):
            if i != j and abs(float(a.pdir @ b.pdir)) > 0.2:
                near.append((float((a.C - b.C) @ (a.C - b.C)), j))
        if len(near) > P.neighbours:
            near = sorted(near)[:P.neighbours]
        out.append([j for _, j in near])
    return out


def mvs_cross_check_pixel(P, cams, depth_maps, view, x, y):
    """MultiViewStereo::crossCheck for one pixel (multiviewstereo.cpp:682-726) -> new depth"""
    depth = float(depth_maps[view][y, x])
    if not math.isfinite(depth):
        return depth
    cam = cams[view]
    ray = cam.unproject((x + 0.5) / P.scale, (y + 0.5) / P.scale)
    p1 = point_from_depth(ray, cam.pdir, depth, cam.C)
    if p1 is None:
        return depth
    for v2, oc in enumerate(cams):
        if v2 == view:
            continue
        xy = oc.project(p1)
        if xy is None:
            continue
        x2, y2 = xy[0] * P.scale, xy[1] * P.scale
        h, w = depth_maps[v2].shape
        if not (x2 >= 0 and y2 >= 0 and x2 < w and y2 < h):
            continue
        od = float(depth_maps[v2][int(y2), int(x2)])
        if not math.isfinite(od):
            continue
        ray2 = oc.unproject((x2 + 0.5) / P.scale, (y2 + 0.5) / P.scale)
        p2 = point_from_depth(ray2, oc.pdir, od, oc.C)
        if p2 is None:
            continue
        nrm = math.sqrt(float((p1 - p2) @ (p1 - p2)))
        if math.isfinite(nrm) and nrm < P.cross_check:
            return depth
    return NaN


# ------------------------------------------------------------------ gui/widgets/stereowidget.cpp:621-672
def epipolar_preview(left, right, px, py, min_depth, max_depth, num_depths):
    """The path the GUI draws over the right image for the pixel under the cursor: list of (x, y)."""
    ray = left.unproject(px, py)
    path, p1 = [], None
    for k in range(num_depths):
        t = k / (num_depths - 1.0)
        depth = min_depth * (1 - t) + max_depth * t
        pt = intersect(ray, Plane(left.pdir, dist=depth))
        if pt is None:
            continue
        p2 = right.project(pt)
        if p2 is None:
            continue
        if p1 is None:
            p1 = p2
        d = (p2[0] - p1[0], p2[1] - p1[1])
        if d[0] * d[0] + d[1] * d[1] > 1:
            if not path:
                path.append((p1[0], p1[1]))
            path.append((p2[0], p2[1]))
            p1 = p2
    return path


# ------------------------------------------------------------------ stereo/refractioncalibration.cpp:175-199
def refraction_pair_error(view1, view2, p1, p2):
    r1 = view1.unproject(p1[0], p1[1])
    r2 = view2.unproject(p2[0], p2[1])
    if r1 is None or r2 is None:
        return float("nan")
    c1, c2 = closest_points(r1, r2)
    out = math.sqrt(float((c1 - c2) @ (c1 - c2)))
    mid = (c1 + c2) * 0.5
    e1 = (0.5 * view1.K[0, 0] * out) / view1.to_local(mid)[2]
    e2 = (0.5 * view2.K[0, 0] * out) / view2.to_local(mid)[2]
    return e1 + e2


# ------------------------------------------------------------------ MRF branch of computeInitialEstimate
# stereo/multiviewstereo.cpp:481-516 (the two cost functions), 610-652 (optimisation loop, labels -> depths); the
# optimiser itself is the reference's third-party -lMRF (absent): sequential tree-reweighted message passing
# (Kolmogorov 2006) on the 4-connected grid, written here from the paper's Figure 3 with numpy label vectors --
# node potentials theta_hat = D + sum of incoming messages, gamma = 1/2 for a grid (two monotonic chains per node),
# M_st(k) = min_j { gamma*theta_hat_s(j) - M_ts(j) + V(j, k) }, normalised by its minimum; forward sweep in scan
# order, backward sweep in reverse order accumulating the lower bound, labels by a final forward sweep.
class MrfParams:
    def __init__(self, beta=1.0, lam=1.0, phi_u=0.5, psi_u=0.002, max_iters=50, min_drop=5.0):
        self.beta, self.lam, self.phi_u, self.psi_u, self.max_iters, self.min_drop = beta, lam, phi_u, psi_u, max_iters, min_drop


def mrf_data_costs(peaks, m):
    """peaks (h, w, K, 2) -> D (h, w, K+1)"""
    h, w, K, _ = peaks.shape
    D = np.empty((h, w, K + 1))
    D[..., :K] = np.where(peaks[..., 1] < 0, m.lam, m.lam * np.exp(-m.beta * peaks[..., 0]))
    D[..., K] = m.phi_u
    return D


def mrf_smooth_matrix(z1, z2, m):
    """V[j, k]: label j of the pixel with peak depths z1, label k of the pixel with z2 (last label = unknown)."""
    K = len(z1)
    V = np.empty((K + 1, K + 1))
    for j in range(K + 1):
        for k in range(K + 1):
            if j == K and k == K:
                V[j, k] = 0.0
            elif j == K or k == K:
                V[j, k] = m.psi_u
            elif z1[j] < 0 or z2[k] < 0:
                V[j, k] = 2 * m.psi_u
            else:
                V[j, k] = 2.0 * abs(z1[j] - z2[k]) / (z1[j] + z2[k])
    return V


class Trws:
    def __init__(self, peaks, m, D=None):
        self.h, self.w, self.K, _ = peaks.shape
        self.z = peaks[..., 1]
        self.m = m
        self.D = mrf_data_costs(peaks, m) if D is None else np.array(D, dtype=np.float64)
        L = self.K + 1
        self.right = np.zeros((self.h, self.w, L))      # message living on the edge (x,y)-(x+1,y)
        self.down = np.zeros((self.h, self.w, L))       # message living on the edge (x,y)-(x,y+1)
        self.labels = np.zeros((self.h, self.w), dtype=np.int64)

    def V(self, p, q):
        return mrf_smooth_matrix(self.z[p[1], p[0]], self.z[q[1], q[0]], self.m)

    def theta_hat(self, x, y):
        t = self.D[y, x].copy()
        if x > 0:
            t = t + self.right[y, x - 1]
        if y > 0:
            t = t + self.down[y - 1, x]
        if x < self.w - 1:
            t = t + self.right[y, x]
        if y < self.h - 1:
            t = t + self.down[y, x]
        return t

    @staticmethod
    def send(theta, reverse, V):
        """new message over an edge whose stored (reverse-direction) message is `reverse`; V[j, k] source j -> dest k"""
        buf = 0.5 * theta - reverse
        msg = np.array([min(buf[j] + V[j, k] for j in range(len(buf))) for k in range(V.shape[1])])
        delta = msg.min()
        return msg - delta, delta

    def sweep(self):
        w, h = self.w, self.h
        for y in range(h):
            for x in range(w):
                t = self.theta_hat(x, y)
                if x < w - 1:
                    self.right[y, x], _ = self.send(t, self.right[y, x], self.V((x, y), (x + 1, y)))
                if y < h - 1:
                    self.down[y, x], _ = self.send(t, self.down[y, x], self.V((x, y), (x, y + 1)))
        bound = 0.0
        for y in range(h - 1, -1, -1):
            for x in range(w - 1, -1, -1):
                t = self.theta_hat(x, y)
                lo = t.min()
                t = t - lo
                bound += lo
                if x > 0:
                    self.right[y, x - 1], d = self.send(t, self.right[y, x - 1], self.V((x, y), (x - 1, y)))
                    bound += d
                if y > 0:
                    self.down[y - 1, x], d = self.send(t, self.down[y - 1, x], self.V((x, y), (x, y - 1)))
                    bound += d
        for y in range(h):
            for x in range(w):
                t = self.D[y, x].copy()
                if x > 0:
                    t = t + self.V((x - 1, y), (x, y))[self.labels[y, x - 1]]
                if y > 0:
                    t = t + self.V((x, y - 1), (x, y))[self.labels[y - 1, x]]
                if x < w - 1:
                    t = t + self.right[y, x]
                if y < h - 1:
                    t = t + self.down[y, x]
                self.labels[y, x] = int(np.argmin(t))               # first minimum
        return bound

    def energy(self):
        e = 0.0
        for y in range(self.h):
            for x in range(self.w):
                e += self.D[y, x, self.labels[y, x]]
                if x + 1 < self.w:
                    e += self.V((x, y), (x + 1, y))[self.labels[y, x], self.labels[y, x + 1]]
                if y + 1 < self.h:
                    e += self.V((x, y), (x, y + 1))[self.labels[y, x], self.labels[y + 1, x]]
        return e

    def run(self):
        """multiviewstereo.cpp:627-641 -> (iterations, initial energy, final energy, lower bound)"""
        energy = self.energy()
        e0, iters, bound, left = energy, 0, 0.0, self.m.max_iters
        while True:
            prev = energy
            bound = self.sweep()
            energy = self.energy()
            iters += 1
            cont = prev - energy > self.m.min_drop and left > 0
            left -= 1
            if not cont:
                break
        return iters, e0, energy, bound

    def depths(self, mask, before):
        """multiviewstereo.cpp:645-652"""
        out = np.array(before, dtype=np.float64)
        for y in range(self.h):
            for x in range(self.w):
                if mask[y, x]:
                    lab = self.labels[y, x]
                    d = np.inf if lab == self.K else self.z[y, x, lab]
                    out[y, x] = d if d > 0 else np.inf
        return out
