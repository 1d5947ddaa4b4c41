"""Deterministic synthetic inputs for the BASELINE.json configurations.

Follows SURVEY.md section 8(d): a 64-bit LCG byte stream, integer-only image
synthesis (box-blurred noise, integer ground-truth disparity, forward warp), and
the rectified / semicircle camera rigs.  Pure numpy; no compute-path code.
"""
import numpy as np

_LCG_A = np.uint64(6364136223846793005)
_LCG_C = np.uint64(1442695040888963407)


def lcg_bytes(seed, n):
    """n bytes of  s <- s*A + C (mod 2^64), byte = s >> 56  (first byte is after one step)."""
    n = int(n)
    out = np.empty(n, dtype=np.uint8)
    s = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    chunk = 1 << 20
    with np.errstate(over="ignore"):
        a_pow = np.empty(chunk, dtype=np.uint64)     # A^(k+1)
        a_pow[0] = _LCG_A
        g = np.empty(chunk, dtype=np.uint64)         # 1 + A + ... + A^k
        g[0] = np.uint64(1)
        # doubling construction keeps this vectorised
        m = 1
        while m < chunk:
            k = min(m, chunk - m)
            a_pow[m:m + k] = a_pow[:k] * a_pow[m - 1]
            g[m:m + k] = g[:k] * a_pow[m - 1] + g[m - 1]
            m += k
        pos = 0
        while pos < n:
            k = min(chunk, n - pos)
            states = a_pow[:k] * s + g[:k] * _LCG_C
            out[pos:pos + k] = (states >> np.uint64(56)).astype(np.uint8)
            s = states[k - 1]
            pos += k
    return out


def box_blur5(img):
    """5x5 integer box blur (sum // 25), edges replicated. img: HxWxC uint8."""
    h, w = img.shape[:2]
    pad = np.pad(img.astype(np.int32), ((2, 2), (2, 2), (0, 0)), mode="edge")
    acc = np.zeros(img.shape, dtype=np.int32)
    for dy in range(5):
        for dx in range(5):
            acc += pad[dy:dy + h, dx:dx + w]
    return (acc // 25).astype(np.uint8)


def noise_image(seed, w, h, blur=True):
    rgb = lcg_bytes(seed, w * h * 3).reshape(h, w, 3)
    return box_blur5(rgb) if blur else rgb


def _tri(u, period):
    """integer triangle wave in [-1024, 1024]"""
    return np.abs(((u % period) * 4096) // period - 2048) - 1024


def gt_disparity(w, h, D, d0):
    """d(x,y) = d0 + ((D-1)*(512 + 400*S(x,y))) >> 10, S separable triangle wave in [-1,1]."""
    tx = _tri(np.arange(w, dtype=np.int64), max(1, w // 2))
    ty = _tri(np.arange(h, dtype=np.int64), max(1, h // 2))
    s1024 = (ty[:, None] * tx[None, :]) >> 10                  # [-1024, 1024]
    return (d0 + (((D - 1) * (512 * 1024 + 400 * s1024)) >> 20)).astype(np.int32)


def rectified_pair(w, h, D, seed, d0=8):
    """Left/right RGBA images (alpha 255), all-ones masks, integer GT disparity."""
    left = noise_image(seed, w, h)
    fill = noise_image(seed + 0x100, w, h)
    disp = gt_disparity(w, h, D, d0)
    right = fill.copy()
    xs = np.arange(w)[None, :].repeat(h, 0)
    ys = np.arange(h)[:, None].repeat(w, 1)
    # paint far (small disparity) to near (large disparity)
    for dv in range(int(disp.min()), int(disp.max()) + 1):
        sel = disp == dv
        if not sel.any():
            continue
        tx = xs[sel] - dv
        ok = tx >= 0
        right[ys[sel][ok], tx[ok]] = left[ys[sel][ok], xs[sel][ok]]

    def rgba(img):
        out = np.empty((h, w, 4), dtype=np.uint8)
        out[..., :3] = img
        out[..., 3] = 255
        return out
    ones = np.ones((h, w), dtype=np.uint8)
    return rgba(left), rgba(right), ones, ones.copy(), disp


def paint_flat_bands(left, right, frac, flat=90, saturated=255):
    """Copies of a pair with `frac` of the area made textureless, in both images at the same place: the top frac/2 of the
    rows a flat gray, the bottom frac/2 saturated -- the regions in which a support window carries no signal (sum2, sum3 ~ 0:
    the certified arithmetic's error bound has no room there).  Returns (left, right, rows painted)."""
    L, R = left.copy(), right.copy()
    h = L.shape[0]
    n = int(round(h * frac / 2.0))
    if n > 0:
        for img in (L, R):
            img[:n, :, :3] = flat
            img[h - n:, :, :3] = saturated
    return L, R, 2 * n


def rectified_cameras(w, h, baseline=1.0):
    """K=[[f,0,W/2],[0,f,H/2],[0,0,1]], f=W, R=I, C_left=0, C_right=(B,0,0) as (K,R,t) triples."""
    f = float(w)
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    R = np.eye(3)
    t_left = np.zeros(3)
    t_right = -R @ np.array([baseline, 0.0, 0.0])
    return (K, R, t_left), (K, R, t_right)


def rectified_depth_range(w, D, d0=8, baseline=1.0):
    f = float(w)
    return f * baseline / (d0 + D - 1), f * baseline / d0


def semicircle_rig(nviews, w, h, radius=10.0, step_deg=22.5, focal=None):
    """Cameras on a semicircle in the XZ plane looking at the origin."""
    f = float(w) if focal is None else float(focal)
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    cams = []
    start = -0.5 * step_deg * (nviews - 1)
    for v in range(nviews):
        a = np.deg2rad(start + v * step_deg)
        C = np.array([radius * np.sin(a), 0.0, -radius * np.cos(a)])
        zc = -C / np.linalg.norm(C)                      # looks at the origin
        yc = np.array([0.0, 1.0, 0.0])
        xc = np.cross(yc, zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R = np.stack([xc, yc, zc], axis=0)               # world -> camera
        t = -R @ C
        cams.append((K.copy(), R, t))
    return cams


def render_sphere_views(cams, w, h, seed, sphere_radius=2.0, tex_size=1024):
    """Ray-cast a textured sphere (LCG noise texture) over a textured back plane.

    Returns (rgba list, mask list, depth list): mask is 1 on the sphere, depth is camera z.
    """
    tex = noise_image(seed, tex_size, tex_size)
    back = noise_image(seed + 0x100, tex_size, tex_size)
    out_rgba, out_mask, out_depth = [], [], []
    ys, xs = np.mgrid[0:h, 0:w]
    for (K, R, t) in cams:
        Kinv = np.linalg.inv(K)
        C = -R.T @ t
        pix = np.stack([xs + 0.5, ys + 0.5, np.ones_like(xs, dtype=np.float64)], axis=-1)
        d_cam = pix @ Kinv.T
        d_w = d_cam @ R                                  # R^T applied to rows
        d_w /= np.linalg.norm(d_w, axis=-1, keepdims=True)
        b = d_w @ C
        c = C @ C - sphere_radius ** 2
        disc = b * b - c
        hit = disc > 0
        tt = -b - np.sqrt(np.where(hit, disc, 0.0))
        hit &= tt > 0
        P = C[None, None, :] + tt[..., None] * d_w
        # texture coordinates from the hit point (integer lattice lookup)
        u = ((np.arctan2(P[..., 0], -P[..., 2]) / (2 * np.pi) + 0.5) * 4 * tex_size).astype(np.int64) % tex_size
        v = ((np.clip(P[..., 1] / sphere_radius, -1, 1) * 0.5 + 0.5) * 2 * (tex_size - 1)).astype(np.int64) % tex_size
        img = np.empty((h, w, 4), dtype=np.uint8)
        img[..., 3] = 255
        # back plane z = +sphere_radius*2 in world, fronto-parallel texture
        tp = (2.0 * sphere_radius - C[2]) / np.where(np.abs(d_w[..., 2]) > 1e-12, d_w[..., 2], 1e-12)
        Pb = C[None, None, :] + tp[..., None] * d_w
        ub = (Pb[..., 0] * 40).astype(np.int64) % tex_size
        vb = (Pb[..., 1] * 40).astype(np.int64) % tex_size
        img[..., :3] = back[vb, ub]
        img[hit, :3] = tex[v[hit], u[hit]]
        zc = (P - C) @ R[2]
        out_rgba.append(img)
        out_mask.append(hit.astype(np.uint8))
        out_depth.append(np.where(hit, zc, np.nan))
    return out_rgba, out_mask, out_depth
