"""Synthetic top-K peak buffers of the kind MultiViewStereo::computeInitialEstimate collects
(multiviewstereo.cpp:549-600): per pixel K (cost, depth) pairs sorted ascending, (0, -1) where fewer than K
candidates passed the 0.95 threshold, all (0, -1) outside the mask."""
import numpy as np


def peaks_case(w, h, K=9, seed=1, fill=0.7, mask_frac=0.85, surface=True):
    rng = np.random.default_rng(seed)
    peaks = np.zeros((h, w, K, 2))
    peaks[..., 1] = -1.0
    yy, xx = np.mgrid[0:h, 0:w]
    zsurf = 2.0 + 0.3 * np.sin(xx / 7.0) + 0.2 * np.cos(yy / 5.0) if surface else np.full((h, w), 2.0)
    mask = (rng.uniform(size=(h, w)) < mask_frac).astype(np.uint8)
    if mask_frac >= 1.0:
        mask[:] = 1
    for y in range(h):
        for x in range(w):
            if not mask[y, x]:
                continue
            n = int(rng.binomial(K + 3, fill))
            if n == 0:
                continue
            cost = 0.95 + 0.05 * rng.uniform(size=n)
            z = np.where(rng.uniform(size=n) < 0.5, zsurf[y, x] * (1 + 0.01 * rng.normal(size=n)),
                         rng.uniform(1.0, 4.0, size=n))
            pairs = sorted([(0.0, -1.0)] * K + list(zip(cost.tolist(), z.tolist())))[-K:]   # std::sort, keep the last K
            peaks[y, x] = np.array(pairs)
    return peaks, mask


def peaks_case_fast(w, h, K=9, seed=1, fill=0.7, mask_frac=0.85):
    """The same kind of buffer, vectorised (for sizes where the per-pixel loop above is too slow)."""
    rng = np.random.default_rng(seed)
    C = K + 3
    yy, xx = np.mgrid[0:h, 0:w]
    zsurf = 2.0 + 0.3 * np.sin(xx / 7.0) + 0.2 * np.cos(yy / 5.0)
    mask = (rng.uniform(size=(h, w)) < mask_frac).astype(np.uint8)
    present = (rng.uniform(size=(h, w, C)) < fill) & (mask[..., None] == 1)
    cost = np.where(present, 0.95 + 0.05 * rng.uniform(size=(h, w, C)), 0.0)
    near = rng.uniform(size=(h, w, C)) < 0.5
    z = np.where(near, zsurf[..., None] * (1 + 0.01 * rng.normal(size=(h, w, C))), rng.uniform(1.0, 4.0, size=(h, w, C)))
    z = np.where(present, z, -1.0)
    # pad with K placeholders, sort ascending by (cost, depth), keep the last K
    cost = np.concatenate([np.zeros((h, w, K)), cost], axis=-1)
    z = np.concatenate([-np.ones((h, w, K)), z], axis=-1)
    order = np.lexsort((z, cost), axis=-1)
    cost = np.take_along_axis(cost, order, axis=-1)[..., -K:]
    z = np.take_along_axis(z, order, axis=-1)[..., -K:]
    return np.ascontiguousarray(np.stack([cost, z], axis=-1)), mask
