"""Writes tests/golden/c5_candidate_lists.json: the ORACLE's candidate lists (sro_epipolar_curve, TwoView flavour) of the 64
fence pixels of the C5 rig, both directions, interface normal on the optical axis and tilted -- as lengths + a SHA-256 per
list and one over all of them.

Why: the refractive projection is third-party arithmetic the oracle cannot be pinned on (the reference takes GSL's roots),
and round 5 edited the oracle's root finder in step with the kernels.  This fixture is the fence: any later edit of
oracle/sr_oracle.c that moves one truncated candidate pixel on this rig shows up as a diff of this file, which has to be
regenerated -- and argued -- in a commit of its own (rule).  Run from the repository root:
    python tests/golden/make_c5_fence.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import c5_fence as F          # noqa: E402
import oracle_ffi as O        # noqa: E402


def candidate_lists(normal_name):
    (Kl, Rl, tl), (Kr, Rr, tr), plane, zmin, zmax = F.rig(normal_name)
    cams = [O.camera_set(Kl, Rl, tl, None, *plane), O.camera_set(Kr, Rr, tr, None, *plane)]
    white = O.OImage(np.zeros((F.H, F.W, 4), dtype=np.uint8), np.ones((F.H, F.W), dtype=np.uint8))
    p = O.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=F.D)
    out = {}
    for ref, oth in ((0, 1), (1, 0)):
        for (x, y) in F.fence_pixels():
            pts = O.epipolar_curve(cams[ref], cams[oth], white, p, False, x, y)
            out["%d>%d:%d,%d" % (ref, oth, x, y)] = np.ascontiguousarray(pts, dtype=np.int32)
    return out


def digest(lists):
    rec, allh = {}, hashlib.sha256()
    for k in sorted(lists):
        b = lists[k].tobytes()
        rec[k] = [int(len(lists[k])), hashlib.sha256(b).hexdigest()[:16]]
        allh.update(k.encode()); allh.update(b)
    return dict(lists=rec, sha256=allh.hexdigest(), points=int(sum(len(v) for v in lists.values())))


if __name__ == "__main__":
    doc = dict(rig=dict(width=F.W, height=F.H, levels=F.D, plane_dist=F.PLANE_DIST, ratio=F.RATIO,
                        normals={k: [float(c) for c in v] for k, v in F.NORMALS.items()}),
               made_by="tests/golden/make_c5_fence.py (oracle/sr_oracle.c sro_epipolar_curve)")
    for name in F.NORMALS:
        doc[name] = digest(candidate_lists(name))
        print(name, doc[name]["points"], "points", doc[name]["sha256"][:16])
    with open(os.path.join(HERE, "c5_candidate_lists.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
