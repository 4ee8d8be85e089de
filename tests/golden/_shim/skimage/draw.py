"""Stand-in for the parts of scikit-image 0.18.1 `skimage.draw` the reference calls (the package is absent from the build
container; test infrastructure only, used by tests/golden/make_golden.py).  `polygon` and `line` restate the published
algorithms of that version (`_draw.pyx: _polygon`, `_line`; `_shared/geometry.pyx: point_in_polygon`, the even-odd
crossing test); `polygon_perimeter` follows `draw.py: polygon_perimeter` and, like it, clips with matplotlib's Path
(`_shared/_geometry.py: polygon_clip`), which IS installed here."""
import sys

import numpy as np

draw = sys.modules[__name__]          # `from skimage.draw import draw` (classics.py:5)


def disk(center, radius, shape=None):
    r0, c0 = center
    lo_r, hi_r = int(np.floor(r0 - radius)), int(np.ceil(r0 + radius))
    lo_c, hi_c = int(np.floor(c0 - radius)), int(np.ceil(c0 + radius))
    rr, cc = np.mgrid[lo_r:hi_r + 1, lo_c:hi_c + 1]
    keep = (rr - r0) ** 2 + (cc - c0) ** 2 < radius ** 2
    if shape is not None:
        keep &= (rr >= 0) & (rr < shape[0]) & (cc >= 0) & (cc < shape[1])
    return rr[keep], cc[keep]


def _point_in_polygon(xp, yp, x, y):
    c = False
    j = len(xp) - 1
    for i in range(len(xp)):
        if (((yp[i] <= y) and (y < yp[j])) or ((yp[j] <= y) and (y < yp[i]))) and \
                (x < (xp[j] - xp[i]) * (y - yp[i]) / (yp[j] - yp[i]) + xp[i]):
            c = not c
        j = i
    return c


def polygon(r, c, shape=None):
    r = np.asarray(r, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    minr = int(max(0, r.min()))
    maxr = int(np.ceil(r.max()))
    minc = int(max(0, c.min()))
    maxc = int(np.ceil(c.max()))
    if shape is not None:
        maxr = min(shape[0] - 1, maxr)
        maxc = min(shape[1] - 1, maxc)
    rr, cc = [], []
    for r_ in range(minr, maxr + 1):
        for c_ in range(minc, maxc + 1):
            if _point_in_polygon(c, r, float(c_), float(r_)):
                rr.append(r_)
                cc.append(c_)
    return np.array(rr, dtype=np.intp), np.array(cc, dtype=np.intp)


def line(r0, c0, r1, c1):
    steep = False
    r, c = int(r0), int(c0)
    dr, dc = abs(int(r1) - r), abs(int(c1) - c)
    sc = 1 if (c1 - c) > 0 else -1
    sr = 1 if (r1 - r) > 0 else -1
    if dr > dc:
        steep = True
        c, r = r, c
        dc, dr = dr, dc
        sc, sr = sr, sc
    d = 2 * dr - dc
    rr = np.zeros(max(dc, dr) + 1, dtype=np.intp)
    cc = np.zeros(max(dc, dr) + 1, dtype=np.intp)
    for i in range(dc):
        if steep:
            rr[i], cc[i] = c, r
        else:
            rr[i], cc[i] = r, c
        while d >= 0:
            r += sr
            d -= 2 * dc
        c += sc
        d += 2 * dr
    rr[dc], cc[dc] = r1, c1
    return rr, cc


def _polygon_clip(rp, cp, r0, c0, r1, c1):
    from matplotlib import path, transforms
    poly = path.Path(np.vstack((rp, cp)).T, closed=True)
    clip_rect = transforms.Bbox([[r0, c0], [r1, c1]])
    poly_clipped = poly.clip_to_bbox(clip_rect).to_polygons()[0]
    if np.all(poly_clipped[-1] == poly_clipped[-2]):
        poly_clipped = poly_clipped[:-1]
    return poly_clipped[:, 0], poly_clipped[:, 1]


def polygon_perimeter(r, c, shape=None, clip=False):
    if clip:
        if shape is None:
            raise ValueError("Must specify clipping shape")
        clip_box = np.array([0, 0, shape[0] - 1, shape[1] - 1])
    else:
        clip_box = np.array([np.min(r), np.min(c), np.max(r), np.max(c)])
    r, c = _polygon_clip(r, c, *clip_box)
    r = np.round(r).astype(int)
    c = np.round(c).astype(int)
    rr, cc = [], []
    for i in range(len(r) - 1):
        line_r, line_c = line(r[i], c[i], r[i + 1], c[i + 1])
        rr.extend(line_r)
        cc.extend(line_c)
    rr = np.asarray(rr)
    cc = np.asarray(cc)
    if shape is None:
        return rr, cc
    mask = (rr >= 0) & (rr < shape[0]) & (cc >= 0) & (cc < shape[1])
    return rr[mask], cc[mask]
