import numpy as np


def _area(pts):
    if len(pts) < 3:
        return 0.0
    p = np.asarray(pts, dtype=float)
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def _ccw(pts):
    p = np.asarray(pts, dtype=float)
    x, y = p[:, 0], p[:, 1]
    s = float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    return p if s >= 0 else p[::-1]


def _clip(subject, clipper):
    out = [tuple(q) for q in subject]
    n = len(clipper)
    for i in range(n):
        a, b = clipper[i], clipper[(i + 1) % n]
        inp, out = out, []
        if not inp:
            break

        def side(q):
            return (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])

        s = inp[-1]
        for e in inp:
            se, ss = side(e), side(s)
            if se >= 0:
                if ss < 0:
                    t = ss / (ss - se)
                    out.append((s[0] + t * (e[0] - s[0]), s[1] + t * (e[1] - s[1])))
                out.append(e)
            elif ss >= 0:
                t = ss / (ss - se)
                out.append((s[0] + t * (e[0] - s[0]), s[1] + t * (e[1] - s[1])))
            s = e
    return out


class Polygon:
    def __init__(self, coords):
        self._pts = _ccw(np.asarray(coords, dtype=float)) if len(coords) >= 3 else np.zeros((0, 2))

    @property
    def area(self):
        return _area(self._pts)

    def intersection(self, other):
        if len(self._pts) < 3 or len(other._pts) < 3:
            return Polygon([])
        # a zero-width rectangle (ratio or size class 0) is a segment: GEOS returns
        # an empty/zero-area intersection; the orientation test above is meaningless there
        if self.area < 1e-12 or other.area < 1e-12:
            return Polygon([])
        return Polygon(_clip(self._pts, other._pts))
