import numpy as np


def disk(center, radius, shape=None):
    r0, c0 = center
    lo_r, hi_r = int(np.floor(r0 - radius)), int(np.ceil(r0 + radius))
    lo_c, hi_c = int(np.floor(c0 - radius)), int(np.ceil(c0 + radius))
    rr, cc = np.mgrid[lo_r:hi_r + 1, lo_c:hi_c + 1]
    keep = (rr - r0) ** 2 + (cc - c0) ** 2 < radius ** 2
    if shape is not None:
        keep &= (rr >= 0) & (rr < shape[0]) & (cc >= 0) & (cc < shape[1])
    return rr[keep], cc[keep]


def polygon(*a, **k):
    raise NotImplementedError("stand-in")


def polygon_perimeter(*a, **k):
    raise NotImplementedError("stand-in")
