"""Constants of the row-window contract (counterpart of the reference's config.py:1-8 and
hybrid_kernel/config.h:4-6).  BLK_H x BLK_W is the window / condensed-block geometry shared with
include/hcspmm.h; WARP_SIZE is kept only because reference-style scripts star-import it -- the
gfx950 kernels are wave64 and never read it."""
BLK_H = 16
BLK_W = 8
WARP_SIZE = 32


def func(x):
    """Degree clamp used for the sqrt-degree vector (reference config.py:5-8): non-positive -> 1."""
    return x if x > 0 else 1
