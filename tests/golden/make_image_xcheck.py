#!/usr/bin/env python3
"""Container-only generator of tests/golden/image_xcheck.npz: an INDEPENDENT derivation of the image stages a1-a5
 in numpy / scipy.ndimage -- whole-array filters, reshapes and broadcasting instead of the oracle's
pixel loops, np.sum instead of its 64-bin tree, np.arctan instead of its own arctangent -- on three small synthetic
frames rendered here (not by the oracle's or the library's synthetic camera); since round 4 also a5's gate and a4.3 (ring tests on the
threshold image and on the grey image, de-duplication) as whole-array gathers.

What it pins: the oracle (tests/test_golden.py, CPU) and the HIP path (tests/test_gpu_parity.py, GPU) must both
reproduce these arrays: grey, Q5 maps and remapped images, threshold images, candidate lists, suppressed lists bit for
bit; refined corners to 1e-9 px (the summation order differs).  What it does not pin: that DESIGN.md section 3 is what
the reference's external detector computes (it is not: SURVEY.md 8(c), parity unpinned) -- it only shows that two
independent programs written from the same definitions agree, so a slip in either restatement would surface.

Neither this script nor scipy travels to the GPU box; only the .npz does.  Usage: python tests/golden/make_image_xcheck.py
"""
import os
import numpy as np
from scipy import ndimage as ndi

W, H = 320, 240
INT32_MIN = -(1 << 31)
P = dict(harris_thresh=200000, cand_margin=8, nms_radius=5, win=5, max_iter=30, eps=1e-3)


# ---- inputs: perspective views of a 9x7-square board, 3x3 supersampled, tinted, blurred, noisy ---------------------------
def render(seed, angle_deg, scale, tx, ty, persp, sigma, noise):
    rng = np.random.default_rng(seed)
    ss = 3
    v, u = np.mgrid[0:H * ss, 0:W * ss]
    u = (u + 0.5) / ss - 0.5
    v = (v + 0.5) / ss - 0.5
    a = np.deg2rad(angle_deg)
    # image -> board plane (squares of side 1, board centred at the origin)
    x = (u - tx) / scale
    y = (v - ty) / scale
    xr = np.cos(a) * x + np.sin(a) * y
    yr = -np.sin(a) * x + np.cos(a) * y
    d = 1.0 + persp[0] * xr + persp[1] * yr
    bx = xr / d + 4.5
    by = yr / d + 3.5
    inside = (bx >= 0) & (bx < 9) & (by >= 0) & (by < 7)
    quiet = (bx >= -1) & (bx < 10) & (by >= -1) & (by < 8)
    black = ((np.floor(bx) + np.floor(by)) % 2 == 0) & inside
    img = np.empty((H * ss, W * ss, 3))
    img[...] = (128.0, 120.0, 135.0)
    img[quiet] = (235.0, 230.0, 240.0)
    img[black] = (20.0, 25.0, 18.0)
    img = img.reshape(H, ss, W, ss, 3).mean(axis=(1, 3))
    if sigma > 0:
        img = np.stack([ndi.gaussian_filter(img[..., c], sigma) for c in range(3)], -1)
    img = img + rng.normal(0.0, noise, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ---- a1 -----------------------------------------------------------------------------------------------------------------
def grey_of(bgr):
    b = bgr.astype(np.int64)
    return ((1868 * b[..., 0] + 9617 * b[..., 1] + 4899 * b[..., 2] + 8192) >> 14).astype(np.uint8)


# ---- a2: Q5 map of the destination grid and the fixed-point bilinear remap ---------------------------------------------
def map_q5(K, model, D, w, h):
    fx, fy, cx, cy = K[0], K[4], K[2], K[5]
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    x = (u - cx) / fx
    y = (v - cy) / fy
    if model == 1:                                  # plumb-bob: kr in Horner form
        k1, k2, p1, p2, k3 = D[:5]
        r2 = x * x + y * y
        kr = 1.0 + ((k3 * r2 + k2) * r2 + k1) * r2
        _2xy = (2.0 * x) * y
        xd = x * kr + p1 * _2xy + p2 * (r2 + 2.0 * (x * x))
        yd = y * kr + p1 * (r2 + 2.0 * (y * y)) + p2 * _2xy
        xs = fx * xd + cx
        ys = fy * yd + cy
    elif model == 2:                                # fisheye (OpenCV convention)
        k1, k2, k3, k4 = D[:4]
        r = np.sqrt(x * x + y * y)
        th = np.arctan(r)
        t2 = th * th
        thd = th * (1.0 + (((k4 * t2 + k3) * t2 + k2) * t2 + k1) * t2)
        s = np.where(r > 1e-8, thd / np.where(r > 1e-8, r, 1.0), 1.0)
        xs = (fx * x) * s + cx
        ys = (fy * y) * s + cy
    else:
        xs, ys = u, v
    q = lambda t: np.clip(np.rint(t * 32.0), INT32_MIN, (1 << 31) - 1).astype(np.int64)
    return q(xs), q(ys)


def remap_q5(grey, X, Y):
    h, w = grey.shape
    g = np.zeros((h + 2, w + 2), np.int64)          # one ring of zeros: BORDER_CONSTANT 0 for taps just outside
    g[1:-1, 1:-1] = grey
    ix, iy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31

    def tap(xx, yy):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(ok, g[np.clip(yy, -1, h) + 1, np.clip(xx, -1, w) + 1], 0)
    acc = (32 - fx) * (32 - fy) * tap(ix, iy) + fx * (32 - fy) * tap(ix + 1, iy) + (32 - fx) * fy * tap(ix, iy + 1) + fx * fy * tap(ix + 1, iy + 1)
    return ((acc + 512) >> 10).astype(np.uint8)


# ---- a3 -----------------------------------------------------------------------------------------------------------------
def threshold_tiles(grey, min_contrast):
    h, w = grey.shape
    assert h % 4 == 0 and w % 4 == 0
    t = grey.reshape(h // 4, 4, w // 4, 4).astype(np.int64)
    tmin, tmax = t.min(axis=(1, 3)), t.max(axis=(1, 3))
    dmin = ndi.minimum_filter(tmin, size=3, mode="nearest")      # a replicated border tile is already in the window
    dmax = ndi.maximum_filter(tmax, size=3, mode="nearest")
    mn = np.kron(dmin, np.ones((4, 4), np.int64))
    mx = np.kron(dmax, np.ones((4, 4), np.int64))
    out = np.where(grey.astype(np.int64) > mn + (mx - mn) // 2, 255, 0)
    return np.where(mx - mn < min_contrast, 127, out).astype(np.uint8)


# ---- a4.1 ---------------------------------------------------------------------------------------------------------------
def harris_response(grey):
    h, w = grey.shape
    g = grey.astype(np.int64)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    gx = ndi.correlate(g, kx, mode="constant") >> 3              # arithmetic shift = floor
    gy = ndi.correlate(g, kx.T, mode="constant") >> 3
    gx[[0, -1], :] = 0; gx[:, [0, -1]] = 0; gy[[0, -1], :] = 0; gy[:, [0, -1]] = 0
    box = np.ones((5, 5), np.int64)
    A = ndi.correlate(gx * gx, box, mode="constant") >> 4
    B = ndi.correlate(gx * gy, box, mode="constant") >> 4
    C = ndi.correlate(gy * gy, box, mode="constant") >> 4
    Rf = A * C - B * B - (((A + C) ** 2) >> 4)
    R = np.full((h, w), INT32_MIN, np.int64)
    ys, xs = np.arange(4, h - 3, 2), np.arange(4, w - 3, 2)
    R[np.ix_(ys, xs)] = Rf[np.ix_(ys, xs)]
    assert R.max() < (1 << 31)
    return R


def harris_candidates(R, thresh, margin):
    h, w = R.shape
    margin = max(margin, 6)
    m0 = (margin + 1) & ~1
    ys, xs = np.arange(m0, h - margin, 2), np.arange(m0, w - margin, 2)
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    r = R[yy, xx]
    ok = r >= thresh
    for dy, dx, strict in ((-2, -2, 1), (-2, 0, 1), (-2, 2, 1), (0, -2, 1), (0, 2, 0), (2, -2, 0), (2, 0, 0), (2, 2, 0)):
        nb = R[yy + dy, xx + dx]
        ok &= (r > nb) if strict else (r >= nb)
    sel = np.argwhere(ok)                                        # row-major = (y, x) order
    return np.array([(xs[j], ys[i], r[i, j]) for i, j in sel], np.int64).reshape(-1, 3)


# ---- a4.2 ---------------------------------------------------------------------------------------------------------------
def suppress(c, radius):
    if len(c) == 0:
        return c
    x, y, s = c[:, 0], c[:, 1], c[:, 2]
    near = (np.abs(x[:, None] - x[None, :]) <= radius) & (np.abs(y[:, None] - y[None, :]) <= radius)
    idx = np.arange(len(c))
    beats = (s[None, :] > s[:, None]) | ((s[None, :] == s[:, None]) & (idx[None, :] < idx[:, None]))
    np.fill_diagonal(near, False)
    return c[~(near & beats).any(axis=1)]


# ---- a5 -----------------------------------------------------------------------------------------------------------------
def corner_subpix(grey, pts, win, max_iter, eps):
    h, w = grey.shape
    g = grey.astype(np.float64)
    k = np.arange(-win, win + 1, dtype=np.float64)
    m1 = np.exp(-((k / win) ** 2))
    mask = m1[:, None] * m1[None, :]
    px, py = np.meshgrid(k, k)                                   # px = column offset, py = row offset
    out = np.zeros((len(pts), 2))
    for q, (x0, y0) in enumerate(pts[:, :2].astype(np.float64)):
        cx, cy = x0, y0
        bad = False
        for it in range(max_iter):
            ix, iy = int(np.floor(cx)), int(np.floor(cy))
            if ix - win - 1 < 0 or iy - win - 1 < 0 or ix + win + 2 > w - 1 or iy + win + 2 > h - 1:
                bad = True
                break
            fx, fy = cx - ix, cy - iy
            p = g[iy - win - 1: iy + win + 3, ix - win - 1: ix + win + 3]          # (2 win + 4)^2 pixels
            S = (1 - fx) * (1 - fy) * p[:-1, :-1] + fx * (1 - fy) * p[:-1, 1:] + (1 - fx) * fy * p[1:, :-1] + fx * fy * p[1:, 1:]
            gx = S[1:-1, 2:] - S[1:-1, :-2]
            gy = S[2:, 1:-1] - S[:-2, 1:-1]
            a = np.sum(gx * gx * mask); b = np.sum(gx * gy * mask); c = np.sum(gy * gy * mask)
            bb1 = np.sum(mask * (gx * gx * px + gx * gy * py)); bb2 = np.sum(mask * (gx * gy * px + gy * gy * py))
            det = a * c - b * b
            if abs(det) <= np.finfo(float).eps ** 2:
                break
            dx, dy = (c * bb1 - b * bb2) / det, (a * bb2 - b * bb1) / det
            cx, cy = cx + dx, cy + dy
            if cx < 0 or cx >= w or cy < 0 or cy >= h:
                break
            if dx * dx + dy * dy <= eps * eps:
                break
        if bad or abs(cx - x0) > win or abs(cy - y0) > win:
            cx, cy = x0, y0
        out[q] = (cx, cy)
    return out


# ---- a5's gate and a4.3 (round 4): ring samples as whole-array gathers, transitions by np.roll, de-duplication pairwise -------------
def ring_offsets(radius):
    """the 16 ring positions: the pixel nearest to the circle point every 22.5 degrees, counter-clockwise from +x"""
    ang = 2.0 * np.pi * np.arange(16) / 16.0
    return np.stack([np.rint(radius * np.cos(ang)), np.rint(radius * np.sin(ang))], 1).astype(np.int64)


def _transitions(above):
    return (above != np.roll(above, -1, axis=1)).sum(1)


def gate_np(grey, pts, min_contrast, radius=11):
    """a5's gate at the UNREFINED pixel: the grey ring of that radius, cut at its own mid level, changes side at least four times and
    spans min_contrast; a ring that leaves the image passes"""
    h, w = grey.shape
    x, y = pts[:, 0].astype(np.int64), pts[:, 1].astype(np.int64)
    room = (x >= radius) & (y >= radius) & (x < w - radius) & (y < h - radius)
    ro = ring_offsets(radius)
    g = grey[np.where(room, y, radius)[:, None] + ro[None, :, 1], np.where(room, x, radius)[:, None] + ro[None, :, 0]].astype(np.int64)
    lo, hi = g.min(1), g.max(1)
    ok = (hi - lo >= min_contrast) & (_transitions(g > ((lo + hi) >> 1)[:, None]) >= 4)
    return np.where(room, ok, True)


def validate_np(pre, xy, binimg, grey, min_contrast, dedupe=2):
    """a4.3: the rounded refined position has a radius-5 ring inside the image on which the threshold image shows no flat sample and
    exactly four changes AND the grey ring, cut at its own mid level, spans min_contrast with exactly four changes; of two such
    entries within +-dedupe pixels the one with the larger score stays (equal: the earlier in the list).  Returns (x, y, score) rows
    and the refined positions, in list order."""
    h, w = grey.shape
    xi = np.floor(xy[:, 0] + 0.5).astype(np.int64)
    yi = np.floor(xy[:, 1] + 0.5).astype(np.int64)
    room = (xi >= 5) & (yi >= 5) & (xi < w - 5) & (yi < h - 5)
    ro = ring_offsets(5)
    yy = np.where(room, yi, 5)[:, None] + ro[None, :, 1]
    xx = np.where(room, xi, 5)[:, None] + ro[None, :, 0]
    b = binimg[yy, xx].astype(np.int64)
    g = grey[yy, xx].astype(np.int64)
    lo, hi = g.min(1), g.max(1)
    ok = room & (b != 127).all(1) & (_transitions(b) == 4) & (hi - lo >= min_contrast) & (_transitions(g > ((lo + hi) >> 1)[:, None]) == 4)
    idx = np.nonzero(ok)[0]
    px, py, sc = xi[idx], yi[idx], pre[idx, 2].astype(np.int64)
    near = (np.abs(px[:, None] - px[None, :]) <= dedupe) & (np.abs(py[:, None] - py[None, :]) <= dedupe)
    order = np.arange(len(idx))
    beats = near & ((sc[None, :] > sc[:, None]) | ((sc[None, :] == sc[:, None]) & (order[None, :] < order[:, None])))      # [a, b]: b beats a
    keep = ~beats.any(1)
    return np.stack([px[keep], py[keep], sc[keep]], 1), xy[idx][keep]


# ---- a4.3 + a6: where the 8 x 6 inner corners of each view lie, and in which order they must be reported ------------------
def ideal_corners(view, cols=8, rows=6):
    """image positions of the inner corners from the view's own projective map (the inverse of render's), ordered by
    DESIGN.md section 3 (a6): index = row * cols + col, columns along the board's long axis, (column, row) right-handed in
    the image (x right, y down), and of the two labelings that leaves (the board is 180-degree symmetric) the one whose
    corner 0 precedes its last corner in (y, x) order"""
    a = np.deg2rad(view["angle_deg"])
    p0, p1 = view["persp"]

    def fwd(bx, by):
        X, Y = bx - 4.5, by - 3.5
        d = 1.0 / (1.0 - p0 * X - p1 * Y)
        xr, yr = X * d, Y * d
        x, y = np.cos(a) * xr - np.sin(a) * yr, np.sin(a) * xr + np.cos(a) * yr
        return np.array([x * view["scale"] + view["tx"], y * view["scale"] + view["ty"]])
    grid = np.array([[fwd(1.0 + c, 1.0 + r) for c in range(cols)] for r in range(rows)])     # [row][col] in board order
    best = None
    for flip in (False, True):
        g = grid[::-1, ::-1] if flip else grid
        cdir, rdir = g[0, cols - 1] - g[0, 0], g[rows - 1, 0] - g[0, 0]
        if cdir[0] * rdir[1] - cdir[1] * rdir[0] < 0:
            g = g[:, ::-1]                                                                    # mirror the columns: right-handed
        pts = g.reshape(-1, 2)
        first, last = pts[0], pts[-1]
        ok = (round(first[1]), round(first[0])) < (round(last[1]), round(last[0]))
        if ok:
            assert best is None
            best = pts
    assert best is not None
    return best


# ---- square fiducials: an 8 x 8-cell tag drawn from its code word, and where its corners must be reported -----------------
FW, FH = 640, 480


def render_tags(codes, placements, seed=11, noise=1.5, sigma=0.6):
    """placements: (index into codes, centre x, centre y, cell size in px, rotation in degrees, (p0, p1) perspective).
    Tag frame: (c, r) in [0, 8]^2, origin at the printed tag's top-left corner, c to the right, r downwards; cell (r, c)
    is black on the one-cell border, else white iff bit 35 - ((r - 1) * 6 + (c - 1)) of the code word is set (payload
    row-major, MSB first).  Returns the frame and, per tag, its corners bl, br, tr, tl in image coordinates."""
    rng = np.random.default_rng(seed)
    ss = 3
    v, u = np.mgrid[0:FH * ss, 0:FW * ss]
    u = (u + 0.5) / ss - 0.5
    v = (v + 0.5) / ss - 0.5
    img = np.full((FH * ss, FW * ss), 225.0)
    corners = []
    for k, cx, cy, cell, ang, (p0, p1) in placements:
        a = np.deg2rad(ang)
        x, y = (u - cx) / cell, (v - cy) / cell
        xr, yr = np.cos(a) * x + np.sin(a) * y, -np.sin(a) * x + np.cos(a) * y
        d = 1.0 + p0 * xr + p1 * yr
        c, r = xr / d + 4.0, yr / d + 4.0
        inside = (c >= 0) & (c < 8) & (r >= 0) & (r < 8)
        ci, ri = np.clip(np.floor(c).astype(int), 0, 7), np.clip(np.floor(r).astype(int), 0, 7)
        bits = np.zeros((8, 8), bool)
        for rr in range(6):
            for cc in range(6):
                bits[rr + 1, cc + 1] = (int(codes[k]) >> (35 - (rr * 6 + cc))) & 1
        img[inside & ~bits[ri, ci]] = 25.0

        def fwd(c_, r_):
            X, Y = c_ - 4.0, r_ - 4.0
            dd = 1.0 / (1.0 - p0 * X - p1 * Y)
            xr_, yr_ = X * dd, Y * dd
            return [(np.cos(a) * xr_ - np.sin(a) * yr_) * cell + cx, (np.sin(a) * xr_ + np.cos(a) * yr_) * cell + cy]
        corners.append([fwd(0, 8), fwd(8, 8), fwd(8, 0), fwd(0, 0)])            # bl, br, tr, tl
    img = img.reshape(FH, ss, FW, ss).mean(axis=(1, 3))
    img = ndi.gaussian_filter(img, sigma) + rng.normal(0.0, noise, (FH, FW))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), np.array(corners)



def corner_class_np(grey, pts, min_contrast):
    """The convex-black-corner test of the fiducial stages (DESIGN.md section 3, a4 for fiducials), written from the text with
    whole-array machinery, for many pixels at once: the 16 pixels nearest to a circle of radius 5 at multiples of 22.5
    degrees (generated here from the angles, not copied from a table), their mid level, and "the dark samples form one arc
    of 2 to 7" read off the circular run structure with np.roll / np.diff -- where the oracle walks the ring.  pts: N x 2
    (x, y); returns N booleans."""
    g = grey.astype(np.int64)
    h, w = g.shape
    ang = np.arange(16) * (np.pi / 8)
    ring = np.stack([np.rint(5.0 * np.cos(ang)), np.rint(5.0 * np.sin(ang))], 1).astype(np.int64)      # nearest pixel to the circle point
    pts = np.asarray(pts, np.int64)
    x, y = pts[:, 0], pts[:, 1]
    room = (x >= 5) & (y >= 5) & (x < w - 5) & (y < h - 5)
    xs = np.where(room, x, 5)[:, None] + ring[None, :, 0]
    ys = np.where(room, y, 5)[:, None] + ring[None, :, 1]
    v = g[ys, xs]                                                   # (N, 16)
    lo, hi = v.min(1), v.max(1)
    white = v > ((lo + hi) // 2)[:, None]
    changes = (white != np.roll(white, -1, axis=1)).sum(1)
    dark = (~white).sum(1)
    return room & (hi - lo >= min_contrast) & (changes == 2) & (dark >= 2) & (dark <= 7)

def refine_edges_np(grey, qi):
    """The refine_edges form of the tag-corner refinement (DESIGN.md section 3, a5 for fiducials), written from the text with
    whole-array machinery: every (edge, sample, offset) position at once by broadcasting, the fixed-point bilinear samples by
    fancy indexing, the moments by np.sum, the line's normal from numpy's symmetric eigen-decomposition (LAPACK) and the
    corners by np.linalg.solve -- where the oracle walks loops, sums along a pairing tree, takes the eigenvector in closed
    form and intersects by Cramer's rule.  qi: 4 x 2 integer corners, clockwise on screen."""
    g = grey.astype(np.int64)
    h, w = g.shape
    a = np.asarray(qi, float)                       # (4, 2)
    b = np.roll(a, -1, axis=0)
    d = b - a
    L = np.hypot(d[:, 0], d[:, 1])
    n = np.stack([d[:, 1] / L, -d[:, 0] / L], 1)    # outward normal
    alpha = (np.arange(16) + 2) / 19.0
    base = a[:, None, :] + alpha[None, :, None] * d[:, None, :]                     # (4, 16, 2)
    j = np.arange(-8, 9) * 0.5
    pos = base[:, :, None, :] + j[None, None, :, None] * n[:, None, None, :]        # (4, 16, 17, 2)
    inside = (pos[..., 0] >= 0) & (pos[..., 1] >= 0) & (pos[..., 0] <= w - 2) & (pos[..., 1] <= h - 2)
    X = np.rint(np.where(inside, pos[..., 0], 0.0) * 16.0).astype(np.int64)
    Y = np.rint(np.where(inside, pos[..., 1], 0.0) * 16.0).astype(np.int64)
    ix, iy, fx, fy = X >> 4, Y >> 4, X & 15, Y & 15
    Pv = (16 - fx) * (16 - fy) * g[iy, ix] + fx * (16 - fy) * g[iy, ix + 1] + (16 - fx) * fy * g[iy + 1, ix] + fx * fy * g[iy + 1, ix + 1]
    k = np.arange(-6, 7)
    g1, g2 = Pv[:, :, k + 2 + 8], Pv[:, :, k - 2 + 8]
    ok = inside[:, :, k + 2 + 8] & inside[:, :, k - 2 + 8] & (g1 > g2)
    wt = np.where(ok, (g1 - g2) ** 2, 0)
    Mc = wt.sum(-1)
    Mn = (wt * k).sum(-1)
    valid = Mc > 0
    n0 = np.where(valid, Mn / np.where(valid, Mc, 1), 0.0) * 0.5
    pts = (alpha[None, :, None] * d[:, None, :]) + n0[..., None] * n[:, None, :]    # relative to a
    lines = []
    for e in range(4):
        p = pts[e][valid[e]]
        if len(p) >= 4:
            mu = p.mean(0)
            C = (p - mu).T @ (p - mu) / len(p)
            wv, V = np.linalg.eigh(C)
            lines.append((a[e] + mu, V[:, 0]))           # normal = eigenvector of the smaller eigenvalue
        else:
            lines.append((a[e], n[e]))
    out = np.zeros((4, 2))
    for c in range(4):
        (Ea, va), (Eb, vb) = lines[(c + 3) & 3], lines[c]
        A = np.array([va, vb])
        if abs(np.linalg.det(A)) ** 2 > 1e-6 * (va @ va) * (vb @ vb):
            P_ = np.linalg.solve(A, np.array([va @ Ea, vb @ Eb]))
            out[c] = P_ if np.sum((P_ - a[c]) ** 2) <= 16.0 else a[c]
        else:
            out[c] = a[c]
    return out


def main():
    views = [dict(seed=1, angle_deg=12.0, scale=21.0, tx=158.0, ty=121.0, persp=(0.012, -0.008), sigma=0.7, noise=1.5),
             dict(seed=2, angle_deg=-33.0, scale=17.5, tx=170.0, ty=112.0, persp=(-0.02, 0.015), sigma=1.0, noise=2.5),
             dict(seed=3, angle_deg=78.0, scale=14.0, tx=150.0, ty=126.0, persp=(0.0, 0.025), sigma=0.5, noise=4.0)]
    frames = np.stack([render(**v) for v in views])
    grey = np.stack([grey_of(f) for f in frames])
    out = dict(frames=frames, grey=grey, **{k: np.array(v) for k, v in P.items()})
    out["bin32"] = np.stack([threshold_tiles(g, 32) for g in grey])
    out["bin5"] = np.stack([threshold_tiles(g, 5) for g in grey])
    cands, pres, xys = [], [], []
    for g in grey:
        R = harris_response(g)
        c = harris_candidates(R, P["harris_thresh"], P["cand_margin"])
        p = suppress(c, P["nms_radius"])
        cands.append(c); pres.append(p); xys.append(corner_subpix(g, p, P["win"], P["max_iter"], P["eps"]))
    out["cand"] = np.concatenate(cands); out["cand_n"] = np.array([len(c) for c in cands])
    out["pre"] = np.concatenate(pres); out["pre_n"] = np.array([len(p) for p in pres])
    out["pre_xy"] = np.concatenate(xys)
    out["ideal_xy"] = np.stack([ideal_corners(v) for v in views])
    # a5's gate and a4.3 at min_contrast 32 (round 4): gated entries are not refined and reach a4.3 as (-1, -1)
    gates, kepts, kxys = [], [], []
    for f, g in enumerate(grey):
        gt = gate_np(g, pres[f], 32)
        gxy = np.where(gt[:, None], xys[f], -1.0)
        k, kxy = validate_np(pres[f], gxy, out["bin32"][f], g, 32)
        gates.append(gt); kepts.append(k); kxys.append(kxy)
    # a4.3 alone on EVERY refined entry (no gate in front), at both contrast settings: more entries reach the ring tests
    for mc, key in ((32, "bin32"), (5, "bin5")):
        ka = [validate_np(pres[f], xys[f], out[key][f], g, mc)[0] for f, g in enumerate(grey)]
        out["keptall%d" % mc] = np.concatenate(ka); out["keptall%d_n" % mc] = np.array([len(k) for k in ka])
        print("a4.3 on every refined entry, min_contrast %d: validated" % mc, out["keptall%d_n" % mc])
    out["gate32"] = np.concatenate(gates)
    out["kept32"] = np.concatenate(kepts); out["kept32_n"] = np.array([len(k) for k in kepts]); out["kept32_xy"] = np.concatenate(kxys)
    print("gate (numpy): held back", [int((~g_).sum()) for g_ in gates], "of", [len(g_) for g_ in gates], "; validated", out["kept32_n"])
    K = np.array([0.9 * W, 0.0, (W - 1) * 0.5, 0.0, 0.9 * W, (H - 1) * 0.5, 0.0, 0.0, 1.0])
    cams = [(1, np.array([-0.28, 0.07, 2e-4, -1e-4, 0.0, 0, 0, 0])), (1, np.array([-0.45, 0.25, 3e-3, -2e-3, -0.05, 0, 0, 0])),
            (2, np.array([-0.2, 0.05, -0.01, 0.002, 0, 0, 0, 0]))]
    out["K"] = K
    out["cam_model"] = np.array([m for m, _ in cams]); out["cam_D"] = np.stack([d for _, d in cams])
    maps = [map_q5(K, m, d, W, H) for m, d in cams]
    out["map_x"] = np.stack([m[0] for m in maps]).astype(np.int32); out["map_y"] = np.stack([m[1] for m in maps]).astype(np.int32)
    out["remapped"] = np.stack([remap_q5(grey[0], mx, my) for mx, my in maps])
    fam = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "robot_camera_calibration_amd", "data", "family36b.txt")
    codes = [int(l, 16) for l in open(fam) if l.strip() and not l.startswith("#")]
    place = [(0, 110.0, 100.0, 11.0, 0.0, (0.0, 0.0)), (7, 300.0, 95.0, 9.0, 90.0, (0.01, 0.0)), (13, 500.0, 110.0, 12.0, 180.0, (0.0, -0.012)),
             (21, 120.0, 330.0, 10.0, 270.0, (0.008, 0.008)), (34, 320.0, 320.0, 13.0, 37.0, (-0.01, 0.006)), (47, 520.0, 340.0, 9.5, -122.0, (0.0, 0.015))]
    fid_frame, fid_corners = render_tags(codes, place)
    out["fid_frame"] = fid_frame; out["fid_ids"] = np.array([p_[0] for p_ in place]); out["fid_corners"] = fid_corners
    # refine_edges form: started from the true corners rounded to pixels and knocked off by up to 2 px (the quad search hands
    # over positions of that quality), in the clockwise-on-screen order tl, tr, br, bl
    knock = np.array([[[1, 0], [0, -1], [-1, 1], [0, 0]], [[0, 0], [2, 1], [0, -2], [-1, 0]], [[-1, -1], [0, 0], [1, 0], [0, 2]],
                      [[0, 1], [-2, 0], [0, 0], [1, -1]], [[1, 1], [0, 0], [-1, 0], [0, -1]], [[0, 0], [0, 0], [0, 0], [0, 0]]])
    fid_qi = np.rint(fid_corners[:, ::-1, :]).astype(np.int64) + knock
    out["fid_qi"] = fid_qi
    out["fid_refined"] = np.stack([refine_edges_np(fid_frame, q) for q in fid_qi])
    print("refine_edges (numpy) vs the drawn corners: max |err| per tag", np.abs(out["fid_refined"] - fid_corners[:, ::-1, :]).max(axis=(1, 2)).round(3))
    # convex-black-corner test: every pixel within 4 px of a drawn tag corner, plus a coarse grid over the frame (payload
    # corners, edges, flat areas)
    near = np.rint(fid_corners.reshape(-1, 2)).astype(np.int64)
    dd = np.stack(np.meshgrid(np.arange(-4, 5), np.arange(-4, 5)), -1).reshape(-1, 2)
    grid = np.stack(np.meshgrid(np.arange(3, FW, 7), np.arange(2, FH, 5)), -1).reshape(-1, 2)
    cls_pts = np.concatenate([(near[:, None, :] + dd[None, :, :]).reshape(-1, 2), grid])
    out["fid_class_pts"] = cls_pts.astype(np.int32)
    out["fid_class"] = corner_class_np(fid_frame, cls_pts, 32)
    print("corner class (numpy): %d points, %d convex black corners; at the drawn corners' own pixels: %s" %
          (len(cls_pts), int(out["fid_class"].sum()), corner_class_np(fid_frame, near, 32).astype(int).tolist()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image_xcheck.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; candidates", out["cand_n"], "suppressed", out["pre_n"],
          "flat fraction", [float((b == 127).mean().round(3)) for b in out["bin32"]])


if __name__ == "__main__":
    main()
