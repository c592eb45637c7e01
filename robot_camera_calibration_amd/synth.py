"""Host-side helpers for the synthetic camera: pose sampling and analytic ground truth.

The reference renders no frames (rviz_simulator/src/simulate.cpp:44-68 publishes one interactive
cube; the camera class its header mentions, rviz_simulator/include/rviz_simulator/target.h:40, is
absent), so the workload generator is the build's own.
This module only samples poses and projects the board's corners in numpy; the pixels are rendered
by the HIP kernel behind rcc_synth_render_batch (csrc/k_synth.hip).
"""
import numpy as np

from . import abi


def rodrigues(rvec):
    r = np.asarray(rvec, dtype=np.float64)
    th = np.linalg.norm(r)
    if th < 2.3e-16:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def rotmat_to_rvec(R):
    R = np.asarray(R, dtype=np.float64)
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(v) / 2
    if s < 1e-9:
        if c > 0:
            return np.zeros(3)
        # angle pi: axis from the diagonal
        ax = np.sqrt(np.maximum((np.diag(R) + 1) / 2, 0))
        if R[0, 1] < 0:
            ax[1] = -ax[1]
        if R[0, 2] < 0:
            ax[2] = -ax[2]
        return ax / np.linalg.norm(ax) * th
    return v / (2 * s) * th


def board_object_points(cols, rows, square):
    """index = row*cols + col; x right, y up, z = 0, origin at the centre
    (object-frame convention of real_preprocessing/src/camera_pose.cpp:158-161)."""
    c, r = np.meshgrid(np.arange(cols), np.arange(rows))
    X = (c - (cols - 1) / 2.0) * square
    Y = ((rows - 1) / 2.0 - r) * square
    return np.stack([X.ravel(), Y.ravel(), np.zeros(cols * rows)], axis=1)


def distort_normalised(x, y, model, D):
    if model == abi.RCC_DIST_PLUMB_BOB:
        k1, k2, p1, p2, k3 = D[:5]
        r2 = x * x + y * y
        cd = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
        xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        return xd, yd
    if model == abi.RCC_DIST_FISHEYE:
        k1, k2, k3, k4 = D[:4]
        r = np.sqrt(x * x + y * y)
        th = np.arctan(r)
        thd = th * (1 + k1 * th ** 2 + k2 * th ** 4 + k3 * th ** 6 + k4 * th ** 8)
        s = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
        return x * s, y * s
    return x, y


def project_points(obj, rvec, tvec, K, model=abi.RCC_DIST_NONE, D=(0,) * 8):
    """numpy projection used for ground truth only (not the oracle, not the product)."""
    R = rodrigues(rvec)
    P = obj @ R.T + np.asarray(tvec, dtype=np.float64)
    x, y = P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]
    xd, yd = distort_normalised(x, y, model, D)
    return np.stack([K[0] * xd + K[2], K[4] * yd + K[5]], axis=1)


def sample_poses(n, cfg, seed=0xC0FFEE, z_range=(0.8, 2.5), max_tilt_deg=45.0, max_roll_deg=180.0,
                 border_px=14, margin_squares=1, first_index=0, half_extent_m=None):
    """Deterministic per-frame poses (cam_T_target as rvec,tvec): frame f is drawn from
    default_rng(seed + f).  Rejection-samples until the board plus its quiet zone projects fully
    inside the (distorted) image.  Returns an (n, 6) float64 array."""
    K = np.array(list(cfg.K))
    D = np.array(list(cfg.D))
    W, H = cfg.width, cfg.height
    cols, rows, sq = cfg.board_cols, cfg.board_rows, cfg.board_square
    hx = (cols + 1) / 2.0 + margin_squares
    hy = (rows + 1) / 2.0 + margin_squares
    if half_extent_m is not None:      # a planar target of arbitrary size (e.g. a fiducial grid)
        hx, hy, sq = half_extent_m[0], half_extent_m[1], 1.0
    outline = []
    for t in np.linspace(-1, 1, 9):
        outline += [(t * hx * sq, -hy * sq, 0), (t * hx * sq, hy * sq, 0), (-hx * sq, t * hy * sq, 0), (hx * sq, t * hy * sq, 0)]
    outline = np.array(outline)
    R0 = np.diag([1.0, -1.0, -1.0])
    out = np.zeros((n, 6))
    for f in range(n):
        rng = np.random.default_rng(seed + first_index + f)
        for attempt in range(20000):
            z = rng.uniform(*z_range)
            tilt = np.deg2rad(rng.uniform(0, max_tilt_deg))
            phi = rng.uniform(0, 2 * np.pi)
            roll = np.deg2rad(rng.uniform(-max_roll_deg, max_roll_deg))
            Rt = rodrigues(np.array([np.cos(phi), np.sin(phi), 0.0]) * tilt)
            Rz = rodrigues(np.array([0, 0, roll]))
            R = Rz @ Rt @ R0
            u0 = rng.uniform(0.15 * W, 0.85 * W)
            v0 = rng.uniform(0.15 * H, 0.85 * H)
            t = np.array([(u0 - K[2]) / K[0] * z, (v0 - K[5]) / K[4] * z, z])
            rvec = rotmat_to_rvec(R)
            P = outline @ R.T + t
            if np.any(P[:, 2] < 0.2):
                continue
            uv = project_points(outline, rvec, t, K, cfg.dist_model, D)
            if (uv[:, 0].min() >= border_px and uv[:, 0].max() <= W - 1 - border_px and
                    uv[:, 1].min() >= border_px and uv[:, 1].max() <= H - 1 - border_px):
                # the forward distortion must be monotone over the board: reject folded views
                out[f, :3] = rvec
                out[f, 3:] = t
                break
        else:
            raise RuntimeError("could not place the board in view for frame %d" % f)
    return out


def fiducial_grid_layout(gx, gy, tag_size, gap_permille=500):
    """Planar grid of square fiducials as k_synth renders it: returns (half_extent (hx, hy) in
    metres incl. the half-tag quiet zone, centres (gy*gx, 3) of the tags in the plane frame
    (x right, y up), ids).  Tag (i, j) = column i, row j (row 0 on top) has id j*gx+i."""
    pitch = tag_size * (1.0 + gap_permille / 1000.0)
    hx = 0.5 * (gx * pitch - (pitch - tag_size)) + 0.5 * tag_size
    hy = 0.5 * (gy * pitch - (pitch - tag_size)) + 0.5 * tag_size
    cs, ids = [], []
    for j in range(gy):
        for i in range(gx):
            cs.append([-hx + 0.5 * tag_size + i * pitch + 0.5 * tag_size, hy - 0.5 * tag_size - j * pitch - 0.5 * tag_size, 0.0])
            ids.append(j * gx + i)
    return (hx, hy), np.array(cs), np.array(ids)


def tag_object_points(tag_size):
    """bl, br, tr, tl in the tag frame (real_preprocessing/src/camera_pose.cpp:158-161)"""
    s = tag_size / 2.0
    return np.array([[-s, -s, 0], [s, -s, 0], [s, s, 0], [-s, s, 0]], float)
