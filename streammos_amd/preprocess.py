"""Host-side (numpy) validation/test preprocessing that feeds ``AttNet.infer``.

Restates what ``DataloadVal`` does to a window of scans (reference: datasets/data_StreamMOS.py:397-599
and datasets/utils.py:98-192) without the disk walk: pose-align T scans, range-filter, pad to a fixed
point count, 4 test-time-augmentation flips, BEV / range-view quantisation and the 7-channel point
feature.  All arithmetic is float32 exactly where the reference's numpy is float32, and float64 only in
the pose transform (datasets/utils.py:116-126).
"""
import numpy as np

PAD_XYZI = -1000.0     # datasets/data_StreamMOS.py:567
PAD_Z = -4000.0        # datasets/data_StreamMOS.py:568
TTA_SIGNS = ((1, 1), (1, -1), (-1, 1), (-1, -1))   # x_sign outer, y_sign inner (data_StreamMOS.py:495-496)


class VoxelSpec:
    """Grid definition = ``General.Voxel`` of the reference config (config/StreamMOS.py:12-19)."""

    def __init__(self, range_x=(-50.0, 50.0), range_y=(-50.0, 50.0), range_z=(-4.0, 2.0),
                 bev_shape=(512, 512, 30), rv_shape=(64, 2048), RV_theta=(-25.0, 3.0)):
        self.range_x, self.range_y, self.range_z = tuple(range_x), tuple(range_y), tuple(range_z)
        self.bev_shape, self.rv_shape, self.RV_theta = tuple(bev_shape), tuple(rv_shape), tuple(RV_theta)

    @classmethod
    def from_config(cls, voxel_cfg):
        return cls(voxel_cfg.range_x, voxel_cfg.range_y, voxel_cfg.range_z,
                   voxel_cfg.bev_shape, voxel_cfg.rv_shape, voxel_cfg.RV_theta)


def pose_align(scan, pose_diff):
    """datasets/utils.py:116-126 (Trans): homogeneous transform in float64 with w forced to 1, result
    stored back as float32; intensity (and any further column) is carried through untouched."""
    homo = np.ones((4, scan.shape[0]), dtype=scan.dtype)
    homo[:3] = scan[:, :3].T
    moved = np.asarray(pose_diff, dtype=np.float64).dot(homo)
    out = scan.copy()
    out[:, :3] = moved[:3].T
    return out


def range_mask(scan, spec):
    """datasets/utils.py:107-113: half-open box lo <= p < hi on x, y, z."""
    keep = np.ones(scan.shape[0], dtype=bool)
    for d, (lo, hi) in enumerate((spec.range_x, spec.range_y, spec.range_z)):
        keep &= (scan[:, d] >= lo) & (scan[:, d] < hi)
    return keep


def pad_scan(scan, frame_point_num):
    """datasets/data_StreamMOS.py:563-572: constant pad with -1000 (z column -4000)."""
    pad = frame_point_num - scan.shape[0]
    if pad <= 0:
        raise ValueError("scan has %d in-range points, frame_point_num=%d leaves no padding "
                         "(the reference asserts pad_length > 0)" % (scan.shape[0], frame_point_num))
    tail = np.full((pad, scan.shape[1]), PAD_XYZI, dtype=scan.dtype)
    tail[:, 2] = PAD_Z
    return np.concatenate((scan, tail), axis=0), pad


def quantize_bev(xyz, spec):
    """datasets/utils.py:151-169: (p - lo) / cell in float32, no floor."""
    cols = []
    for d, (rng, size) in enumerate(zip((spec.range_x, spec.range_y, spec.range_z), spec.bev_shape)):
        cell = (rng[1] - rng[0]) / size
        cols.append((xyz[:, d] - rng[0]) / cell)
    return np.stack(cols, axis=-1)


def quantize_sphere(xyz, spec):
    """datasets/utils.py:172-192: (theta_quan, phi_quan) over a 64 x 2048 range image."""
    h, w = spec.rv_shape
    phi_hi = 180.0 * np.pi / 180.0
    phi_lo = -180.0 * np.pi / 180.0
    th_lo = spec.RV_theta[0] * np.pi / 180.0
    th_hi = spec.RV_theta[1] * np.pi / 180.0
    dphi = (phi_hi - phi_lo) / w
    dtheta = (th_hi - th_lo) / h
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    d = np.sqrt(x ** 2 + y ** 2 + z ** 2) + 1e-12
    phi_q = (phi_hi - np.arctan2(x, y)) / dphi
    theta_q = (th_hi - np.arcsin(z / d)) / dtheta
    return np.stack((theta_q, phi_q), axis=-1)


def point_features(xyzi, bev_coord):
    """datasets/data_StreamMOS.py:25-50: (x, y, z, intensity, dist, frac(x_quan), frac(y_quan))."""
    x, y, z = xyzi[:, 0], xyzi[:, 1], xyzi[:, 2]
    dist = np.sqrt(x ** 2 + y ** 2 + z ** 2) + 1e-12
    fx = bev_coord[:, 0] - np.floor(bev_coord[:, 0])
    fy = bev_coord[:, 1] - np.floor(bev_coord[:, 1])
    return np.stack((x, y, z, xyzi[:, 3], dist, fx, fy), axis=-1)


def form_batch(stacked, seq_num, spec):
    """One TTA variant: stacked = (T*N, 4) float32 -> xyzi (T,7,N,1), coord (T,N,3,1), sphere (T,N,2,1)."""
    n = stacked.shape[0] // seq_num
    xyzi = stacked[:, :4]
    bev = quantize_bev(xyzi, spec)
    sph = quantize_sphere(xyzi, spec)
    feat = point_features(xyzi, bev).astype(np.float32)
    feat = np.ascontiguousarray(feat.reshape(seq_num, n, 7).transpose(0, 2, 1))[..., None]
    bev = bev.astype(np.float32).reshape(seq_num, n, 3, 1)
    sph = sph.astype(np.float32).reshape(seq_num, n, 2, 1)
    return feat, bev, sph


def window_indices(i, n_frames, seq_num):
    """Which scans make up sample i (datasets/data_StreamMOS.py:424-467, ``meta_list_raw``): frames
    i, i-1, ..., except that the first seq_num-1 samples look forward instead."""
    if i < seq_num - 1:
        return [i + ht for ht in range(seq_num)]
    return [i - ht for ht in range(seq_num)]


def build_sample(scans, poses, frame_point_num, spec, tta=True):
    """scans: list of T raw (n_t,4) float32 scans, current first; poses: matching 4x4 float64 poses.

    Returns a dict with the arrays ``AttNet.infer`` consumes *without* the DataLoader's leading
    batch dim: pcds_xyzi (B,T,7,N,1), pcds_coord (B,T,N,3,1), pcds_sphere_coord (B,T,N,2,1) with
    B = 4 TTA variants (or 1), plus valid_mask (current scan) and pad_length.
    """
    seq_num = len(scans)
    inv_cur = np.linalg.inv(poses[0])
    aligned, masks, pads = [], [], []
    for scan, pose in zip(scans, poses):
        moved = pose_align(scan, inv_cur.dot(pose))
        keep = range_mask(moved, spec)
        padded, pad = pad_scan(moved[keep], frame_point_num)
        aligned.append(padded)
        masks.append(keep)
        pads.append(pad)
    stacked = np.concatenate(aligned, axis=0)
    feats, bevs, sphs = [], [], []
    for sx, sy in (TTA_SIGNS if tta else TTA_SIGNS[:1]):
        flipped = stacked.copy()
        flipped[:, 0] *= sx
        flipped[:, 1] *= sy
        f, b, s = form_batch(flipped, seq_num, spec)
        feats.append(f)
        bevs.append(b)
        sphs.append(s)
    return {
        "pcds_xyzi": np.stack(feats, axis=0),
        "pcds_coord": np.stack(bevs, axis=0),
        "pcds_sphere_coord": np.stack(sphs, axis=0),
        "valid_mask": masks[0],
        "pad_length": pads[0],
    }
