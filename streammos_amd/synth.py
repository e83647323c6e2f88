"""Deterministic synthetic inputs: LiDAR scans, ego poses, labels and network weights.

There is no SemanticKITTI data and no pretrained checkpoint offline, so
every test and the benchmark run on the generator below (definition: SURVEY.md section 8d).
Everything is keyed on ``numpy.random.PCG64`` seeds, whose streams are stable across numpy
versions and platforms, so the build container and the GPU box see identical inputs.
"""
import zlib

import numpy as np

SCAN_SEED = 20240808
N_BEAMS = 64
N_AZIMUTH = 1875          # 64 * 1875 = 120 000 raw points per scan
N_MOVERS = 10
MOVER_POINTS = 200


def synthetic_pose(k):
    """Ego pose of frame k: x = 1.0*k metres, yaw = 0.01*k rad (KITTI poses.txt convention,
    identity calibration ``Tr``; the reference composes Tr^-1 * pose * Tr, datasets/utils.py:36-54)."""
    c, s = np.cos(0.01 * k), np.sin(0.01 * k)
    pose = np.eye(4, dtype=np.float64)
    pose[0, 0], pose[0, 1], pose[1, 0], pose[1, 1] = c, -s, s, c
    pose[0, 3] = 1.0 * k
    return pose


def synthetic_scan(frame_idx, n_beams=N_BEAMS, n_azimuth=N_AZIMUTH, with_labels=False):
    """HDL-64E-like scan in the sensor frame, float32 (n,4) = x, y, z, intensity (KITTI .bin layout).

    70 % of the returns hit the ground plane z = -1.73 m (range clipped to [2.5, 80] m), 30 % hit an
    obstacle at a log-uniform range in [2.5, 80] m.  Ten Gaussian blobs (sigma 0.5 m, 200 points each)
    translating 1 m/frame in the world frame replace the first 2000 points and carry label 2 (moving);
    every other point has label 1 (static).  Labels use the reference's learning-map ids
    (0 unlabeled / 1 static / 2 moving, config/StreamMOS.py:9).
    """
    rng = np.random.Generator(np.random.PCG64(SCAN_SEED + int(frame_idx)))
    elev = np.deg2rad(np.linspace(2.0, -24.8, n_beams))
    azim = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False)
    el, az = np.meshgrid(elev, azim, indexing="ij")
    el, az = el.reshape(-1), az.reshape(-1)
    n = el.shape[0]
    ground = rng.random(n) < 0.7
    with np.errstate(divide="ignore"):
        r_ground = np.where(el < 0, -1.73 / np.sin(np.minimum(el, -1e-6)), 80.0)
    r_ground = np.clip(r_ground, 2.5, 80.0)
    r_obst = np.exp(rng.uniform(np.log(2.5), np.log(80.0), n))
    r = np.where(ground, r_ground, r_obst)
    xyz = np.stack((r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)), axis=1)
    intensity = rng.random(n)
    labels = np.ones(n, dtype=np.int64)

    # moving objects live in the world frame; bring them into this frame's sensor frame
    n_mov = min(N_MOVERS * MOVER_POINTS, n // 4)
    per = n_mov // N_MOVERS
    if per > 0:
        obj_rng = np.random.Generator(np.random.PCG64(SCAN_SEED - 1))
        start = obj_rng.uniform(-30.0, 30.0, (N_MOVERS, 2))
        heading = obj_rng.uniform(0.0, 2 * np.pi, N_MOVERS)
        inv_pose = np.linalg.inv(synthetic_pose(frame_idx))
        for o in range(N_MOVERS):
            centre = np.array([start[o, 0] + frame_idx * np.cos(heading[o]),
                               start[o, 1] + frame_idx * np.sin(heading[o]), -1.0, 1.0])
            local = inv_pose.dot(centre)[:3]
            blob = local[None, :] + rng.normal(0.0, 0.5, (per, 3)) * np.array([1.0, 1.0, 0.4])
            xyz[o * per:(o + 1) * per] = blob
            labels[o * per:(o + 1) * per] = 2
    scan = np.concatenate((xyz, intensity[:, None]), axis=1).astype(np.float32)
    return (scan, labels) if with_labels else scan


# --------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------
_ALIASES = (
    # the reference registers these two modules twice (networks/multi_view_encoder.py:344-354), so the
    # same tensors show up under two key prefixes; both keys must receive identical values
    ("bev_net.header_unbalance_conv.", "bev_net.header_bev.1."),
    ("bev_net.res1_unbalance_conv.", "bev_net.res1_bev.1."),
)


def _canonical_key(key):
    for alias, canon in _ALIASES:
        if key.startswith(alias):
            return canon + key[len(alias):]
    return key


def seeded_tensor(key, shape, seed=0):
    """Deterministic float32/int64 numpy value for one state-dict entry, a pure function of
    (canonical key, shape, seed) -- independent of module construction order."""
    key = _canonical_key(key)
    rng = np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) << 8) + seed))
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return rng.normal(0.0, 0.1, shape).astype(np.float32)
    if leaf == "running_var":
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if "sampling_offsets" in key:
        # offsets of a few cells so that the four sampling points of a head differ per query
        if leaf == "weight":
            return rng.normal(0.0, 0.05, shape).astype(np.float32)
        return rng.uniform(-2.5, 2.5, shape).astype(np.float32)
    if "attention_weights" in key:
        scale = 0.2 if leaf == "weight" else 0.5
        return rng.normal(0.0, scale, shape).astype(np.float32)
    if len(shape) <= 1:
        if leaf == "weight":          # BatchNorm / LayerNorm scale
            return rng.uniform(0.5, 1.5, shape).astype(np.float32)
        return rng.normal(0.0, 0.1, shape).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    return rng.normal(0.0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)


def seeded_state_dict(layout, seed=0):
    """layout: iterable of (key, shape[, dtype]) or a module ``state_dict()``. Returns {key: torch.Tensor}."""
    import torch
    if hasattr(layout, "items"):
        layout = [(k, tuple(v.shape)) for k, v in layout.items()]
    out = {}
    for entry in layout:
        key, shape = entry[0], entry[1]
        out[key] = torch.from_numpy(seeded_tensor(key, shape, seed))
    return out
