"""SemanticKITTI on-disk formats: the wire contract between this runner and the reference's scripts
(SURVEY.md section 8, row f4).

* scans      ``velodyne/NNNNNN.bin``   float32 x 4 (x, y, z, intensity)
* labels     ``labels/NNNNNN.label``   uint32: low 16 bits semantic id, high 16 bits instance id
* predictions ``predictions/NNNNNN.label`` int32/uint32 LUT values 0 / 9 / 251 (val_StreamMOS.py:121-126,
  voxel_voting.py:244-249)
* ``poses.txt`` (12 floats per line, row-major 3x4) and ``calib.txt`` (``Tr: 12 floats``); the pose used by
  the network is ``inv(Tr) * pose * Tr`` (datasets/utils.py:11-54).
"""
import os

import numpy as np

# datasets/semantic-kitti.yaml `learning_map`: 0 = unlabeled/outlier, 2 = the moving classes, 1 = everything else
MOVING_IDS = (251, 252, 253, 254, 255, 256, 257, 258, 259)
STATIC_IDS = (9, 10, 11, 13, 15, 16, 18, 20, 30, 31, 32, 40, 44, 48, 49, 50, 51, 52, 60, 70, 71, 72, 80, 81, 99)
LEARNING_MAP_INV = {0: 0, 1: 9, 2: 251}
VALID_SEQUENCES = (8,)
TEST_SEQUENCES = tuple(range(11, 22))


def learning_map_lut():
    lut = np.zeros(260 + 100, dtype=np.int32)
    lut[list(STATIC_IDS)] = 1
    lut[list(MOVING_IDS)] = 2
    return lut


def _read_3x4(values):
    m = np.eye(4, dtype=np.float64)
    m[:3, :4] = np.asarray(values, dtype=np.float64).reshape(3, 4)
    return m


def read_calibration(path):
    calib = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            key, content = line.strip().split(":", 1)
            calib[key] = _read_3x4([float(v) for v in content.split()])
    return calib


def read_poses(path, calib):
    tr = calib["Tr"]
    tr_inv = np.linalg.inv(tr)
    poses = []
    with open(path) as f:
        for line in f:
            vals = [float(v) for v in line.split()]
            if len(vals) == 12:
                poses.append(tr_inv.dot(_read_3x4(vals)).dot(tr))
    return poses


def write_poses(path, poses):
    with open(path, "w") as f:
        for p in poses:
            f.write(" ".join("%.17g" % v for v in np.asarray(p)[:3, :4].reshape(-1)) + "\n")


def write_calibration(path, tr=None):
    tr = np.eye(4) if tr is None else tr
    row = " ".join("%.17g" % v for v in np.asarray(tr)[:3, :4].reshape(-1))
    with open(path, "w") as f:
        for key in ("P0", "P1", "P2", "P3"):
            f.write("%s: %s\n" % (key, " ".join(["0"] * 12)))
        f.write("Tr: %s\n" % row)


def read_scan(path):
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)


def read_label(path, mapped=True):
    raw = np.fromfile(path, dtype=np.uint32)
    sem = raw & 0xFFFF
    return learning_map_lut()[sem] if mapped else sem


def write_prediction(path, labels_012=None, lut_labels=None):
    """Writes the reference's prediction format: the learning_map_inv value per point as a 32-bit word."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if lut_labels is None:
        lut = np.zeros(3, dtype=np.int32)
        for k, v in LEARNING_MAP_INV.items():
            lut[k] = v
        lut_labels = lut[np.asarray(labels_012)]
    np.asarray(lut_labels, dtype=np.int32).tofile(path)


class MovingIoU:
    """Per-class IoU with label 0 ignored -- the formula of utils/metric.py:18-58 (``moving_iou`` is class 2)."""

    def __init__(self, n_classes=2):
        self.tp = np.zeros(n_classes, dtype=np.float64)
        self.pred = np.zeros(n_classes, dtype=np.float64)
        self.gt = np.zeros(n_classes, dtype=np.float64)

    def add(self, gt, pred):
        gt, pred = np.asarray(gt), np.asarray(pred)
        keep = gt != 0
        for i in range(self.tp.shape[0]):
            p, g = (pred == i + 1) & keep, (gt == i + 1) & keep
            self.tp[i] += (p & g).sum()
            self.pred[i] += p.sum()
            self.gt[i] += g.sum()

    def result(self):
        iou = self.tp / (self.gt + self.pred - self.tp + 1e-12)
        return {"static_iou": float(iou[0]), "moving_iou": float(iou[1]), "mean_iou": float(iou.mean())}
