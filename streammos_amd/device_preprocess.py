"""Device-side validation preprocessing (row f1): raw scans + pose differences in, network inputs out.

Host equivalent: ``streammos_amd.preprocess.build_sample`` (itself bit-exact against the reference's DataloadVal
arithmetic).  Only the raw scans cross PCIe; compaction, padding, TTA flips, quantisation and the point feature are
computed by csrc/preprocess.hip.  No host synchronisation: the number of in-range points stays on the device (the
un-padding of the labels is a kernel too).
"""
import numpy as np
import torch

from . import _lib, preprocess


class DevicePreprocessor:
    def __init__(self, device, spec=None, frame_point_num=160000, tta=True):
        self.device = torch.device(device)
        self.spec = spec or preprocess.VoxelSpec()
        self.N = int(frame_point_num)
        signs = preprocess.TTA_SIGNS if tta else preprocess.TTA_SIGNS[:1]
        self.V = len(signs)
        self._sx = _lib.f32_array([s[0] for s in signs])
        self._sy = _lib.f32_array([s[1] for s in signs])
        sp = self.spec
        self._range6 = _lib.f64_array([sp.range_x[0], sp.range_x[1], sp.range_y[0], sp.range_y[1], sp.range_z[0], sp.range_z[1]])
        self._bev3 = _lib.i64_array(sp.bev_shape)
        h, w = sp.rv_shape
        phi_hi, phi_lo = 180.0 * np.pi / 180.0, -180.0 * np.pi / 180.0          # datasets/utils.py:176
        th_lo, th_hi = sp.RV_theta[0] * np.pi / 180.0, sp.RV_theta[1] * np.pi / 180.0
        self._rv4 = _lib.f64_array([phi_hi, (phi_hi - phi_lo) / w, th_hi, (th_hi - th_lo) / h])

    def build(self, scans, pose_diffs, strict=False):
        """scans: list of T device tensors [n_t, 4] float32, current scan first; pose_diffs: list of T 4x4 float64
        arrays (inv(P_cur) * P_t; None / identity for the current scan).  Returns the infer() inputs plus what is
        needed to un-pad the current scan's labels.

        A scan with >= frame_point_num in-range points does not fit (the reference asserts pad_length > 0,
        datasets/data_StreamMOS.py:563-566; the host path raises in preprocess.pad_scan).  The kernels never write out
        of bounds -- surplus points are dropped -- and the per-scan in-range counts stay on the device in
        ``in_range_counts``; ``check_capacity`` turns them into the host path's error.  strict=True checks at once
        (one stream synchronisation); the streaming runner checks when it reads the labels back anyway."""
        lib = _lib.load()
        T, N, V = len(scans), self.N, self.V
        dev = self.device
        xyzi = torch.empty((V, T, 7, N, 1), dtype=torch.float32, device=dev)
        coord = torch.empty((V, T, N, 3, 1), dtype=torch.float32, device=dev)
        sphere = torch.empty((V, T, N, 2, 1), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        first = None
        counts = torch.zeros(T, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            for t, (scan, pose) in enumerate(zip(scans, pose_diffs)):
                if not (scan.is_cuda and scan.dtype == torch.float32 and scan.is_contiguous() and scan.shape[1] == 4):
                    raise RuntimeError("DevicePreprocessor: scans must be contiguous float32 [n, 4] GPU tensors")
                n = scan.shape[0]
                moved = torch.empty_like(scan)
                mask = torch.empty(n, dtype=torch.int32, device=dev)
                pd = None
                if pose is not None and not np.array_equal(np.asarray(pose), np.eye(4)):
                    pd = _lib.f64_array(np.asarray(pose, dtype=np.float64).reshape(-1)[:16])
                _lib.check(lib.smos_prep_transform_mask(scan.data_ptr(), n, pd, self._range6, moved.data_ptr(), mask.data_ptr(),
                                                        stream), "smos_prep_transform_mask")
                prefix = torch.cumsum(mask, 0, dtype=torch.int32)
                _lib.check(lib.smos_prep_emit(moved.data_ptr(), mask.data_ptr(), prefix.data_ptr(), n, t, T, N, V, self._sx,
                                              self._sy, self._range6, self._bev3, self._rv4, xyzi.data_ptr(), coord.data_ptr(),
                                              sphere.data_ptr(), stream), "smos_prep_emit")
                if n > 0:
                    counts[t:t + 1].copy_(prefix[-1:])
                if t == 0:
                    first = (mask, prefix, n)
        built = {"pcds_xyzi": xyzi, "pcds_coord": coord, "pcds_sphere_coord": sphere, "mask": first[0], "prefix": first[1],
                 "n_raw": first[2], "in_range_counts": counts, "n_live": counts[0:1]}
        if strict:
            self.check_capacity(built)
        return built

    def check_capacity(self, built):
        """Raises like preprocess.pad_scan when a scan of `built` left no padding (synchronises the stream)."""
        counts = built["in_range_counts"].cpu().tolist()
        for t, c in enumerate(counts):
            if c >= self.N:
                raise ValueError("scan %d of the window has %d in-range points, frame_point_num=%d leaves no padding "
                                 "(the reference asserts pad_length > 0)" % (t, c, self.N))

    def unpad_labels(self, labels, built):
        """labels [N] uint8 of the padded sample -> [n_raw] uint8 for the raw scan (0 where out of range)."""
        lib = _lib.load()
        raw = torch.empty(built["n_raw"], dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.smos_prep_unpad_labels(labels.data_ptr(), labels.shape[0], built["mask"].data_ptr(),
                                                  built["prefix"].data_ptr(), built["n_raw"], raw.data_ptr(),
                                                  torch.cuda.current_stream(self.device).cuda_stream), "smos_prep_unpad_labels")
        return raw
