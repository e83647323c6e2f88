"""Dev: full-size end-to-end run of run_sequence on a synthetic SemanticKITTI-layout sequence (120k-point scans,
frame_point_num 160000), stage-2 model + instance voting; prints the IoU report and the wall time per scan."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streammos_amd import kitti, run_sequence, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
root = tempfile.mkdtemp(prefix="smos_seq_")
seq = os.path.join(root, "sequences", "08")
os.makedirs(os.path.join(seq, "velodyne")); os.makedirs(os.path.join(seq, "labels"))
for k in range(n):
    scan, lab = synth.synthetic_scan(k, with_labels=True)
    scan.tofile(os.path.join(seq, "velodyne", "%06d.bin" % k))
    kitti.write_prediction(os.path.join(seq, "labels", "%06d.label" % k), lut_labels=np.where(lab == 2, 251, 9).astype(np.uint32))
kitti.write_poses(os.path.join(seq, "poses.txt"), [synth.synthetic_pose(k) for k in range(n)])
kitti.write_calibration(os.path.join(seq, "calib.txt"))
for seg, vote, devpre in ((False, True, False), (True, "instance", False), (True, "instance", True)):
    model = run_sequence.load_model(None, "cuda:0", seg=seg)
    t = time.time()
    res = run_sequence.run_sequence(model, seq, os.path.join(root, "out_%d%d" % (seg, devpre)), "cuda:0", vote=vote, device_preprocess=devpre)
    dt = time.time() - t
    print("seg=%s vote=%s device_preprocess=%s  %.1f ms/scan (disk IO included)" % (seg, vote, devpre, 1e3 * dt / n), res, flush=True)
import numpy as np
a = [np.fromfile(os.path.join(root, "out_10", "refined", "%06d.label" % k), dtype=np.uint32) for k in range(n)]
b = [np.fromfile(os.path.join(root, "out_11", "refined", "%06d.label" % k), dtype=np.uint32) for k in range(n)]
print("refined labels host-vs-device preprocessing agreement: %.5f" % np.mean([np.mean(x == y) for x, y in zip(a, b)]))
