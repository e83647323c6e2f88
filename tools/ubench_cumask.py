"""Dev: give the two pipeline stages their own CUs.  The encoder of frame t+1 (side stream) and the
decoder of frame t (main stream) run on HIP streams created with hipExtStreamCreateWithCUMask; scans/s of the bench step for
several splits, next to the unmasked two-stream pipeline.  Two mask layouts are tried, because the bit -> (XCD, CU) mapping
is not documented here: "rr" assumes bit i = CU i // 8 of XCD i % 8 (round-robin over the XCDs), "blk" assumes bit i = CU
i % 32 of XCD i // 32; p of every XCD's 32 CUs go to the side stream, the rest to the main one."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from streammos_amd import streaming, synth
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.models import StreamMOS

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    words = [0] * 8
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    arr = (ctypes.c_uint32 * 8)(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


model = StreamMOS.AttNet(cfg.get_config()[2])
model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
frames = bench.make_frames(6, seq_seed=0)


def run(main_s, side_s, steps=30, warm=8):
    runner = streaming.StreamRunner(model, dev, vote=True, pipeline=True)
    if side_s is not None:
        runner._side = side_s
    ctx = torch.cuda.stream(main_s) if main_s is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        devf = [(runner.upload(s, raw), pose) for s, raw, pose in frames]
        for i in range(warm):
            runner.step(devf[i % 6][0], devf[i % 6][1], next_dev=devf[(i + 1) % 6][0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            runner.step(devf[i % 6][0], devf[i % 6][1], next_dev=devf[(i + 1) % 6][0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    runner.close()
    return steps / dt


print("unmasked two-stream pipeline: %.1f scans/s" % run(None, None), flush=True)
allb = list(range(256))
print("both stages on streams masked to ALL CUs: %.1f scans/s" % run(masked_stream(allb), masked_stream(allb)), flush=True)
for layout in ("rr", "blk"):
    for p in (8, 12, 16):
        side = [i for i in allb if ((i // 8) if layout == "rr" else (i % 32)) < p]
        main = [i for i in allb if i not in side]
        print("%s: side %d CUs/XCD (%d), main %d: %.1f scans/s" % (layout, p, len(side), len(main),
                                                                   run(masked_stream(main), masked_stream(side))), flush=True)
for nx in (2, 3):
    side = [i for i in allb if i % 8 < nx]
    main = [i for i in allb if i not in side]
    print("rr whole XCDs: side %d XCDs, main %d: %.1f scans/s" % (nx, 8 - nx, run(masked_stream(main), masked_stream(side))), flush=True)
    side = [i for i in allb if i // 32 < nx]
    main = [i for i in allb if i not in side]
    print("blk whole XCDs: side %d XCDs, main %d: %.1f scans/s" % (nx, 8 - nx, run(masked_stream(main), masked_stream(side))), flush=True)
