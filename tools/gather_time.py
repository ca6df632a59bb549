"""Producer-side time of recorder captures that never wait for the ring (fewer captures than one ring holds)."""
import torch
from isaacgym_amd.recorder import TrajectoryRecorder

DEV = "cuda:0"
for n, width, cols in ((1 << 20, 4, dict(x=0, theta=2)), (1 << 20, 60, dict(a=0, b=7, c=30, d=59))):
    rec = TrajectoryRecorder(n, width, cols, 64, DEV, ring_slots=8)
    src = torch.randn(n, width, device=DEV)
    rec.capture(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        rec.capture(src)
    e1.record()
    torch.cuda.synchronize()
    print("n=%d width=%d cols=%d: %.1f us per capture on the producer stream" % (n, width, len(cols), e0.elapsed_time(e1) / 30 * 1e3))
    rec.finish()
    rec.close()
