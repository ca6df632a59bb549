"""Trajectory recorder (pinned host ring fed from a side stream) and the CartpoleSample
task's sampling mode against the sequence the reference's own sample mode accumulated
(tests/golden/cartpole_sample.npz)."""
import numpy as np
import pytest
import torch

from isaacgym_amd import cfg as C
from isaacgym_amd import tasks
from isaacgym_amd._lib import IgxError
from isaacgym_amd.recorder import TrajectoryRecorder, load_sequence
from isaacgym_amd.sim import ReplaySim
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,width,ring", [(1, 4, 1), (1000, 60, 2), (70001, 105, 4)])
def test_recorder_matches_per_step_host_copies(n, width, ring):
    cols = dict(a=0, b=width - 1, c=width // 2)
    steps = 9
    rec = TrajectoryRecorder(n, width, cols, steps, DEV, ring_slots=ring)
    want = {k: [] for k in cols}
    side = torch.cuda.Stream()
    buf = torch.empty(n, width, device=DEV)
    for t in range(steps):
        stream = side if t % 2 else torch.cuda.current_stream()
        stream.wait_stream(side if stream is not side else torch.cuda.current_stream())
        with torch.cuda.stream(stream):  # captures follow the producer's stream, whichever it is
            buf.copy_(torch.randn(n, width, device=DEV))
            rec.capture(buf)
            for k, c in cols.items():
                want[k].append(buf[:, c].cpu().clone())
    assert rec.steps == steps
    got = rec.finish()
    for k in cols:
        assert got[k].shape == (n, steps) and got[k].dtype == torch.float32
        assert torch.equal(got[k], torch.stack(want[k], dim=1)), k
    with pytest.raises(IgxError):
        rec.capture(buf)  # all steps taken
    rec.close()


@pytest.mark.parametrize("n,width,ring,offset", [(1, 3, 1, 0), (1000, 210, 2, 0), (4097, 105, 3, 0), (333, 7, 2, 1)])
def test_whole_row_recorder_matches_per_step_host_copies(n, width, ring, offset):
    """columns=None: the buffer as it is, [steps, n, width] on the host (slabs whose size is not a multiple of 16
    bytes, a source that is only 4-byte aligned, and more captures than ring slots)."""
    steps = 7
    rec = TrajectoryRecorder(n, width, None, steps, DEV, ring_slots=ring)
    flat = torch.empty(n * width + offset, device=DEV)
    buf = flat[offset:].view(n, width)
    want = []
    for t in range(steps):
        buf.copy_(torch.randn(n, width, device=DEV))
        rec.capture(buf)
        want.append(buf.cpu().clone())
    got = rec.finish()
    assert got.shape == (steps, n, width) and torch.equal(got, torch.stack(want))
    with pytest.raises(IgxError):
        rec.capture(buf)
    rec.close()


def test_recorder_rejects_bad_arguments():
    with pytest.raises(IgxError):
        TrajectoryRecorder(8, 4, dict(x=4), 3, DEV)
    rec = TrajectoryRecorder(8, 4, dict(x=0), 3, DEV)
    with pytest.raises(ValueError):
        rec.capture(torch.zeros(8, 5, device=DEV))
    with pytest.raises(ValueError):
        rec.capture(torch.zeros(8, 4))
    assert rec.finish()["x"].shape == (8, 0)


def test_cartpole_sample_sequence_equals_reference(tmp_path):
    g = H.load_golden("cartpole_sample")
    m = g["meta"]
    N, T = m["N"], m["T"]
    frames = [dict(dof_state=g["in_dof_state"][t]) for t in range(T)]
    sim = ReplaySim(N, DEV, dict(dof_state=g["dof_state0"].reshape(N, 2, 2)), frames)
    cfg = C.task_config("CartpoleSample", N, sample=True, max_frame_length=T, seed=42)
    task = tasks.CartpoleSample(cfg, DEV, DEV, sim=sim)
    task.reset_buf.copy_(torch.from_numpy(g["reset0"]).to(DEV))
    task.progress_buf.copy_(torch.from_numpy(g["progress0"]).to(DEV))
    for t in range(T):
        task.reset_rand = torch.from_numpy(g["rand"][t]).to(DEV)
        _, rew, reset, _ = task.step(torch.zeros(N, 1, device=DEV))
        assert np.array_equal(rew.cpu().numpy().view(np.uint32), g["rew"][t].view(np.uint32))  # variant reward
        assert np.array_equal(reset.cpu().numpy(), g["reset"][t])
    seq = task.sequence
    assert np.array_equal(seq["x"].numpy().view(np.uint32), g["sequence_x"].view(np.uint32))
    assert np.array_equal(seq["theta"].numpy().view(np.uint32), g["sequence_theta"].view(np.uint32))
    path = task.save_sequence(str(tmp_path / "data" / "seq.pt"))
    back = load_sequence(path)
    assert sorted(back) == ["theta", "x"] and back["x"].shape == (N, T + 1)
    assert torch.equal(back["theta"], seq["theta"])
    assert task.sequence_path.endswith("seq_CartpoleSample-42-%d-%d.pt" % (N, T))


@pytest.mark.parametrize("n,ring", [(50, 2), (3000, 1), (200000, 3)])
def test_recorder_batches_partial_flushes_and_ring_wrap(n, ring):
    """Small captures are batched into one host copy per staging slot; a sync in the middle of a
    batch flushes the part taken so far, later captures continue in the same slot, and the ring
    wraps several times.  Every step must land in its own place on the host."""
    width, steps = 3, 333
    rec = TrajectoryRecorder(n, width, dict(x=0, z=2), steps, DEV, ring_slots=ring)
    want_x, want_z = [], []
    buf = torch.empty(n, width, device=DEV)
    for t in range(steps):
        buf.copy_(torch.full((n, width), float(t), device=DEV) + torch.arange(width, device=DEV) * 0.25)
        buf[:, 0] += torch.arange(n, device=DEV) * 1e-3
        rec.capture(buf)
        want_x.append(buf[:, 0].cpu())
        want_z.append(buf[:, 2].cpu())
        if t in (0, 4, 70, 71, 200):   # syncs at arbitrary positions inside a batch
            part = rec.finish()
            assert part["x"].shape == (n, t + 1)
            assert torch.equal(part["x"][:, t], want_x[t]) and torch.equal(part["z"][:, 0], want_z[0])
    got = rec.finish()
    assert torch.equal(got["x"], torch.stack(want_x, 1)) and torch.equal(got["z"], torch.stack(want_z, 1))
    with pytest.raises(IgxError):
        rec.capture(buf)   # max_steps taken
    rec.close()


@pytest.mark.parametrize("blocks", ["0", "3", "64"])
def test_both_copy_paths_land_the_same_bytes(blocks, monkeypatch):
    """IGX_RECORDER_DRAIN_BLOCKS: 0 = the runtime's device-to-host copy, otherwise the narrow copy kernel."""
    monkeypatch.setenv("IGX_RECORDER_DRAIN_BLOCKS", blocks)
    n, width, steps = 4096, 6, 40
    rec = TrajectoryRecorder(n, width, dict(a=1, b=4, c=5), steps, DEV, ring_slots=2)
    gen = torch.Generator(device=DEV).manual_seed(int(blocks))
    frames = [torch.randn(n, width, generator=gen, device=DEV) for _ in range(steps)]
    for f in frames:
        rec.capture(f)
    got = rec.finish()
    rec.close()
    for name, col in (("a", 1), ("b", 4), ("c", 5)):
        assert torch.equal(got[name], torch.stack([f[:, col] for f in frames], 1).cpu()), name
