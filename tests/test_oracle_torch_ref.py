"""oracle/torch_ref.py (the torch-op CPU baseline of bench.py) against the recording of the reference's own
Ant class (tests/golden/ant.npz): everything it computes is the same ATen program, so every output -
trigonometric columns included - must match bit for bit."""
import numpy as np
import torch

from oracle import torch_ref
from tests import helpers as H


def test_torch_restatement_of_the_ant_step_is_bit_identical_to_the_reference():
    g = H.load_golden("ant")
    m = g["meta"]
    N, T = m["N"], m["T"]
    p = H.loco_params(m, g)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).copy())
    b = dict(root_states=torch.zeros(N, 13), dof_state=torch.zeros(N, 8, 2), sensors=torch.zeros(N, 24),
             actions=torch.zeros(N, 8), initial_root_states=t(g["initial_root_states"]),
             initial_dof_pos=t(g["initial_dof_pos"]), obs=torch.zeros(N, 60), potentials=t(g["potentials0"]),
             prev_potentials=t(g["potentials0"]), up_vec=torch.zeros(N, 3), heading_vec=torch.zeros(N, 3),
             rew=torch.zeros(N), reset=t(g["reset0"]), progress=t(g["progress0"]), randomize=torch.zeros(N, dtype=torch.int64),
             true_objective=torch.zeros(N))
    ref = torch_ref.AntStepRef(p, b, N)
    for k in range(T):
        b["root_states"].copy_(t(g["in_root"][k]))
        b["dof_state"].copy_(t(g["in_dof_state"][k]).view(N, 8, 2))
        b["sensors"].copy_(t(g["in_sensors"][k]))
        b["actions"].copy_(t(g["in_actions"][k]))
        ref.step(rand=t(g["rand"][k]))
        for name in ("obs", "potentials", "prev_potentials", "up_vec", "heading_vec", "rew", "true_objective"):
            assert np.array_equal(b[name].numpy().view(np.uint32), g[name][k].view(np.uint32)), "%s step %d" % (name, k)
        for name in ("reset", "progress", "randomize"):
            assert np.array_equal(b[name].numpy(), g[name][k]), "%s step %d" % (name, k)
        assert np.array_equal(b["dof_state"].numpy().reshape(-1, 2).view(np.uint32), g["out_dof_state"][k].view(np.uint32))
