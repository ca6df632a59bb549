"""CPU checker's HumanoidAMP step / reset against the recording of the
reference task class (tests/golden/amp.npz; motions from motion.npz)."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H


@pytest.mark.parametrize("recording", ["amp", "amp_alt", "amp_nonfinite", "amp_extreme", "amp_nonunit"])  # amp_alt: heading-frame root observation, 3 history steps,
def test_amp_task_oracle_matches_reference(recording):     # other termination height and episode length
    g = H.load_golden(recording)
    tabs = H.motion_tabs_from_golden(H.load_golden("motion"))
    fns = dict(step=oracle.humanoid_amp_step,
               reset=lambda p, b, env_ids, state: oracle.humanoid_amp_reset(p, b, env_ids, state),
               motion_state=lambda ids, times: {k: v for k, v in oracle.motion_state(tabs, ids, times).items()
                                                if k not in ("frame_idx0", "frame_idx1", "blend", "key_pos")},
               motion_amp_obs=lambda ids, times, lro, out, rows, stride, off:
                   oracle.motion_amp_obs(tabs, ids, times, lro, out, rows, stride, off))
    H.amp_replay(g, fns, to_buf=lambda a: np.ascontiguousarray(a).copy(), to_np=lambda a: a,
                 zeros=lambda shape, dt: np.zeros(shape, dt))
