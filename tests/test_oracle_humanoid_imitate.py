"""HumanoidImitate: CPU checker against the recording of the reference task class
(tests/golden/humanoid_imitate.npz)."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H


@pytest.mark.parametrize("recording", ["humanoid_imitate", "humanoid_imitate_extreme"])
def test_humanoid_imitate_oracle_matches_reference(recording):
    H.him_replay(H.load_golden(recording), oracle.humanoid_imitate_step,
                 to_buf=lambda a: np.ascontiguousarray(a).copy(), to_np=lambda a: a,
                 zeros=lambda shape, dt: np.zeros(shape, dt))
