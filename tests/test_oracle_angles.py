"""The CPU checker on the reference's recording of get_euler_xyz / compute_ant_observations /
compute_humanoid_observations at the remainder wrap (tests/golden/angles_edge.npz): strict comparison."""
import numpy as np

from oracle import oracle
from tests import helpers as H
from tests.test_oracle_math import OracleTJ


def test_oracle_angles_at_the_wrap():
    g = H.load_golden("angles_edge")
    seen = H.angles_edge_check(g, OracleTJ.get_euler_xyz, oracle.ant_step, oracle.humanoid_step,
                               to_buf=lambda a: np.ascontiguousarray(a).copy(), to_np=np.asarray)
    print(seen)
    # the recording does sit on the wrap: hundreds of reference angles are float32(2 pi) itself
    assert seen["euler_roll"] > 100 and seen["euler_yaw"] > 100 and seen["angles_edge ant at 2 pi"] > 200
