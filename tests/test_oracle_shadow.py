"""CPU checker's ShadowHand pre/post physics step against golden vectors recorded from the
reference task class (tests/golden/shadow_*.npz: the four observation layouts, asymmetric
states, pen object with max consecutive successes, relative control, action moving average)."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H


@pytest.mark.parametrize("variant", H.SHADOW_VARIANTS)
def test_shadow_oracle_matches_reference(variant):
    g = H.load_golden("shadow_" + variant)
    H.shadow_replay(g, oracle.shadow_pre_physics, oracle.shadow_post_physics,
                    to_buf=lambda a: np.ascontiguousarray(a).copy(), to_np=lambda a: a,
                    zeros=lambda shape, dt: np.zeros(shape, dt), force_fn=oracle.hand_random_forces)
