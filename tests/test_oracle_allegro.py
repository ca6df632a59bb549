"""CPU checker's AllegroHand pre/post physics step against golden vectors
recorded from the reference task class (all observation layouts, pen object,
relative control, action moving average, max consecutive successes)."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H


@pytest.mark.parametrize("variant", H.ALLEGRO_VARIANTS)
def test_allegro_oracle_matches_reference(variant):
    g = H.load_golden("allegro_" + variant)
    H.allegro_replay(g, oracle.allegro_pre_physics, oracle.allegro_post_physics,
                     to_buf=lambda a: np.ascontiguousarray(a).copy(), to_np=lambda a: a,
                     zeros=lambda shape, dt: np.zeros(shape, dt), force_fn=oracle.hand_random_forces)
