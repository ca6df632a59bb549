"""tasks/amp/humanoid_amp_base.py of the reference: class HumanoidAMPBase and its constants."""
from ..humanoid_amp import BODY_NAMES, KEY_BODY_NAMES, NUM_ACTIONS, NUM_OBS, HumanoidAMPBase  # noqa: F401
from .utils_amp.motion_lib import DOF_BODY_IDS, DOF_OFFSETS  # noqa: F401
