"""Task classes with the reference's names (tasks/__init__.py: isaacgym_task_map)."""
from .allegro_hand import AllegroHand
from .ant import Ant
from .ant_imitate import AntImitate
from .cartpole import Cartpole
from .cartpole_sample import CartpoleSample
from .humanoid import Humanoid
from .humanoid_amp import HumanoidAMP
from .humanoid_imitate import HumanoidImitate
from .shadow_hand import ShadowHand

isaacgym_task_map = {
    "AllegroHand": AllegroHand,
    "Ant": Ant,
    "AntImitate": AntImitate,
    "Cartpole": Cartpole,
    "CartpoleSample": CartpoleSample,
    "Humanoid": Humanoid,
    "HumanoidAMP": HumanoidAMP,
    "HumanoidImitate": HumanoidImitate,
    "ShadowHand": ShadowHand,
}


def make(name, num_envs, device="cuda:0", seed=0, sim=None, **env_overrides):
    """Task `name` with the default configuration on a SyntheticSim (or the given sim)."""
    from ..cfg import task_config
    from ..sim import SyntheticSim
    cfg = task_config(name, num_envs, **env_overrides)
    if sim is None:
        extra = {}
        if name in ("AllegroHand", "ShadowHand"):
            extra["obs_type"] = cfg["env"]["observationType"]
        if name == "ShadowHand":
            extra["asymmetric"] = cfg["env"]["asymmetric_observations"]
        if name == "HumanoidAMP":
            extra["num_amp_obs_steps"] = cfg["env"]["numAMPObsSteps"]
        sim = SyntheticSim(name, num_envs, device, seed=seed, **extra)
    kw = {} if name == "HumanoidAMP" else dict(seed=seed)
    if name == "HumanoidImitate" and "motionToImitate" not in cfg:
        import numpy as np
        kw["motion"] = np.random.default_rng(seed).standard_normal((48, 63 + 22 * 13)) * 0.5
    if name == "AntImitate" and "motionToImitate" not in cfg:
        import numpy as np
        from .. import synthetic
        lo, hi = (np.asarray(v, np.float64) for v in synthetic.loco_limits(8))
        kw["motion"] = lo + (hi - lo) * np.random.default_rng(seed).random((64, 8))
    return isaacgym_task_map[name](cfg, device, device, 0, True, False, False, sim=sim, **kw)
