"""Import shim for the reference's own Python modules (TEST INFRASTRUCTURE ONLY).

The reference tree (/root/reference) is pure Python but cannot be imported as a
package: its __init__ files pull in hydra, isaacgym (closed source), gym and
imageio, none of which exist in this image.  This shim registers empty stub
modules for those names and namespace-style packages (``__path__`` set, the
reference's ``__init__.py`` files never executed) so that the individual
modules holding the ``@torch.jit.script`` functions can be imported by path.

It is used only by ``oracle/gen_golden.py`` (in the build container, where
/root/reference exists) to produce the golden vectors under ``tests/golden``.
Nothing in the product package, in ``bench.py`` or in the GPU tests imports it.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("IGX_REFERENCE_ROOT", "/root/reference")
IGE = os.path.join(REF_ROOT, "IsaacGymEnvs", "isaacgymenvs")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _ns(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Register stubs + namespace packages; idempotent."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(IGE):
        raise RuntimeError("reference tree not found at %s" % IGE)
    import torch

    # the task classes turn the profiling executor off (vec_task.py:250-251)
    torch._C._jit_set_profiling_mode(False)
    torch._C._jit_set_profiling_executor(False)

    class _Any:
        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    for name in ("isaacgym", "isaacgym.gymtorch", "isaacgym.gymapi", "isaacgym.gymutil",
                 "imageio", "cv2", "gym", "gym.spaces"):
        _stub(name)
    sys.modules["isaacgym"].gymtorch = sys.modules["isaacgym.gymtorch"]
    sys.modules["isaacgym"].gymapi = sys.modules["isaacgym.gymapi"]
    sys.modules["isaacgym"].gymutil = sys.modules["isaacgym.gymutil"]
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]
    sys.modules["gym.spaces"].Box = _Any()
    sys.modules["gym"].Space = object

    _ns("isaacgymenvs", IGE)
    _ns("isaacgymenvs.utils", os.path.join(IGE, "utils"))
    _ns("isaacgymenvs.tasks", os.path.join(IGE, "tasks"))
    _ns("isaacgymenvs.tasks.base", os.path.join(IGE, "tasks", "base"))
    _ns("isaacgymenvs.tasks.amp", os.path.join(IGE, "tasks", "amp"))
    _ns("isaacgymenvs.tasks.amp.utils_amp", os.path.join(IGE, "tasks", "amp", "utils_amp"))

    class VecTask:  # dummy base so `class Ant(VecTask)` evaluates
        pass

    _stub("isaacgymenvs.tasks.base.vec_task", VecTask=VecTask)
    _installed = True


def load_real_vec_task():
    """The reference's own tasks/base/vec_task.py under a private module name (the
    task modules keep importing the dummy VecTask registered above)."""
    import importlib.util
    install()
    name = "_ref_vec_task_real"
    if name not in sys.modules:
        # annotations such as gymapi.SimParams are evaluated at class creation
        sys.modules["isaacgym.gymapi"].__getattr__ = lambda k: object
        spec = importlib.util.spec_from_file_location(name, os.path.join(IGE, "tasks", "base", "vec_task.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
    return sys.modules[name]


def load(modname):
    """Import e.g. 'isaacgymenvs.tasks.ant' from the reference tree."""
    install()
    return importlib.import_module(modname)


if __name__ == "__main__":
    for m in ("isaacgymenvs.utils.torch_jit_utils", "isaacgymenvs.tasks.cartpole",
              "isaacgymenvs.tasks.ant", "isaacgymenvs.tasks.humanoid",
              "isaacgymenvs.tasks.allegro_hand", "isaacgymenvs.tasks.amp.humanoid_amp_base",
              "isaacgymenvs.tasks.humanoid_amp", "isaacgymenvs.tasks.amp.utils_amp.motion_lib"):
        mod = load(m)
        print("ok", m, getattr(mod, "__file__", None))


def install_learning_stubs():
    """Enough of rl_games / tensorboardX (neither is in this image nor in the reference tree) for the
    reference's own utils/rlgames_utils.py and learning/common_agent.py to be imported: empty base
    classes and placeholder attributes only - none of the code under test lives in them.  The observer
    (RLGPUAlgoObserver) and the agent loop (CommonAgent.play_steps) that then run are the reference's."""
    install()
    from unittest import mock

    class _Base:
        def __init__(self, *a, **k):
            pass

    def mod(name, **attrs):
        m = sys.modules.get(name) or _stub(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    placeholder = mock.MagicMock()
    mod("rl_games")
    mod("rl_games.common", env_configurations=placeholder, vecenv=mock.MagicMock(IVecEnv=_Base))
    mod("rl_games.common.env_configurations", configurations={})
    mod("rl_games.common.vecenv", IVecEnv=_Base)
    mod("rl_games.common.algo_observer", AlgoObserver=_Base)
    mod("rl_games.common.a2c_common", A2CBase=_Base, swap_and_flatten01=lambda x: x)
    mod("rl_games.common.datasets", PPODataset=_Base)
    mod("rl_games.common.schedulers")
    mod("rl_games.algos_torch")
    mod("rl_games.algos_torch.a2c_continuous", A2CAgent=_Base)
    mod("rl_games.algos_torch.torch_ext")
    mod("rl_games.algos_torch.central_value")
    mod("rl_games.algos_torch.running_mean_std", RunningMeanStd=_Base)
    for child in ("env_configurations", "vecenv", "algo_observer", "a2c_common", "datasets", "schedulers"):
        setattr(sys.modules["rl_games.common"], child, sys.modules["rl_games.common." + child])
    for child in ("a2c_continuous", "torch_ext", "central_value", "running_mean_std"):
        setattr(sys.modules["rl_games.algos_torch"], child, sys.modules["rl_games.algos_torch." + child])
    mod("tensorboardX", SummaryWriter=_Base)
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]
    sys.modules["gym.spaces"].Space = object  # appears in annotations only
    sys.modules["isaacgymenvs.tasks"].isaacgym_task_map = {}
    if "isaacgymenvs.learning" not in sys.modules:
        _ns("isaacgymenvs.learning", os.path.join(IGE, "learning"))
