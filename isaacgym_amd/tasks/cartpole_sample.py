"""CartpoleSample: the fork's trajectory-sampling variant of Cartpole (reference
tasks/cartpole_sample.py).  Same fused step as Cartpole with the variant reward
(:316-335); in sample mode (`env.sample`) cart position and pole angle of every
environment are streamed to the host at every step and written as
./data/seq_{name}-{seed}-{numEnvs}-{max_frame_length}.pt by `save_sequence()`.
Camera rendering (`use_cam`, `is_train: False`) belongs to the simulator and has no
counterpart here."""
import os

from ..recorder import TrajectoryRecorder
from .cartpole import Cartpole


class CartpoleSample(Cartpole):
    REWARD_VARIANT = 1

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=0, headless=True, virtual_screen_capture=False,
                 force_render=False, sim=None, seed=0):
        env = cfg["env"]
        self.is_train = env.get("is_train", True)
        self.sample = env.get("sample", False)
        self.max_frame_length = env.get("max_frame_length", 900)
        self.sample_seed = env.get("seed", 42)
        super().__init__(cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture,
                         force_render, sim=sim, seed=seed)
        self.frame_no = 0
        self.sequence_path = os.path.join("./data", "seq_{}-{}-{}-{}.pt".format(
            cfg.get("name", "CartpoleSample"), self.sample_seed, self.num_environments, self.max_frame_length))
        self._recorder = None
        if self.sample:
            self._recorder = TrajectoryRecorder(self.num_envs, 4, dict(x=0, theta=2), self.max_frame_length + 1,
                                                self.device)
            self.compute_observations()
            self._recorder.capture(self.obs_buf)  # frame 0: the state before the first step

    def compute_observations(self, env_ids=None):  # cartpole_sample.py:214-225
        self.obs_buf.copy_(self.dof_state.view(self.num_envs, 4))
        return self.obs_buf

    def post_physics_step(self):
        super().post_physics_step()
        if self._recorder is not None and self._recorder.steps <= self.max_frame_length:
            self._recorder.capture(self.obs_buf)
            self.frame_no += 1

    @property
    def sequence(self):
        """dict "x", "theta" -> [num_envs, frames] of what has been captured so far."""
        return self._recorder.finish()

    def save_sequence(self, path=None):
        path = path or self.sequence_path
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self._recorder.save(path)
        return path
