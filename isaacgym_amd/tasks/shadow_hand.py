"""ShadowHand task: mirror of the reference's tasks/shadow_hand.py (class ShadowHand)
on two fused HIP calls per control step.

`pre_physics_step(actions)`  = goal / environment resets of the flagged envs
(reset_target_pose :586-602, reset_idx :604-668) followed by the action path over
the 20 actuated joints (:684-698); `post_physics_step()` = progress / randomize
counters, the chosen observation layout (:437-584), compute_hand_reward (:747-800)
including the global consecutive_successes average, with no host synchronisation.

`reset_idx(env_ids, goal_env_ids)` only raises the flags: the reference calls it from
pre_physics_step on exactly the flagged ids, which is where the fused kernel applies
them.  Random object forces (forceScale > 0) live in the physics engine and are not
part of this path; with task.randomize set, the observation / action noise of
randomization_params is applied (VecTask.apply_randomizations), the physical
parameters are the engine's.
"""
import torch

from .. import _abi, distributed, ops, synthetic
from .allegro_hand import success_threshold_c
from .base.vec_task import VecTask
from .hand_forces import HandForcesMixin


class ShadowHand(HandForcesMixin, VecTask):
    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=0, headless=True, virtual_screen_capture=False,
                 force_render=False, sim=None, seed=0, process_group=None):
        self.cfg = cfg
        self._group = process_group
        env = cfg["env"]
        self.randomize = cfg["task"]["randomize"]
        self.randomization_params = cfg["task"]["randomization_params"]
        self.dist_reward_scale = env["distRewardScale"]
        self.rot_reward_scale = env["rotRewardScale"]
        self.action_penalty_scale = env["actionPenaltyScale"]
        self.success_tolerance = env["successTolerance"]
        self.reach_goal_bonus = env["reachGoalBonus"]
        self.fall_dist = env["fallDistance"]
        self.fall_penalty = env["fallPenalty"]
        self.rot_eps = env["rotEps"]
        self.vel_obs_scale = 0.2
        self.force_torque_obs_scale = 10.0
        self.reset_position_noise = env["resetPositionNoise"]
        self.reset_rotation_noise = env["resetRotationNoise"]
        self.reset_dof_pos_noise = env["resetDofPosRandomInterval"]
        self.reset_dof_vel_noise = env["resetDofVelRandomInterval"]
        self.force_scale = env.get("forceScale", 0.0)
        self.shadow_hand_dof_speed_scale = env["dofSpeedScale"]
        self.use_relative_control = env["useRelativeControl"]
        self.act_moving_average = env["actionsMovingAverage"]
        self.debug_viz = env["enableDebugVis"]
        self.max_episode_length = env["episodeLength"]
        self.reset_time = env.get("resetTime", -1.0)
        self.print_success_stat = env["printNumSuccesses"]
        self.max_consecutive_successes = env["maxConsecutiveSuccesses"]
        self.av_factor = env.get("averFactor", 0.1)
        self.object_type = env["objectType"]
        assert self.object_type in ["block", "egg", "pen"]
        self.ignore_z = self.object_type == "pen"
        self.obs_type = env["observationType"]
        if self.obs_type not in _abi.IGX_SHADOW_OBS:
            raise Exception("Unknown type of observations!\nobservationType should be one of: "
                            "[openai, full_no_vel, full, full_state]")
        self.num_obs_dict = dict(_abi.IGX_SHADOW_NUM_OBS)
        self.up_axis = "z"
        self.num_fingertips = _abi.IGX_SHADOW_FINGERTIPS
        self.asymmetric_obs = env["asymmetric_observations"]
        env["numObservations"] = self.num_obs_dict[self.obs_type]
        env["numStates"] = 211 if self.asymmetric_obs else 0
        env["numActions"] = _abi.IGX_SHADOW_ACTIONS
        super().__init__(cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture,
                         force_render, sim=sim)
        if self.randomize:  # create_sim :212-213
            self.apply_randomizations(self.randomization_params)
        if self.reset_time > 0.0:
            self.max_episode_length = int(round(self.reset_time / (self.control_freq_inv * self.dt)))
        dev, n = self.device, self.num_envs
        self.num_shadow_hand_dofs = self.num_dofs = _abi.IGX_SHADOW_DOF
        asset = self.sim.asset
        self.shadow_hand_dof_lower_limits = torch.tensor(asset["dof_limits_lower"], device=dev, dtype=torch.float)
        self.shadow_hand_dof_upper_limits = torch.tensor(asset["dof_limits_upper"], device=dev, dtype=torch.float)
        self.actuated_dof_indices = torch.tensor(asset["actuated_dof_indices"], device=dev, dtype=torch.long)
        self.fingertip_handles = torch.tensor(asset["fingertip_handles"], device=dev, dtype=torch.long)
        self.dof_state = self.sim.acquire("dof_state")                      # [N,24,2]
        self.shadow_hand_dof_state = self.dof_state.view(n, -1, 2)[:, :self.num_shadow_hand_dofs]
        self.shadow_hand_dof_pos = self.shadow_hand_dof_state[..., 0]
        self.shadow_hand_dof_vel = self.shadow_hand_dof_state[..., 1]
        self._full = self.obs_type == "full_state" or self.asymmetric_obs
        if self._full:
            self.vec_sensor_tensor = self.sim.acquire("vec_sensor_tensor").view(n, self.num_fingertips * 6)
            self.dof_force_tensor = self.sim.acquire("dof_force").view(n, self.num_shadow_hand_dofs)
        self.rigid_body_states = self.sim.acquire("rigid_body_states").view(n, -1, 13)
        self.num_bodies = self.rigid_body_states.shape[1]
        self.root_state_tensor = self.sim.acquire("root_state_tensor").view(-1, 13)   # [3N,13]: hand, object, goal
        self.object_indices = self.sim.acquire("object_indices")
        self.goal_object_indices = self.sim.acquire("goal_object_indices")
        self.object_init_state = self.sim.acquire("object_init_state")
        self.goal_init_state = self.sim.acquire("goal_init_state")
        self.goal_states = self.sim.acquire("goal_states")
        self.goal_displacement_tensor = torch.tensor([-0.2, -0.06, 0.12], device=dev)
        self.prev_targets = torch.zeros((n, self.num_dofs), dtype=torch.float, device=dev)
        self.cur_targets = torch.zeros((n, self.num_dofs), dtype=torch.float, device=dev)
        self.reset_goal_buf = self.reset_buf.clone()
        self.successes = torch.zeros(n, dtype=torch.float, device=dev)
        self.consecutive_successes = torch.zeros(1, dtype=torch.float, device=dev)
        self._cons_scalar = self.consecutive_successes.view(())
        self.actions = torch.zeros(n, _abi.IGX_SHADOW_ACTIONS, device=dev)
        self._action_tensor = self.actions  # VecTask.step's clamp writes the task's own action tensor: no second copy
        self.total_successes = 0
        self.total_resets = 0
        self.seed = int(seed)
        self.reset_rand = None  # optional [N,61] pre-drawn uniforms
        self._draws = 0
        self._workspace = torch.empty(synthetic.allegro_workspace_floats(n), device=dev)  # reduction scratch
        tol = self.success_tolerance * (2.0 if self.ignore_z else 1.0)
        self._success_c = success_threshold_c(tol)
        av32 = float(torch.tensor(self.av_factor, dtype=torch.float32))
        self._one_minus_av = 1.0 - av32
        self._init_forces(env)  # rb_forces, random_force_prob (forceScale > 0: random object forces)

    _PARAM_KEY = ("obs_type", "max_consecutive_successes", "ignore_z", "use_relative_control", "dist_reward_scale",
                  "rot_reward_scale", "rot_eps", "action_penalty_scale", "success_tolerance", "reach_goal_bonus", "fall_dist",
                  "fall_penalty", "max_episode_length", "av_factor", "vel_obs_scale", "force_torque_obs_scale",
                  "reset_position_noise", "reset_dof_pos_noise", "reset_dof_vel_noise", "act_moving_average",
                  "shadow_hand_dof_speed_scale", "dt", "seed")

    def _build_params(self):
        ama = float(self.act_moving_average)
        asset = self.sim.asset
        return _abi.ShadowParams(
            obs_type=_abi.IGX_SHADOW_OBS[self.obs_type], max_consecutive_successes=self.max_consecutive_successes,
            ignore_z_rot=int(self.ignore_z), use_relative_control=int(bool(self.use_relative_control)),
            dist_reward_scale=self.dist_reward_scale, rot_reward_scale=self.rot_reward_scale, rot_eps=self.rot_eps,
            action_penalty_scale=self.action_penalty_scale, success_tolerance=self.success_tolerance,
            success_threshold_c=self._success_c, reach_goal_bonus=float(self.reach_goal_bonus),
            fall_dist=self.fall_dist, fall_penalty=self.fall_penalty,
            max_episode_length_m1=float(self.max_episode_length - 1), av_factor=self.av_factor,
            one_minus_av_factor=self._one_minus_av, vel_obs_scale=self.vel_obs_scale,
            force_torque_obs_scale=self.force_torque_obs_scale, reset_position_noise=self.reset_position_noise,
            reset_dof_pos_noise=self.reset_dof_pos_noise, reset_dof_vel_noise=self.reset_dof_vel_noise,
            act_moving_average=ama, one_minus_act_moving_average=1.0 - ama,
            dof_speed_scale_dt=self.shadow_hand_dof_speed_scale * self.dt,
            goal_displacement=[-0.2, -0.06, 0.12], num_bodies=self.num_bodies,
            dof_lower=asset["dof_limits_lower"], dof_upper=asset["dof_limits_upper"],
            actuated_dof_indices=[int(j) for j in asset["actuated_dof_indices"]],
            fingertip_handles=[int(j) for j in asset["fingertip_handles"]], seed=self.seed)

    def _buffers(self):
        return dict(dof_state=self.dof_state, dof_force=self.dof_force_tensor if self._full else None,
                    vec_sensor_tensor=self.vec_sensor_tensor if self._full else None,
                    rigid_body_states=self.rigid_body_states,
                    root_state_tensor=self.root_state_tensor, object_indices=self.object_indices,
                    goal_object_indices=self.goal_object_indices, goal_states=self.goal_states,
                    goal_init_state=self.goal_init_state, object_init_state=self.object_init_state,
                    actions=self.actions, prev_targets=self.prev_targets, cur_targets=self.cur_targets,
                    rand=self.reset_rand, obs=self.obs_buf, states=self.states_buf if self.asymmetric_obs else None,
                    rew=self.rew_buf, reset=self.reset_buf, reset_goal=self.reset_goal_buf,
                    progress=self.progress_buf, randomize=self.randomize_buf, successes=self.successes,
                    consecutive_successes=self.consecutive_successes, workspace=self._workspace)

    def reset_target_pose(self, env_ids, apply_reset=False):
        self.reset_goal_buf[env_ids] = 1

    def reset_idx(self, env_ids, goal_env_ids=None):
        self.reset_buf[env_ids] = 1
        if goal_env_ids is not None and len(goal_env_ids) > 0:
            self.reset_goal_buf[goal_env_ids] = 1

    FUSED_ACTION_STAGE = True  # step() leaves the clamp of the actions to the pre-physics launch

    def pre_physics_step(self, actions):
        # :606-607: the reference re-randomises inside reset_idx, i.e. on steps where some env is reset
        if self.randomize and bool(self.reset_buf.any()):
            self.apply_randomizations(self.randomization_params)
        stage, self._stage = self._stage, None
        p, b = self._params(), self._buffers()
        if stage is not None and "noise" not in stage:   # called from step(): raw actions, clamp folded into the kernel
            b["actions"], b["actions_out"] = actions, self.actions
            p.clip_actions = float(stage["clip_actions"])
        elif stage is not None:                          # action noise configured: noise, then the clamp
            noised = ops.noise_apply(stage["noise"], actions, fresh=stage.get("fresh"))
            ops.clamp(noised, -stage["clip_actions"], stage["clip_actions"], self.actions)
        elif actions is not self.actions:                # a direct call: the actions are taken as they are
            self.actions.copy_(actions.reshape(self.actions.shape))
        self._apply_random_forces()  # reads reset_buf before the kernel below clears it
        ops.shadow_pre_physics(p, b, self.num_envs, self._call_cache)
        self.sim.notify_root_state(self.root_state_tensor)
        self.sim.notify_dof_state()
        self.sim.apply_dof_position_targets(self.cur_targets)

    def _apply_shard_records(self, records, num_steps):
        """consecutive_successes over a queue of gathered per-step records (runs on the queue's side stream)."""
        ops.shadow_finalize_shards(self._build_params(), dict(consecutive_successes=self._shard_queue.pending), records,
                                    num_steps)

    def _sharded_post(self, p, b):
        """One shard of a multi-GPU run.  The step itself exchanges nothing: its 16-byte record waits in a
        device-side queue.  Default (`statsSyncInterval` 0 in the env config): the records of all ranks are
        gathered on the step's own stream every step and consecutive_successes is current after every step,
        as in the reference.  With `statsSyncInterval: K` > 0 (opt-in) the queues of all ranks are gathered
        once per K steps with one collective on a side stream and consecutive_successes is advanced over them
        in step order (distributed.ShardRecordQueue): same bits, but what the task shows lags by up to two
        intervals until sync_consecutive_successes() is called."""
        interval = int(self.cfg["env"].get("statsSyncInterval", 0))
        if interval <= 0:
            ops.shadow_post_physics_shard(p, b, self.num_envs, self._call_cache)
            records = distributed.gather_records(self._workspace.view(torch.uint8)[:16], self._group)
            ops.shadow_finalize_shards(p, b, records)
            return
        q = self.__dict__.get("_shard_queue")
        if q is None:
            q = self._shard_queue = distributed.ShardRecordQueue(self.consecutive_successes, interval, self._group)
        b["shard_record"] = q.slot()
        ops.shadow_post_physics_shard(p, b, self.num_envs, self._call_cache)
        if q.push():
            q.flush(self._apply_shard_records)

    def sync_consecutive_successes(self):
        """Bring consecutive_successes up to the last step (sharded runs with statsSyncInterval > 0 keep it up to
        two sync intervals behind)."""
        q = self.__dict__.get("_shard_queue")
        if q is not None:
            q.drain(self._apply_shard_records)
        return self.consecutive_successes

    def post_physics_step(self):
        p, b = self._params(), self._buffers()
        if self._can_fuse_tail(states_too=True):  # clamped copies and time-out flags of VecTask.step in the same launch
            self._fuse_tail(p, b)
        if distributed.world_size(self._group) > 1:
            self._sharded_post(p, b)
        else:
            ops.shadow_post_physics(p, b, self.num_envs, self._call_cache)
        # .mean() of the one-element tensor (:393) is that element: a 0-dim view made once, no launch per step
        # (with env.freshStepOutputs a copy, like everything else step() hands out: a learner may keep extras)
        self.extras["consecutive_successes"] = self._cons_scalar.clone() if self.fresh_step_outputs else self._cons_scalar
        if self.print_success_stat:  # :426-435 (host prints: synchronises, as in the reference)
            self.total_resets = self.total_resets + self.reset_buf.sum()
            direct = self.total_successes + self.successes.sum()
            self.total_successes = self.total_successes + (self.successes * self.reset_buf).sum()
            print("Direct average consecutive successes = {:.1f}".format(direct / (self.total_resets + self.num_envs)))
            if self.total_resets > 0:
                print("Post-Reset average consecutive successes = {:.1f}".format(self.total_successes / self.total_resets))
