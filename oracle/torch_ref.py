"""The Ant post_physics_step as plain torch tensor ops on the host (TEST INFRASTRUCTURE / CPU BASELINE ONLY).

This is the "reference CPU pipeline" stand-in of bench.py's cpu_baseline leg: the reference's Python cannot
travel to the GPU box, so its per-step tensor program (tasks/ant.py:376-386 post_physics_step, :339-366
reset_idx, :463-497 compute_ant_observations, :414-460 compute_ant_reward, with the helpers of
utils/torch_jit_utils.py:41-103, :175-195, :239-277) is written out here from its formulas, one ATen op per
arithmetic step like the original (which runs with the profiling executor off, so nothing is fused).
tests/test_oracle_torch_ref.py checks it bit for bit against tests/golden/ant.npz (the recording of the
reference's own classes).  Nothing in isaacgym_amd/ imports this file.
"""
import math

import torch

TWO_PI = 2 * math.pi


def _qmul(a, b):  # torch_jit_utils.py:41-62, the 8-multiply form
    x1, y1, z1, w1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    x2, y2, z2, w2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([x, y, z, w], dim=-1)


def _rotate(q, v, sign):  # :80-103: sign +1 quat_rotate, -1 quat_rotate_inverse
    n = q.shape[0]
    qw, qv = q[:, -1], q[:, :3]
    a = v * (2.0 * qw ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(qv, v, dim=-1) * qw.unsqueeze(-1) * 2.0
    c = qv * torch.bmm(qv.view(n, 1, 3), v.view(n, 3, 1)).squeeze(-1) * 2.0
    return a + b + c if sign > 0 else a - b + c


def _euler_roll_yaw(q):  # :175-195 (pitch is computed by the reference and never used by Ant)
    qx, qy, qz, qw = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = torch.atan2(2.0 * (qw * qx + qy * qz), qw * qw - qx * qx - qy * qy + qz * qz)
    sinp = 2.0 * (qw * qy - qz * qx)
    _pitch = torch.where(torch.abs(sinp) >= 1, torch.abs(torch.full_like(sinp, math.pi / 2.0)) * torch.sign(sinp),
                         torch.asin(sinp))
    yaw = torch.atan2(2.0 * (qw * qz + qx * qy), qw * qw + qx * qx - qy * qy - qz * qz)
    return roll % TWO_PI, yaw % TWO_PI


class AntStepRef:
    """State and step of N Ant environments; attribute names follow the task class."""

    def __init__(self, p, bufs, n):
        """p: an isaacgym_amd._abi.LocoParams; bufs: name -> torch CPU tensors laid out like igx_loco_buffers."""
        self.n, self.p, self.b = n, p, bufs
        D = 8
        self.lo = torch.tensor(list(p.dof_limits_lower)[:D])
        self.hi = torch.tensor(list(p.dof_limits_upper)[:D])
        self.targets = torch.tensor(list(p.targets)).repeat(n, 1)
        self.inv_start_rot = torch.tensor(list(p.inv_start_rot)).repeat(n, 1)
        self.basis_vec0 = torch.tensor(list(p.basis_vec0)).repeat(n, 1)
        self.basis_vec1 = torch.tensor(list(p.basis_vec1)).repeat(n, 1)

    def reset_idx(self, env_ids, u):  # ant.py:339-366; u [len(env_ids), 16] uniforms in [0, 1)
        b, p = self.b, self.p
        dof = b["dof_state"].view(self.n, 8, 2)
        positions = (0.2 - -0.2) * u[:, :8] + -0.2
        velocities = (0.1 - -0.1) * u[:, 8:] + -0.1
        dof[env_ids, :, 0] = torch.max(torch.min(b["initial_dof_pos"][env_ids] + positions, self.hi), self.lo)
        dof[env_ids, :, 1] = velocities
        to_target = self.targets[env_ids] - b["initial_root_states"][env_ids, 0:3]
        to_target[:, 2] = 0.0
        b["prev_potentials"][env_ids] = -torch.norm(to_target, p=2, dim=-1) / p.dt
        b["potentials"][env_ids] = b["prev_potentials"][env_ids].clone()
        b["progress"][env_ids] = 0
        b["reset"][env_ids] = 0

    def step(self, rand=None):
        b, p, n = self.b, self.p, self.n
        b["progress"] += 1
        b["randomize"] += 1
        env_ids = b["reset"].nonzero(as_tuple=False).squeeze(-1)
        if len(env_ids) > 0:
            u = rand[env_ids] if rand is not None else torch.rand(len(env_ids), 16)
            self.reset_idx(env_ids, u)

        # compute_ant_observations
        root = b["root_states"]
        pos, rot, vel, ang = root[:, 0:3], root[:, 3:7], root[:, 7:10], root[:, 10:13]
        dof = b["dof_state"].view(n, 8, 2)
        to_target = self.targets - pos
        to_target[:, 2] = 0.0
        prev_potentials_new = b["potentials"].clone()
        potentials = -torch.norm(to_target, p=2, dim=-1) / p.dt
        torso_quat = _qmul(rot, self.inv_start_rot)
        up_vec = _rotate(torso_quat, self.basis_vec1, +1).view(n, 3)
        heading_vec = _rotate(torso_quat, self.basis_vec0, +1).view(n, 3)
        up_proj = up_vec[:, 2]
        target_dirs = to_target / to_target.norm(p=2, dim=-1).clamp(min=1e-9, max=None).unsqueeze(-1)
        heading_proj = torch.bmm(heading_vec.view(n, 1, 3), target_dirs.view(n, 3, 1)).view(n)
        vel_loc = _rotate(torso_quat, vel, -1)
        angvel_loc = _rotate(torso_quat, ang, -1)
        roll, yaw = _euler_roll_yaw(torso_quat)
        walk_target_angle = torch.atan2(self.targets[:, 2] - pos[:, 2], self.targets[:, 0] - pos[:, 0])
        angle_to_target = walk_target_angle - yaw
        dof_pos_scaled = (2.0 * dof[:, :, 0] - self.hi - self.lo) / (self.hi - self.lo)
        obs = torch.cat((pos[:, 2].view(-1, 1), vel_loc, angvel_loc, yaw.unsqueeze(-1), roll.unsqueeze(-1),
                         angle_to_target.unsqueeze(-1), up_proj.unsqueeze(-1), heading_proj.unsqueeze(-1), dof_pos_scaled,
                         dof[:, :, 1] * p.dof_vel_scale, b["sensors"].view(n, -1) * p.contact_force_scale, b["actions"]),
                        dim=-1)
        b["obs"][:] = obs
        b["potentials"][:] = potentials
        b["prev_potentials"][:] = prev_potentials_new
        b["up_vec"][:] = up_vec
        b["heading_vec"][:] = heading_vec

        # compute_ant_reward
        obs, actions = b["obs"], b["actions"]
        heading_weight_tensor = torch.ones_like(obs[:, 11]) * p.heading_weight
        heading_reward = torch.where(obs[:, 11] > 0.8, heading_weight_tensor, p.heading_weight * obs[:, 11] / 0.8)
        up_reward = torch.zeros_like(heading_reward)
        up_reward = torch.where(obs[:, 10] > 0.93, up_reward + p.up_weight, up_reward)
        actions_cost = torch.sum(actions ** 2, dim=-1)
        electricity_cost = torch.sum(torch.abs(actions * obs[:, 20:28]), dim=-1)
        dof_at_limit_cost = torch.sum(obs[:, 12:20] > 0.99, dim=-1)
        alive_reward = torch.ones_like(potentials) * 0.5
        progress_reward = b["potentials"] - b["prev_potentials"]
        total = (progress_reward + alive_reward + up_reward + heading_reward - p.actions_cost_scale * actions_cost
                 - p.energy_cost_scale * electricity_cost - dof_at_limit_cost * p.joints_at_limit_cost_scale)
        total = torch.where(obs[:, 0] < p.termination_height, torch.ones_like(total) * p.death_cost, total)
        reset = torch.where(obs[:, 0] < p.termination_height, torch.ones_like(b["reset"]), b["reset"])
        reset = torch.where(b["progress"] >= p.max_episode_length_m1, torch.ones_like(b["reset"]), reset)
        b["rew"][:] = total
        b["reset"][:] = reset
        if b.get("true_objective") is not None:  # ant.py:332-337
            b["true_objective"][:] = root[:, 7]


def configure_like_the_task(num_threads=None):
    """tasks/base/vec_task.py:250-251 turns the TorchScript profiling executor off; here there is no script
    at all (eager ops = the same unfused ATen kernels).  Returns the intra-op thread count in use."""
    if num_threads:
        torch.set_num_threads(int(num_threads))
    return torch.get_num_threads()
