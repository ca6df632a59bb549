"""isaacgym_amd: MI355X-native per-step observation / reward / reset path for
IsaacGymEnvs-style vectorised tasks (Cartpole, Ant, Humanoid, AllegroHand,
HumanoidAMP), the torch_jit_utils quaternion math and the AMP motion library.

The compute runs in hand-written HIP kernels (libigx.so, C ABI in
include/igx.h); this package is the Python host side that mirrors the
reference's task-class interface on top of it.
"""
__version__ = "0.1.0"
