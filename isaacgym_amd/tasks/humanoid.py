"""tasks/humanoid.py of the reference: class Humanoid."""
from .locomotion import Humanoid  # noqa: F401
