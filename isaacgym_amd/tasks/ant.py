"""tasks/ant.py of the reference: class Ant."""
from .locomotion import Ant  # noqa: F401
