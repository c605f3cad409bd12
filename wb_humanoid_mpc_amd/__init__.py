"""MI355X-native multiple-shooting SQP iteration for the ocs2-based whole-body humanoid MPC
(Unitree G1) — host-side mirror of the reference's solver interface over a HIP C-ABI library."""
from .model import G1Model, load_model  # noqa: F401
