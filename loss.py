"""Drop-in module name of the reference (loss.py) -> MI355X implementation in lcgan_amd.loss."""
from lcgan_amd.loss import *  # noqa: F401,F403
