"""Drop-in module name of the reference (ema.py) -> MI355X implementation in lcgan_amd.ema."""
from lcgan_amd.ema import *  # noqa: F401,F403
