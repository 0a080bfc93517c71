"""Drop-in module name of the reference (cnn.py) -> MI355X implementation in lcgan_amd.cnn."""
from lcgan_amd.cnn import *  # noqa: F401,F403
