"""Drop-in module name of the reference (custom_layers.py) -> MI355X implementation in lcgan_amd.custom_layers."""
from lcgan_amd.custom_layers import *  # noqa: F401,F403
