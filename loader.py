"""Drop-in module name of the reference (loader.py) -> MI355X implementation in lcgan_amd.loader."""
from lcgan_amd.loader import *  # noqa: F401,F403
