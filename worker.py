"""Drop-in module name of the reference (worker.py) -> MI355X implementation in lcgan_amd.worker."""
from lcgan_amd.worker import *  # noqa: F401,F403
