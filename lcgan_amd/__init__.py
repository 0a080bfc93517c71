"""lcgan_amd -- MI355X-native (gfx950) LC-GAN training step: hand-written HIP kernels behind a C ABI
(include/lcgan_hip.h), bound into the reference's nn.Module / function surface (cnn, custom_layers, loss, ema, worker)."""
from . import config  # noqa: F401

__all__ = ["config", "cnn", "custom_layers", "loss", "ema", "optim", "worker", "kernels", "ops"]
