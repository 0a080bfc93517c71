"""Process bootstrap + the training loop -- counterpart of the reference's loader.py:13-82 (train phase)."""
from __future__ import annotations

import json
import os
from datetime import datetime

import torch
import torch.distributed as dist

from . import worker


def multi_gpu_setup(local_rank, args, gpus_per_node, port_number):
    """loader.py:13-19: one process per GPU; backend "nccl" is RCCL on ROCm; rendezvous on 127.0.0.1."""
    if not torch.cuda.is_available():
        raise RuntimeError("lcgan_amd needs a HIP device: there is no CPU fallback for the training step")
    torch.cuda.set_device(local_rank)
    if gpus_per_node > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%s" % str(port_number), rank=local_rank,
                                world_size=gpus_per_node)


def _barrier(w):
    if w.group is not None:
        dist.barrier(w.group)


def train_iteration(gan_worker, args, epoch):
    """One iteration in the reference's order (loader.py:45-54): G step, EMA, D step."""
    gan_worker.requires_grad(gan_worker.generator, True)
    gan_worker.requires_grad(gan_worker.discriminator, False)
    g_loss = gan_worker.train_generator(epoch)
    gan_worker.ema_update(epoch)
    gan_worker.requires_grad(gan_worker.generator, False)
    gan_worker.requires_grad(gan_worker.discriminator, True)
    if epoch >= args.freezeD_start:
        gan_worker.freeze_discriminator(args.freezeD_layer)
    d_loss = gan_worker.train_discriminator(epoch)
    return g_loss, d_loss


def load_worker(local_rank, args, gpus_per_node, port_number):
    multi_gpu_setup(local_rank, args, gpus_per_node, port_number)
    if args.phase == "fake_image_generation":                             # loader.py:95-99
        gan_worker = worker.WORKER(args, local_rank, gpus_per_node)
        gan_worker.load_model()
        _barrier(gan_worker)
        gan_worker.fake_image_generation(num_images=args.num_fakes)
        return
    if args.phase != "train":
        raise NotImplementedError(f"phase {args.phase!r}: FID (needs downloaded InceptionV3 weights) and the PyAV video tooling are "
                                  "outside the accelerated path (SURVEY.md section 2); checkpoints are compatible with the reference's tools")
    with open(os.path.join(args.model_name, "args.txt"), "w") as f:
        json.dump(args.__dict__, f, indent=2)
    gan_worker = worker.WORKER(args, local_rank, gpus_per_node)
    epoch = 0
    start_time = datetime.now()
    epoch_file_path = os.path.join(args.model_name, "epoch.txt")
    if os.path.exists(epoch_file_path):                                   # resume, loader.py:35-42
        with open(epoch_file_path) as f:
            epoch = int(f.read().strip()) + 1
        print("restart training from:", epoch)
        gan_worker.load_model()
        _barrier(gan_worker)

    while epoch <= args.epoch:
        g_loss, d_loss = train_iteration(gan_worker, args, epoch)
        if epoch % args.print_interval == 0:                              # loader.py:56-68
            if local_rank == 0:
                elapsed = str(datetime.now() - start_time).split(".")[0]
                with open(os.path.join(args.model_name, "log.txt"), "w" if epoch == 0 else "a") as f:
                    f.write("epoch:{}, elapsed:{}, g_loss:{:.6f}, d_loss:{:.6f} \n".format(epoch, elapsed, g_loss, d_loss))
            _barrier(gan_worker)
        if epoch % args.save_interval == 0 and epoch > 0:                 # loader.py:75-80
            gan_worker.flush()
            if local_rank == 0:
                gan_worker.save_model()
                with open(epoch_file_path, "w") as f:
                    f.write(str(epoch))
            _barrier(gan_worker)
        epoch += 1
    gan_worker.flush()
    if dist.is_initialized():
        dist.destroy_process_group()
