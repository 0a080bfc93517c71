"""Per-rank trainer -- counterpart of the reference's worker.py:30-253 (training part) on the HIP kernels.

Kept: class name, constructor signature, `train_generator` / `train_discriminator` / `ema_update` / `freeze_discriminator` /
`requires_grad` / `save_model` / `load_model`, attribute names (`generator`, `discriminator`, `generator_ema`, `g_optimizer`,
`d_optimizer`, `local_batch_size`), the order of random draws, loss assembly and the `module.`-prefixed checkpoints.
Data: `--dataset_path synthetic` feeds uniform [-1,1] tensors of the dataset's shape/range (custom_dataset.py:81-86); any other
path is an image folder `<path>/train/<class>/*` read with PIL on the host (lcgan_amd/data.py) whose geometry / appearance views are
generated on the device.  Inference: `fake_image_generation` (worker.py:427-441).  Out of scope: FID and the
PyAV video tooling.
"""
from __future__ import annotations

import copy
import os

import torch
import torch.distributed as dist

from . import cnn, config, loss
from .ema import Ema
from .optim import Adam, DataParallel


class LazyLoss:
    """What train_generator / train_discriminator return: a handle on the loss scalar whose value is fetched from the device ON
    DEMAND.  The reference calls `.item()` right after the optimiser step (worker.py:177, 214), which drains the GPU queue twice
    per iteration; here the scalar is copied asynchronously into pinned memory and `float()` / `.item()` / formatting / arithmetic /
    comparisons / `round()` / `int()` / `bool()` wait for that copy only when somebody looks (log lines every print_interval).
    Deliberately NOT a float subclass (a subclass would have to carry a placeholder value that C-level consumers read without calling
    back): consumers that convert through `__float__` (math.isnan, numpy, `"%f" %`, f-strings) get the real value; `json.dumps` only
    serialises real floats, so pass `float(loss)` there."""
    __slots__ = ("_host", "_event")

    def __init__(self, tensor):
        if tensor.is_cuda:
            self._host = torch.empty((), dtype=torch.float32, pin_memory=True)
            self._host.copy_(tensor.detach(), non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host, self._event = tensor.detach().float().clone(), None

    def __float__(self):
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        return float(self._host)

    item = __float__

    def __int__(self):
        return int(float(self))

    def __bool__(self):
        return bool(float(self))

    def __round__(self, ndigits=None):
        return round(float(self), ndigits)

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))

    def __hash__(self):
        return hash(float(self))

    def __array__(self, dtype=None, copy=None):
        import numpy as np
        return np.asarray(float(self), dtype=dtype)


def _delegate(name):
    import numbers
    op = getattr(float, name)

    def f(self, *o):
        if not all(isinstance(v, (numbers.Real, LazyLoss)) for v in o):
            return NotImplemented                   # e.g. `loss == None`, `loss < "x"`: let Python apply its own fallback rules
        return op(float(self), *(float(v) for v in o))
    return f


for _n in ("eq", "ne", "lt", "le", "gt", "ge", "add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "neg", "abs"):
    setattr(LazyLoss, f"__{_n}__", _delegate(f"__{_n}__"))


class SyntheticTriples:
    """Endless (image, geometry_change, appearance_change) batches, uniform in [-1, 1], fp32, resident on the device."""

    def __init__(self, batch, res, device, seed=1234, pool=4):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.items = [tuple((torch.rand(batch, 3, res, res, generator=g) * 2 - 1).to(device) for _ in range(3)) for _ in range(pool)]
        self.i = 0

    def next(self):
        item = self.items[self.i % len(self.items)]
        self.i += 1
        return item


class SyntheticHostTriples:
    """The same batches held in PINNED HOST memory and copied to the device per iteration, as the reference's three `.to(device)` of a
    DataLoader batch do (worker.py:141-143): the copy of batch i + 1 runs on a side stream while iteration i computes (two device
    buffers in turn), `next()` makes the compute stream wait for the copy it hands out.  `--dataset_path synthetic-host`; the
    PCIe-inclusive bench line (`bench.py --h2d`)."""

    def __init__(self, batch, res, device, seed=1234, pool=4):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.device = torch.device(device)
        cuda = self.device.type == "cuda"
        self.hosts = [tuple((torch.rand(batch, 3, res, res, generator=g) * 2 - 1) for _ in range(3)) for _ in range(pool)]
        if cuda:
            self.hosts = [tuple(t.pin_memory() for t in item) for item in self.hosts]
        self.copy_stream = torch.cuda.Stream(device=self.device) if cuda else None
        self.i = 0
        self._ahead = None
        self._submit()

    def _submit(self):
        item = self.hosts[self.i % len(self.hosts)]
        self.i += 1
        if self.copy_stream is None:
            self._ahead = (tuple(t.clone() for t in item), None)
            return
        with torch.cuda.stream(self.copy_stream):
            dev = tuple(t.to(self.device, non_blocking=True) for t in item)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self._ahead = (dev, done)

    def next(self):
        dev, done = self._ahead
        if done is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            for t in dev:
                t.record_stream(cur)                     # allocated on the copy stream, consumed on the compute stream
        self._submit()
        return dev


class WORKER(object):
    def __init__(self, args, local_rank, gpus_per_node, device=None):
        self.args = args
        self.local_rank = local_rank
        self.gpus_per_node = gpus_per_node
        self.local_batch_size = args.batch_size // gpus_per_node                      # worker.py:35
        self.global_iter_counter = 0
        self.device = torch.device(device) if device is not None else torch.device("cuda", local_rank)
        if self.device.type == "cuda" and not torch.cuda.is_available():
            raise RuntimeError("lcgan_amd needs a HIP device: there is no CPU fallback for the training step")
        self.group = dist.new_group(list(range(gpus_per_node))) if (dist.is_available() and dist.is_initialized()) else None
        self.data = self.prepare_training_dataset()
        self.generator, self.discriminator, self.g_optimizer, self.d_optimizer = self.set_cnn_models()
        self.generator_ema = copy.deepcopy(self.generator)                            # worker.py:40 (keeps the `module.` prefix)
        self.ema = Ema(self.generator, self.generator_ema, self.args.g_ema_decay, self.args.g_ema_start)
        self.best_fid = 9999
        # Deferred optimiser work (N > 1 only): after a backward the gradient bucket is all-reduced asynchronously (RCCL runs
        # on its own stream) and the Adam step that needs it is postponed until the OTHER network's parameter-independent
        # forward has been issued, so the reduction over xGMI overlaps compute (BASELINE.json north_star):
        #   G all-reduce  ||  D(real) forward of the D step        D all-reduce  ||  G forward(s) of the next G step
        self._pending = {"g": [], "d": []}

    def flush(self, which=("g", "d")):
        """Run postponed all-reduce waits / Adam / EMA for the given network(s) (no-op when nothing is pending)."""
        for k in which:
            todo, self._pending[k] = self._pending[k], []
            for fn in todo:
                fn()

    def _after_backward(self, key, model, optimizer):
        handle = model.sync_gradients(async_op=True)
        if model.world_size == 1:
            optimizer.step()
            return

        def finish():
            handle.wait()
            optimizer.step()
        self._pending[key].append(finish)

    # ---- data -----------------------------------------------------------------------------------------------------
    def prepare_training_dataset(self):
        path = str(getattr(self.args, "dataset_path", "synthetic"))
        if path == "synthetic-host":
            return SyntheticHostTriples(self.local_batch_size, self.args.img_resolution, self.device, seed=1234 + self.local_rank)
        if path.startswith("synthetic"):
            return SyntheticTriples(self.local_batch_size, self.args.img_resolution, self.device, seed=1234 + self.local_rank)
        from .data import FolderTriples                                            # custom_dataset.py:10-100, worker.py:44-73
        return FolderTriples(path, self.args.img_resolution, self.local_batch_size, self.device, rank=self.local_rank,
                             world=self.gpus_per_node, train=getattr(self.args, "phase", "train") == "train")

    def sample_data_basket(self):
        return self.data.next()

    # ---- models / optimisers (worker.py:75-112) ---------------------------------------------------------------------
    def set_cnn_models(self):
        generator = cnn.Generator(self.args).to(self.device)
        discriminator = cnn.Discriminator(self.args).to(self.device)
        g_parameters = [p for _, p in generator.named_parameters()]
        d_parameters = [p for _, p in discriminator.named_parameters()]
        generator = DataParallel(generator, self.group)
        discriminator = DataParallel(discriminator, self.group)
        betas = (self.args.beta1, self.args.beta2)
        g_optimizer = Adam(g_parameters, lr=self.args.g_lr, betas=betas, eps=1e-8, on_zero_grad=generator.reset_reduction)
        d_optimizer = Adam(d_parameters, lr=self.args.d_lr, betas=betas, eps=1e-8, on_zero_grad=discriminator.reset_reduction)
        return generator, discriminator, g_optimizer, d_optimizer

    def freeze_discriminator(self, freeze_up_to_index=5):
        """worker.py:127-131: freezes conv1x1, LeakyReLU and the first `freeze_up_to_index` blocks of D.shared_model."""
        for i, (_, layer) in enumerate(self.discriminator.module.shared_model.named_children()):
            if i < freeze_up_to_index + 2:
                for param in layer.parameters():
                    param.requires_grad = False

    def requires_grad(self, model, flag=True):
        for p in model.parameters():
            p.requires_grad = flag

    def _randn(self, dim):
        return torch.randn(self.local_batch_size, dim, device=self.device)

    # ---- D step (worker.py:137-177) -----------------------------------------------------------------------------------
    def train_discriminator(self, epoch):
        self.flush(("d",))
        self.d_optimizer.zero_grad()
        image, geometry_change, appearance_change = self.sample_data_basket()
        rand1 = self._randn(self.args.geo_noise_dim)
        rand2 = self._randn(self.args.app_noise_dim)
        odd = epoch % 2 == 1

        if self._can_batch(2 if odd else 4) and (epoch % 8 != 1 or self._r1_batched()):
            return self._train_discriminator_batched(epoch, image, geometry_change, appearance_change, rand1, rand2)
        # (1) everything that only needs D's parameters: the real-image passes.  With N > 1 these overlap G's gradient all-reduce.
        if odd:
            image = image.detach().clone().requires_grad_(True)                       # worker.py:152
            real_logit, _, _ = self.discriminator(image, False)
        else:
            real_logit, geometry_feat, appearance_feat = self.discriminator(image, True)
            _, geometry_positive, appearance_negative = self.discriminator(geometry_change, True)
            _, geometry_negative, appearance_positive = self.discriminator(appearance_change, True)

        # (2) G's postponed Adam + EMA (needs the reduced gradients), then the fake batch with the UPDATED generator (worker.py:145-149)
        self.flush(("g",))
        with torch.no_grad():                      # G is frozen in this phase (loader.py:50); same numbers, no graph
            fake_img = self.generator(rand1, rand2)
        fake_logit, _, _ = self.discriminator(fake_img, False)

        if odd:
            d_loss = loss.bce_with_logits(real_logit, True) + loss.bce_with_logits(fake_logit, False)
            if epoch % 8 == 1:
                d_loss = d_loss + loss.cal_r1_reg(real_logit, image, self.device) * self.args.l_r1
        else:
            d_adv_loss = loss.bce_with_logits(real_logit, True) + loss.bce_with_logits(fake_logit, False)
            d_aug_loss = (loss.contrastive_loss(geometry_feat, geometry_positive, geometry_negative, self.args.tau)
                          + loss.contrastive_loss(appearance_feat, appearance_positive, appearance_negative, self.args.tau)) * self.args.l_aux
            d_loss = d_adv_loss + d_aug_loss

        d_loss.backward()
        self._after_backward("d", self.discriminator, self.d_optimizer)
        return LazyLoss(d_loss)

    def _can_batch(self, n_sub: int) -> bool:
        """n_sub calls as one batch only while the largest feature map of the merged pass (n_sub x local batch x R^2 x the full-resolution
        channel count, cnn.py:17,54) stays addressable by the fast kernels' 32-bit element offsets (and the warp kernels' 4 GB): at
        1024 x 1024 / batch 32 on ONE GPU three generator calls would be 3.2e9 elements -- those configurations keep the reference's
        separate calls"""
        res = self.args.img_resolution
        return config.batched_passes() and n_sub * self.local_batch_size * res * res * cnn._base_nf(res) < (1 << 31) - (1 << 24)

    def _r1_batched(self) -> bool:
        """R1 iterations too evaluate [real | fake] as one discriminator batch when the local batch is small enough that the step is bound
        by launch count, not by arithmetic: the penalty's first-order pass and its double backward then also run over the fake half (zero
        cotangents: +33 % discriminator FLOPs) but in a third fewer launches.  LCGAN_R1_BATCHED_PIXELS: largest local batch x resolution^2
        for which that pays (default: local batch 4 at 256 x 256; measured 19.8 -> 19.0 ms there; 27.4 -> 28.6 ms at local batch 8)."""
        import os
        lim = int(os.environ.get("LCGAN_R1_BATCHED_PIXELS", str(4 * 256 * 256)))
        return self.local_batch_size * self.args.img_resolution ** 2 <= lim

    def _train_discriminator_batched(self, epoch, image, geometry_change, appearance_change, rand1, rand2):
        """The D step of an iteration WITHOUT the R1 penalty with all its discriminator evaluations as ONE batch: [real | fake] on odd
        iterations (worker.py:152-157), [image | geometry view | appearance view | fake] on even ones (worker.py:163-169).  The reference
        calls D once per tensor; the calls share the weights and are independent per sample except for the minibatch-stddev statistic,
        which `n_sub` keeps per call (custom_layers.py:243-256) -- so this is the same function evaluated in fewer, fatter launches (the
        low-resolution and latency-class kernels run once instead of 2-4 times; at local batch 4 that is most of the step).  The R1
        iteration keeps separate passes: its double backward belongs to the real batch alone (loss.py:18-34)."""
        B = image.shape[0]
        self.flush(("g",))                         # the fake batch needs the UPDATED generator (worker.py:145-149)
        with torch.no_grad():
            fake_img = self.generator(rand1, rand2)
        if epoch % 2 == 1:
            image = image.detach().clone().requires_grad_(True)                       # worker.py:152 (every odd iteration)
            logit, _, _ = self.discriminator(torch.cat([image, fake_img], dim=0), False, n_sub=2)
            real_logit, fake_logit = logit[:B], logit[B:]
            d_loss = loss.bce_with_logits(real_logit, True) + loss.bce_with_logits(fake_logit, False)
            if epoch % 8 == 1:                                                       # (only with _r1_batched(): worker.py:159-161)
                d_loss = d_loss + loss.cal_r1_reg(real_logit, image, self.device) * self.args.l_r1
        else:
            logit, gf, af = self.discriminator(torch.cat([image, geometry_change, appearance_change, fake_img], dim=0), True, n_sub=4)
            real_logit, fake_logit = logit[:B], logit[3 * B:]
            geometry_feat, geometry_positive, geometry_negative = gf[:B], gf[B:2 * B], gf[2 * B:3 * B]
            appearance_feat, appearance_negative, appearance_positive = af[:B], af[B:2 * B], af[2 * B:3 * B]
            d_adv_loss = loss.bce_with_logits(real_logit, True) + loss.bce_with_logits(fake_logit, False)
            d_aug_loss = (loss.contrastive_loss(geometry_feat, geometry_positive, geometry_negative, self.args.tau)
                          + loss.contrastive_loss(appearance_feat, appearance_positive, appearance_negative, self.args.tau)) * self.args.l_aux
            d_loss = d_adv_loss + d_aug_loss
        d_loss.backward()
        self._after_backward("d", self.discriminator, self.d_optimizer)
        return LazyLoss(d_loss)

    # ---- G step (worker.py:179-214) -----------------------------------------------------------------------------------
    def train_generator(self, epoch):
        self.flush(("g",))
        self.g_optimizer.zero_grad()
        rand1 = self._randn(self.args.geo_noise_dim)
        rand2 = self._randn(self.args.app_noise_dim)
        resample1 = self._randn(self.args.geo_noise_dim)
        resample2 = self._randn(self.args.app_noise_dim)

        # G forwards need only G's parameters: with N > 1 they overlap the all-reduce of the previous D step's gradients
        batched = epoch % 2 == 0 and self._can_batch(3)
        if epoch % 2 == 1:
            images = (self.generator(rand1, rand2),)
        elif batched:
            # the three generator calls of an even iteration (worker.py:194-196) as one batch of 3 B, and below the three discriminator
            # calls on their results (worker.py:198-200) as one too; avg-latent updates and minibatch-stddev stay per call (n_sub)
            B = rand1.shape[0]
            images = self.generator(torch.cat([rand1, resample1, rand1], dim=0), torch.cat([rand2, rand2, resample2], dim=0), n_sub=3)
        else:
            images = (self.generator(rand1, rand2), self.generator(resample1, rand2), self.generator(rand1, resample2))   # worker.py:194-196
        self.flush(("d",))                         # D's postponed Adam must land before D is evaluated

        if epoch % 2 == 1:
            logit, _, _ = self.discriminator(images[0], False)
            g_loss = loss.bce_with_logits(logit, True)
        else:
            if batched:
                lg, gf, af = self.discriminator(images, True, n_sub=3)
                logit, geometry_feat, appearance_feat = lg[:B], gf[:B], af[:B]
                geometry_positive, appearance_negative = gf[B:2 * B], af[B:2 * B]
                geometry_negative, appearance_positive = gf[2 * B:], af[2 * B:]
            else:
                logit, geometry_feat, appearance_feat = self.discriminator(images[0], True)
                _, geometry_positive, appearance_negative = self.discriminator(images[1], True)
                _, geometry_negative, appearance_positive = self.discriminator(images[2], True)
            g_adv_loss = loss.bce_with_logits(logit, True)
            g_aug_loss = (loss.contrastive_loss(geometry_feat, geometry_positive, geometry_negative, self.args.tau)
                          + loss.contrastive_loss(appearance_feat, appearance_positive, appearance_negative, self.args.tau)) * self.args.l_aux
            g_sparsity_loss = loss.l1_sparsity([self.generator.module.geometry_mapping.diagonal_params,
                                                self.generator.module.appearance_mapping.diagonal_params], self.args.l_s)
            g_loss = g_adv_loss + g_aug_loss + g_sparsity_loss

        g_loss.backward()
        self._after_backward("g", self.generator, self.g_optimizer)
        return LazyLoss(g_loss)

    def ema_update(self, current_step):
        if self._pending["g"]:                      # the EMA reads the UPDATED generator: keep it behind the postponed Adam
            self._pending["g"].append(lambda: self.ema.update(current_step))
        else:
            self.ema.update(current_step)

    # ---- checkpoints (worker.py:219-253): `module.`-prefixed state_dicts, same file names ------------------------------
    def _paths(self, best=False):
        d = os.path.join(self.args.model_name, self.args.save_dir)
        sfx = "_best" if best else ""
        return (f"{d}/gen_model{sfx}.ckpt", f"{d}/gen_ema_model{sfx}.ckpt", f"{d}/disc_model{sfx}.ckpt", f"{d}/optim_state{sfx}.ckpt")

    def save_model(self, best=False):
        self.flush()
        g, e, d, o = self._paths(best)
        torch.save(self.generator.state_dict(), g)
        torch.save(self.generator_ema.state_dict(), e)
        torch.save(self.discriminator.state_dict(), d)
        # not in the reference (its Adam moments restart from zero on resume, worker.py:239-253): an extra file its tools ignore
        torch.save({"g": self.g_optimizer.state_dict(), "d": self.d_optimizer.state_dict()}, o)

    def save_best_model(self):
        self.save_model(best=True)

    def load_model(self):
        self.flush()
        g, e, d, o = self._paths(bool(getattr(self.args, "best", False)))
        self.generator.load_state_dict(torch.load(g, map_location=self.device))
        self.generator_ema.load_state_dict(torch.load(e, map_location=self.device))
        self.discriminator.load_state_dict(torch.load(d, map_location=self.device))
        from . import ops
        ops.bump_weight_epoch()                    # load_state_dict copies in place: every prepared weight is stale
        if os.path.exists(o):                      # a checkpoint written by the reference has none: moments start at zero, as there
            st = torch.load(o, map_location=self.device)
            self.g_optimizer.load_state_dict(st["g"])
            self.d_optimizer.load_state_dict(st["d"])

    # ---- inference (worker.py:427-441) ---------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, geometry_code, appearance_code, w_psi=None):
        """generator_ema(geo, app, w_psi) under no_grad (worker.py:277-281, 406, 433, 462): forward only -- no autograd graph, so
        no activation is kept beyond its consumer."""
        return self.generator_ema(geometry_code, appearance_code, float(self.args.w_psi if w_psi is None else w_psi))

    def fid_evaluate(self, feature_extractor, num_batches=None):
        """worker.py:381-425 with the feature network passed in (lcgan_amd/fid.py): features of `num_batches` real batches vs as many
        batches of generator_ema(randn, randn, w_psi); returns the FID and tracks best_fid in the train phase."""
        import numpy as np
        from . import fid
        if num_batches is None:
            n = min(len(getattr(self.data, "files", ())) or 50000, 50000)
            num_batches = max(n // self.local_batch_size, 1)
        real, fake = [], []
        with torch.no_grad():
            for _ in range(num_batches):
                image = self.sample_data_basket()[0]
                real.append(feature_extractor(image).reshape(image.shape[0], -1).float().cpu())
            for _ in range(num_batches):
                img = self.generate(self._randn(self.args.geo_noise_dim), self._randn(self.args.app_noise_dim))
                fake.append(feature_extractor(img).reshape(img.shape[0], -1).float().cpu())
        sm, sc = fid.feature_statistics(torch.cat(fake, 0).numpy())
        rm, rc = fid.feature_statistics(torch.cat(real, 0).numpy())
        value = fid.calc_fid(sm, sc, rm, rc)
        if value < self.best_fid and getattr(self.args, "phase", "train") == "train":
            self.best_fid = value
        return value

    def fake_image_generation(self, num_images=50):
        """worker.py:427-441: `num_images` files, each the local batch stacked in one column (save_image(nrow=1, padding=0))."""
        from .data import save_image_column
        folder = os.path.join(self.args.model_name, "fakes")
        os.makedirs(folder, exist_ok=True)
        for count in range(num_images):
            fake = self.generate(self._randn(self.args.geo_noise_dim), self._randn(self.args.app_noise_dim))
            fake = ((fake + 1) / 2).clamp(0.0, 1.0)
            save_image_column(fake, os.path.join(folder, "{num:04d}_images.jpg".format(num=count)))
