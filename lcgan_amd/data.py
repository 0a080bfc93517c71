"""Training data for the step: the reference's Dataset_ + DistributedSampler + DataLoader (custom_dataset.py:10-100, worker.py:44-73)
re-cut for the MI355X: the host only decodes and resizes (PIL, LANCZOS -- custom_dataset.py:16, 64-66) and draws the per-sample
randomness; the three views (h-flip :68, perspective :27-33, dropout / colour jitter :35-49, normalisation :81-86) are produced on the
device by ONE call (`lcgan_make_views`, csrc/views.hip) from a pinned, asynchronously copied batch; decode runs in a thread pool one
batch ahead of the training thread (`FolderTriples`).  Measured decode throughput: scripts/bench_data.py (recorded in DESIGN.md).

The random draws restate albumentations 1.x (`A.Perspective(scale=(0.05, 0.1), keep_size=True, fit_output=True|False)`,
`A.CoarseDropout(max_holes=1, 0.3..0.5)`, `A.ColorJitter(0.2, 0.2, 0.2, 0.2)`: custom_dataset.py:19-24) from its published algorithm;
albumentations / OpenCV are absent from this image and from /root/reference, so this row is "parity unpinned" against the
dependency: tests pin the kernel against the CPU restatement of the same pixel math (oracle/hip_emulation.py:make_views) and the
sampler against the geometric properties the algorithm guarantees (tests/test_data.py).
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")     # torchvision.datasets.folder


# ---- randomness of one sample -> 32 floats (layout: csrc/views.hip) ---------------------------------------------------------
def _perspective_transform(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """3x3 homography with H @ (src_i, 1) ~ (dst_i, 1) for 4 point pairs (cv2.getPerspectiveTransform)."""
    A, rhs = [], []
    for (x, y), (u, v) in zip(src, dst):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); rhs.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); rhs.append(v)
    h = np.linalg.solve(np.asarray(A, dtype=np.float64), np.asarray(rhs, dtype=np.float64))
    return np.append(h, 1.0).reshape(3, 3)


def _order_points(pts: np.ndarray) -> np.ndarray:
    """top-left, top-right, bottom-right, bottom-left (albumentations Perspective._order_points)."""
    xs = pts[np.argsort(pts[:, 0])]
    left, right = xs[:2], xs[2:]
    tl, bl = left[np.argsort(left[:, 1])]
    d = np.linalg.norm(right - tl, axis=1)
    br, tr = right[np.argsort(d)[::-1]]
    return np.array([tl, tr, br, bl], dtype=np.float64)


def perspective_inverse(rng: np.random.Generator, size: int, fit_output: bool, scale=(0.05, 0.1)) -> np.ndarray:
    """Inverse map (output pixel -> source pixel) of A.Perspective(scale, keep_size=True, fit_output): the four image corners are
    jittered inwards by |N(0, s)| mod 0.32 of the image size, s ~ U(scale); the quadrilateral is mapped onto a rectangle of its own
    mean extent (fit_output=False: the quadrilateral fills the frame) or the whole warped image is fitted into the frame with black
    borders (fit_output=True); keep_size resizes the result back to size x size."""
    s = rng.uniform(*scale)
    pts = np.mod(np.abs(rng.normal(0.0, s, size=(4, 2))), 0.32)
    pts[1, 0] = 1.0 - pts[1, 0]                    # top right
    pts[2] = 1.0 - pts[2]                          # bottom right
    pts[3, 1] = 1.0 - pts[3, 1]                    # bottom left
    pts = _order_points(pts * size)
    tl, tr, br, bl = pts
    max_w = max(int(np.linalg.norm(br - bl)), int(np.linalg.norm(tr - tl)), 2)
    max_h = max(int(np.linalg.norm(tr - br)), int(np.linalg.norm(tl - bl)), 2)
    dst = np.array([[0, 0], [max_w, 0], [max_w, max_h], [0, max_h]], dtype=np.float64)
    M = _perspective_transform(pts, dst)
    if fit_output:                                 # expand the canvas to the warped image's bounding box
        corners = np.array([[0, 0, 1], [size, 0, 1], [size, size, 1], [0, size, 1]], dtype=np.float64).T
        w = M @ corners
        w = (w[:2] / w[2]).T
        mn, mx = w.min(axis=0), w.max(axis=0)
        T = np.array([[1, 0, -mn[0]], [0, 1, -mn[1]], [0, 0, 1]], dtype=np.float64)
        M = T @ M
        max_w, max_h = max(mx[0] - mn[0], 2.0), max(mx[1] - mn[1], 2.0)
    S = np.diag([max_w / size, max_h / size, 1.0])     # keep_size: output pixel -> warped-canvas pixel
    Hinv = np.linalg.inv(M) @ S
    return Hinv / Hinv[2, 2]


def sample_view_params(rng: np.random.Generator, size: int, mean_luma: float = -1.0, quantize: bool = True) -> np.ndarray:
    """One row of per-sample randomness (layout: csrc/views.hip).  mean_luma < 0 (default): the contrast pivot is computed on the
    device from the image as it reaches the contrast op (after the jitter ops drawn in front of it), which is what ColorJitter's
    contrast adjustment uses; a value in [0, 1] forces the pivot (tests).  quantize: round the two augmented views to the uint8
    grid, as the reference's `Image.fromarray(...)` -> `ToTensor()` round trip does (custom_dataset.py:76-79)."""
    p = np.zeros(32, dtype=np.float32)
    p[0] = float(rng.random() < 0.5)                                                        # RandomHorizontalFlip, custom_dataset.py:68
    p[1:10] = perspective_inverse(rng, size, fit_output=bool(rng.random() < 0.5)).reshape(-1)   # :27-33
    if rng.random() < 0.5:                                                                  # :35-41 CoarseDropout, one hole
        hh, hw = int(size * rng.uniform(0.3, 0.5)), int(size * rng.uniform(0.3, 0.5))
        y0, x0 = int(rng.integers(0, size - hh + 1)), int(rng.integers(0, size - hw + 1))
        p[10], p[11:15] = 0.0, (x0, y0, x0 + hw, y0 + hh)
    else:                                                                                   # ColorJitter(0.2, 0.2, 0.2, 0.2)
        p[10] = 1.0
        p[15], p[16], p[17] = rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2)
        p[18] = rng.uniform(-0.2, 0.2)
        p[19:23] = rng.permutation(4)
        p[23] = mean_luma
    p[25] = float(quantize)
    return p


# ---- folder source -----------------------------------------------------------------------------------------------------------
def list_image_folder(root: str) -> List[str]:
    """torchvision.datasets.ImageFolder(root) file order: classes sorted, files sorted within (custom_dataset.py:51-54)."""
    files = []
    for cls in sorted(e.name for e in os.scandir(root) if e.is_dir()):
        for dirpath, _, names in sorted(os.walk(os.path.join(root, cls), followlinks=True)):
            files += [os.path.join(dirpath, n) for n in sorted(names) if n.lower().endswith(IMG_EXTENSIONS)]
    if not files:
        raise FileNotFoundError(f"no images under {root}/<class>/ (ImageFolder layout, custom_dataset.py:51-54)")
    return files


class FolderTriples:
    """Endless (image, geometry_change, appearance_change) batches from `<data_dir>/train/<class>/*`: per-rank shards of an
    epoch permutation (DistributedSampler(shuffle=True, drop_last=True), worker.py:56-60); decode + LANCZOS resize + the per-sample
    random draws run in a small thread pool ONE BATCH AHEAD of the training thread (the reference: 4 persistent DataLoader workers
    per GPU, worker.py:37,62-69; PIL and numpy release the GIL), into one of two pinned buffers; the training thread only waits for
    the finished batch, issues the H2D copy on a side stream and launches the view kernel.
    Randomness is per sample: `default_rng([seed, rank, batch counter, slot])`, so the result does not depend on thread timing."""

    def __init__(self, data_dir: str, res: int, batch: int, device, rank: int = 0, world: int = 1, train: bool = True, seed: int = 0,
                 workers: int = 4):
        from concurrent.futures import ThreadPoolExecutor
        from . import kernels as KM
        self.K = KM.K
        self.files = list_image_folder(os.path.join(data_dir, "train"))
        self.res, self.batch, self.device, self.rank, self.world, self.train = res, batch, torch.device(device), rank, world, train
        self.epoch, self.pos, self.order = 0, 0, None
        self.seed, self.counter = seed, 0
        cuda = self.device.type == "cuda"
        self.hosts = [torch.empty((batch, 3, res, res), dtype=torch.float32) for _ in range(2)]
        self.phosts = [torch.zeros((batch, 32), dtype=torch.float32) for _ in range(2)]
        if cuda:
            self.hosts = [t.pin_memory() for t in self.hosts]
            self.phosts = [t.pin_memory() for t in self.phosts]
        self.copy_stream = torch.cuda.Stream(device=self.device) if cuda else None
        self._copied = [None, None]               # per buffer: event "the H2D copy out of this buffer has finished"
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="lcgan-data")
        self._inflight = None                     # (buffer index, futures) of the batch being decoded

    def _next_indices(self):
        per_rank = len(self.files) // self.world                      # drop_last
        if self.order is None or self.pos + self.batch > per_rank:
            g = np.random.default_rng(self.seed + self.epoch)         # the same permutation on every rank (sampler.set_epoch)
            perm = g.permutation(len(self.files))[:per_rank * self.world]
            self.order, self.pos = perm[self.rank::self.world], 0
            self.epoch += 1
            if per_rank < self.batch:
                raise ValueError(f"{len(self.files)} images cannot fill a batch of {self.batch} on {self.world} rank(s)")
        idx = self.order[self.pos:self.pos + self.batch]
        self.pos += self.batch
        return idx

    def _decode(self, path: str) -> np.ndarray:
        from PIL import Image
        with Image.open(path) as im:
            im = im.convert("RGB")
            if im.size[0] != self.res:                                # custom_dataset.py:62-66
                im = im.resize((self.res, self.res), Image.LANCZOS)
            return np.array(im, dtype=np.uint8)

    def _load_one(self, buf: int, slot: int, path: str, counter: int) -> None:
        a = self._decode(path)
        self.hosts[buf][slot] = torch.from_numpy(a).permute(2, 0, 1).float().mul_(2.0 / 255.0).sub_(1.0)    # ToTensor, *2-1 (:70, :81)
        rng = np.random.default_rng([self.seed, self.rank, counter, slot])
        p = sample_view_params(rng, self.res)
        if not self.train:
            p[0] = 0.0
        self.phosts[buf][slot] = torch.from_numpy(p)

    def _submit(self) -> None:
        buf = self.counter & 1
        if self._copied[buf] is not None:         # the copy that last read this pinned buffer (two batches ago) must be done
            self._copied[buf].synchronize()
            self._copied[buf] = None
        idx = self._next_indices()
        futs = [self.pool.submit(self._load_one, buf, i, self.files[int(j)], self.counter) for i, j in enumerate(idx)]
        self._inflight = (buf, futs)
        self.counter += 1

    def next(self):
        if self._inflight is None:
            self._submit()
        buf, futs = self._inflight
        for f in futs:
            f.result()                            # (re-raises a worker's exception here)
        host, phost = self.hosts[buf], self.phosts[buf]
        if self.copy_stream is not None:
            with torch.cuda.stream(self.copy_stream):
                src = host.to(self.device, non_blocking=True)
                par = phost.to(self.device, non_blocking=True)
                self._copied[buf] = torch.cuda.Event()
                self._copied[buf].record(self.copy_stream)
            torch.cuda.current_stream().wait_stream(self.copy_stream)
            src.record_stream(torch.cuda.current_stream()); par.record_stream(torch.cuda.current_stream())
        else:
            src, par = host.clone(), phost.clone()
        self._submit()                            # the next batch decodes while this one trains
        return self.K.make_views(src, par)

    def close(self) -> None:
        self.pool.shutdown(wait=False, cancel_futures=True)


def save_image_column(images: torch.Tensor, path: str) -> None:
    """torchvision.utils.save_image(images, path, padding=0, nrow=1) (worker.py:440): the batch stacked in one column, [0,1] -> uint8
    with round-half-up."""
    from PIL import Image
    x = images.detach().float().cpu().clamp(0, 1)
    B, C, H, W = x.shape
    grid = x.permute(0, 2, 3, 1).reshape(B * H, W, C)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).numpy()
    Image.fromarray(arr if C == 3 else arr[..., 0]).save(path)
