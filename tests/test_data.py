"""The data step in front of the hot path: the host-side random draws of lcgan_amd/data.py (geometric properties the
albumentations algorithm guarantees), the folder reader, the image writer, and -- on the GPU -- the view kernel against the CPU
restatement of the same pixel math (oracle/hip_emulation.py:make_views)."""
import os

import numpy as np
import pytest
import torch

from lcgan_amd import data


def _apply(H, pts):
    p = H @ np.concatenate([pts, np.ones((len(pts), 1))], axis=1).T
    return (p[:2] / p[2]).T


def test_perspective_without_fit_maps_the_frame_onto_an_inner_quadrilateral():
    """fit_output=False: the output frame shows the jittered quadrilateral, whose corners lie INSIDE the source image, each within
    0.32 * size of its own image corner (custom_dataset.py:23: A.Perspective(scale=(0.05, 0.1), fit_output=False))."""
    rng = np.random.default_rng(1)
    for _ in range(50):
        size = 256
        Hinv = data.perspective_inverse(rng, size, fit_output=False)
        q = _apply(Hinv, np.array([[0, 0], [size, 0], [size, size], [0, size]], dtype=np.float64))
        corners = np.array([[0, 0], [size, 0], [size, size], [0, size]], dtype=np.float64)
        assert (q >= -1e-6).all() and (q <= size + 1e-6).all()
        assert (np.abs(q - corners) <= 0.32 * size + 1.5).all()          # (+ the int() truncation of the target rectangle)


def test_perspective_with_fit_keeps_the_whole_image_in_frame():
    """fit_output=True: every source corner lands inside the output frame and the warped image touches all four frame edges."""
    rng = np.random.default_rng(2)
    for _ in range(50):
        size = 128
        Hinv = data.perspective_inverse(rng, size, fit_output=True)
        H = np.linalg.inv(Hinv)
        c = _apply(H, np.array([[0, 0], [size, 0], [size, size], [0, size]], dtype=np.float64))
        assert (c >= -1e-6).all() and (c <= size + 1e-6).all()
        assert abs(c[:, 0].min()) < 1e-6 and abs(c[:, 0].max() - size) < 1e-6 and abs(c[:, 1].min()) < 1e-6 and abs(c[:, 1].max() - size) < 1e-6


def test_view_params_ranges():
    rng = np.random.default_rng(3)
    modes = []
    for _ in range(200):
        p = data.sample_view_params(rng, 64, 0.4)
        assert p[0] in (0.0, 1.0) and p[10] in (0.0, 1.0)
        modes.append(p[10])
        if p[10] == 0:                                                    # CoarseDropout(1 hole, 0.3..0.5 of the image)
            w, h = p[13] - p[11], p[14] - p[12]
            assert 0.3 * 64 - 1 <= w <= 0.5 * 64 and 0.3 * 64 - 1 <= h <= 0.5 * 64 and p[11] >= 0 and p[13] <= 64 and p[12] >= 0 and p[14] <= 64
        else:                                                             # ColorJitter(0.2, 0.2, 0.2, 0.2)
            assert all(0.8 <= p[i] <= 1.2 for i in (15, 16, 17)) and -0.2 <= p[18] <= 0.2
            assert sorted(p[19:23]) == [0, 1, 2, 3] and abs(p[23] - 0.4) < 1e-6
    assert 0.3 < np.mean(modes) < 0.7


def _make_folder(root, n=10, size=40):
    from PIL import Image
    rng = np.random.default_rng(0)
    for cls in ("a", "b"):
        os.makedirs(os.path.join(root, "train", cls), exist_ok=True)
        for i in range(n // 2):
            Image.fromarray(rng.integers(0, 256, size=(size, size + 8, 3), dtype=np.uint8)).save(os.path.join(root, "train", cls, f"{i:03d}.png"))


def test_folder_listing_and_rank_shards(tmp_path):
    _make_folder(str(tmp_path))
    files = data.list_image_folder(os.path.join(str(tmp_path), "train"))
    assert len(files) == 10 and files == sorted(files) and os.sep + "a" + os.sep in files[0]
    from oracle.hip_emulation import EmulatedKernels
    from tests.helpers import install_backend
    install_backend(EmulatedKernels())
    try:
        shards = []
        for rank in range(2):
            src = data.FolderTriples(str(tmp_path), 32, 2, "cpu", rank=rank, world=2, seed=5)
            shards.append([int(i) for _ in range(2) for i in src._next_indices()])
            img, geo, app = src.next()
            assert img.shape == geo.shape == app.shape == (2, 3, 32, 32) and float(img.abs().max()) <= 1.0
            assert not torch.equal(img, geo) and not torch.equal(img, app)
        assert not set(shards[0]) & set(shards[1])                        # DistributedSampler: disjoint shards of one permutation
    finally:
        install_backend(None)


def test_save_image_column(tmp_path):
    from PIL import Image
    x = torch.rand(3, 3, 8, 8)
    data.save_image_column(x, str(tmp_path / "c.png"))
    back = np.asarray(Image.open(tmp_path / "c.png"))
    assert back.shape == (24, 8, 3)
    assert np.array_equal(back, (x.permute(0, 2, 3, 1).reshape(24, 8, 3) * 255 + 0.5).clamp(0, 255).to(torch.uint8).numpy())


def test_prefetching_loader_is_deterministic_and_overlapped(tmp_path):
    """The thread-pool loader (one batch ahead, lcgan_amd/data.py) returns the same batches whatever the worker count, walks the
    rank's shard in order, and has the next batch in flight when next() returns."""
    _make_folder(str(tmp_path), n=24)
    from oracle.hip_emulation import EmulatedKernels
    from tests.helpers import install_backend
    install_backend(EmulatedKernels())
    try:
        runs = []
        for workers in (1, 4):
            src = data.FolderTriples(str(tmp_path), 32, 4, "cpu", seed=9, workers=workers)
            batches = [src.next() for _ in range(3)]
            assert src._inflight is not None and len(src._inflight[1]) == 4        # batch 4 is already decoding
            src.close()
            runs.append(batches)
        for a, b in zip(*runs):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
        assert not torch.equal(runs[0][0][0], runs[0][1][0])
        # the augmented views sit on the uint8 grid (custom_dataset.py:76-79), the plain image on the decoder's
        for v in runs[0][0]:
            k = (v + 1) * 0.5 * 255
            assert float((k - k.round()).abs().max()) < 2e-3
    finally:
        install_backend(None)


@pytest.mark.gpu
@pytest.mark.parametrize("device_pivot,quant", [(False, False), (True, False), (True, True)])
def test_make_views_kernel_vs_emulation(device_pivot, quant):
    from lcgan_amd import kernels as KM
    from oracle.hip_emulation import EmulatedKernels
    rng = np.random.default_rng(7)
    for R, B in ((64, 6), (256, 3)):
        src = torch.from_numpy(rng.random((B, 3, R, R), dtype=np.float32) * 2 - 1)
        params = torch.from_numpy(np.stack([data.sample_view_params(rng, R, -1.0 if device_pivot else 0.5, quant) for _ in range(B)]))
        params[0, 10], params[1, 10] = 0.0, 1.0                              # both appearance modes in every batch
        params[1, 15:19] = torch.tensor([1.1, 0.9, 1.15, 0.13]); params[1, 19:23] = torch.tensor([3.0, 1.0, 0.0, 2.0])
        params[1, 23] = -1.0 if device_pivot else 0.5                        # order hue, CONTRAST, brightness, saturation: the pivot follows the hue shift
        params[2, 10] = 1.0
        params[2, 15:19] = torch.tensor([0.85, 1.2, 0.8, -0.17]); params[2, 19:23] = torch.tensor([0.0, 2.0, 3.0, 1.0])
        params[2, 23] = -1.0 if device_pivot else 0.45                       # contrast last: the pivot follows all three other ops
        params[0, 11:15] = torch.tensor([R // 4, R // 8, R // 4 + R // 3, R // 8 + R // 2], dtype=torch.float32)
        hip = KM.K.make_views(src.cuda(), params.clone().cuda())
        emu = EmulatedKernels().make_views(src, params.clone())
        for name, h, e in zip(("image", "geometry", "appearance"), hip, emu):
            h = h.cpu()
            # bilinear taps whose source coordinate sits within rounding of an integer may pick the neighbouring texel pair
            # (the weights then differ by ~1e-5 too), and with the uint8 rounding a value within ~1e-6 of a half step lands on the
            # neighbouring level (2 / 255): compare at 2e-4 on all but a handful of pixels, and in the mean
            diff = (h - e).abs()
            assert float((diff > 2e-4).float().mean()) < (3e-3 if quant else 1e-3), (name, R, float(diff.max()))
            assert float(diff.mean()) < (3e-5 if quant else 1e-5), (name, R)
            if quant and name != "image":
                k = (h + 1) * 0.5 * 255
                assert float((k - k.round()).abs().max()) < 2e-3
