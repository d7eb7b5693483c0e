"""Server-side degrade filters on the device against the numpy oracle: bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.random((h // 4 + 1, w // 4 + 1, 3))
    up = np.kron(base, np.ones((4, 4, 1)))[:h, :w]
    return np.round(np.clip(up + rng.normal(0, 0.08, up.shape), 0, 1) * 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,b", [(64, 96, 8), (48, 64, 16), (32, 40, 4), (16, 16, 2)])
def test_filter_frame_downsample(gpu_device, h, w, b):
    from elvis_amd import degrade
    from oracle import degrade_ref as R
    img = _img(h, w, 1)
    scores = np.random.default_rng(2).random((h // b, w // b))
    scores.flat[0], scores.flat[1] = 1.0, 0.0          # the coarsest level (one sample per block) and an untouched block
    got, maps = degrade.filter_frame_downsample(img, scores, b, gpu_device)
    ref, ref_maps = R.filter_frame_downsample(img, scores, b)
    assert maps.dtype == np.int32 and np.array_equal(maps, ref_maps) and maps.max() == int(np.log2(b))
    assert np.array_equal(got, ref)
    keep = np.repeat(np.repeat(maps == 0, b, 0), b, 1)
    assert np.array_equal(got[keep], img[keep]) and not np.array_equal(got, img)
    with pytest.raises(ValueError):
        degrade.filter_frame_downsample(img[:h - 1], scores, b, gpu_device)


@pytest.mark.parametrize("h,w,b", [(64, 96, 8), (32, 48, 16), (24, 24, 4)])
def test_filter_frame_gaussian(gpu_device, h, w, b):
    from elvis_amd import degrade
    from oracle import degrade_ref as R
    img = _img(h, w, 3)
    scores = np.random.default_rng(4).random((h // b, w // b))
    scores.flat[0], scores.flat[1] = 0.0, 1.0          # rounds 0 and 10
    got, rounds = degrade.filter_frame_gaussian(img, scores, b, gpu_device)
    ref, ref_rounds = R.filter_frame_gaussian(img, scores, b)
    assert np.array_equal(rounds, ref_rounds) and rounds.max() == 10
    assert np.array_equal(got, ref)
    assert np.array_equal(got[:b, :b], img[:b, :b])     # rounds == 0: untouched
    # nothing leaks between blocks: changing one block's pixels changes only that block
    img2 = img.copy()
    img2[b:2 * b, b:2 * b] = 255 - img2[b:2 * b, b:2 * b]
    got2, _ = degrade.filter_frame_gaussian(img2, scores, b, gpu_device)
    diff = np.any(got2 != got, axis=2)
    diff[b:2 * b, b:2 * b] = False
    assert not diff.any()


def test_filter_frame_dct(gpu_device):
    from elvis_amd import degrade
    from oracle import degrade_ref as R
    img = _img(64, 96, 5)
    scores = np.random.default_rng(6).random((8, 12))
    got, levels = degrade.filter_frame_dct(img, scores, 8, gpu_device)
    assert np.array_equal(got, R.dct_dampen(img, levels, degrade.DCT_LEVELS))
    keep = np.repeat(np.repeat(levels == 0, 8, 0), 8, 1)
    assert np.array_equal(got[keep], img[keep])
    # the DC coefficient is untouched: block means move by rounding only
    m0 = img.reshape(8, 8, 12, 8, 3).astype(np.float64).mean(axis=(1, 3))
    m1 = got.reshape(8, 8, 12, 8, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(m0 - m1).max() < 0.6
    # and the high frequencies shrink with the level
    hf = lambda a: np.abs(np.diff(a.astype(np.float64), axis=1)).reshape(8, 8, -1).mean()
    assert hf(got) < hf(img)


def test_device_forms_batch_and_1080p(gpu_device):
    from elvis_amd import degrade, synth
    from oracle import degrade_ref as R
    clip = synth.synth_clip(7, 2, 1080, 1920)
    lv = synth.synth_level_maps(8, 2, 135, 240).astype(np.int32)
    fd, ld = torch.from_numpy(clip).to(gpu_device), torch.from_numpy(lv).to(gpu_device)
    down = degrade.degrade_downsample_device(fd, ld, 8).cpu().numpy()
    blur = degrade.degrade_gaussian_device(fd, ld, 8).cpu().numpy()
    dct = degrade.degrade_dct_device(fd, ld).cpu().numpy()
    # one 1080p frame against the oracle on a 128 x 256 crop (block-local filters: crops commute)
    ys, xs = slice(512, 640), slice(1024, 1280)
    mys, mxs = slice(64, 80), slice(128, 160)
    assert np.array_equal(down[1][ys, xs], R.degrade_downsample(clip[1][ys, xs], lv[1][mys, mxs], 8))
    assert np.array_equal(blur[1][ys, xs], R.degrade_gaussian(clip[1][ys, xs], lv[1][mys, mxs], 8))
    assert np.array_equal(dct[1][ys, xs], R.dct_dampen(clip[1][ys, xs], lv[1][mys, mxs], degrade.DCT_LEVELS))
    with pytest.raises(ValueError):
        degrade.degrade_downsample_device(fd, ld[:1], 8)
