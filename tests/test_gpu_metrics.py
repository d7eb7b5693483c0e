"""On-device quality metrics against the reference-generated PSNR goldens and the numpy oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_psnr_mse_match_reference_goldens(gpu_device, golden_dir):
    from elvis_amd import metrics
    g = np.load(os.path.join(golden_dir, "psnr.npz"))
    a, b, m = g["a"], g["b"], g["mask"]
    # (the reference averages squared float32 differences in float32; the device sums them exactly in integers:
    #  agreement to float32 rounding of the mean, 1e-6 relative)
    for i in range(5):
        assert metrics.masked_psnr(a[i], b[i], device=gpu_device) == pytest.approx(g["psnr_full"][i], abs=1e-5)
        assert metrics.masked_psnr(a[i], b[i], m[i], gpu_device) == pytest.approx(g["psnr_masked"][i], abs=1e-5)
        assert metrics.masked_mse(a[i], b[i], device=gpu_device) == pytest.approx(g["mse_full"][i], rel=1e-6)
        assert metrics.masked_mse(a[i], b[i], m[i], gpu_device) == pytest.approx(g["mse_masked"][i], rel=1e-6)
    from oracle import glue_ref
    ps = metrics.calculate_psnr(list(a), list(b), device=gpu_device)
    assert ps == pytest.approx([glue_ref.psnr_whole(x, y) for x, y in zip(a, b)], abs=1e-4)
    assert metrics.calculate_psnr([a[0]], [a[0]], device=gpu_device) == [float("inf")]
    assert metrics.masked_psnr(a[0], b[0], np.zeros_like(m[0]), gpu_device) == 100.0
    assert metrics.calculate_mse([], [], gpu_device) == []


@pytest.mark.parametrize("b", [8, 16, 12])
def test_block_ssim(gpu_device, b):
    from elvis_amd import metrics
    from oracle import glue_ref
    rng = np.random.default_rng(b)
    f1 = [rng.integers(0, 256, size=(50, 70, 3), dtype=np.uint8) for _ in range(2)]
    f2 = [np.clip(f.astype(int) + rng.integers(-12, 13, f.shape), 0, 255).astype(np.uint8) for f in f1]
    f2[1][:b, :b] = f1[1][:b, :b]                       # an identical block -> SSIM 1
    got = metrics.calculate_block_ssim(f1, f2, b, gpu_device)
    assert len(got) == 2 and got[0].shape == (50 // b, 70 // b) and got[0].dtype == np.float32
    for x, y, s in zip(f1, f2, got):
        assert np.abs(s - glue_ref.block_ssim(x, y, b)).max() < 2e-5
    assert got[1][0, 0] == pytest.approx(1.0, abs=1e-6)
    assert (got[0] < 0.999).all() and (got[0] > 0.0).all()
