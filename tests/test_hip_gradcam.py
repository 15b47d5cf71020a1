"""GPU parity of the image-level Grad-CAM branch against the reference's own outputs
(tests/golden/gradcam.npz, made by src/self_supervised/gradcam.py) and the oracle restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4          # fp32 bar of the north star, on maps normalised to [0, 1]


@pytest.fixture(scope="module")
def cam(seeded_sd):
    from self_supervised.models import PeraNet
    from self_supervised.gradcam import GradCam
    assert torch.cuda.is_available()
    m = PeraNet(); m.load_state_dict(seeded_sd); m.to("cuda:0")
    return GradCam(m)


def test_gradcam_matches_reference_fixture(cam, golden):
    from oracle import weights as ow
    g = golden("gradcam")
    x = ow.synthetic_images(2, 64, seed=301)
    cases = [("cam64_auto", x[0:1], None), ("cam64_c1", x[1:2], 1), ("cam64_c2", x[1:2], torch.tensor(2)),
             ("cam128_auto", ow.synthetic_images(1, 128, seed=302), None),
             ("cam32_c1", ow.synthetic_images(1, 32, seed=303), 1)]
    for key, xx, c in cases:
        got = cam(xx, c)
        assert got.shape == g[key].shape and got.is_cuda
        np.testing.assert_allclose(got.cpu().numpy(), g[key], atol=TOL, rtol=0, err_msg=key)
        assert got.min().item() == 0.0 and got.max().item() == 1.0


def test_gradcam_batch_equals_per_image_oracle(cam, seeded_sd):
    from oracle import weights as ow
    from oracle.gradcam import gradcam_batch
    from oracle.peranet import OraclePeraNet
    ref = OraclePeraNet(); ref.load_state_dict(seeded_sd)
    x = ow.synthetic_images(5, 256, seed=304)
    cls = torch.tensor([1, 2, 3, 1, 2])
    want = gradcam_batch(ref, x, cls)
    got = cam(x, cls)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=TOL, rtol=0)
    # leaves no parameter gradient behind and the model stays usable for scoring
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in cam.localizer.parameters())
    out = cam.localizer(x.to("cuda:0"))
    assert out["classifier"].shape == (5, 4)


def test_gradcam_maps_image_level_branch(cam, seeded_sd):
    """tools.gradcam_maps == the evaluator.py:268-282 loop over the oracle (zero map for images predicted good)."""
    from oracle import weights as ow
    from oracle.gradcam import gradcam
    from oracle.peranet import OraclePeraNet
    from self_supervised import tools
    ref = OraclePeraNet(); ref.load_state_dict(seeded_sd)
    x = ow.synthetic_images(4, 64, seed=305)
    y_hat = torch.tensor([0, 2, 0, 1])
    want = torch.cat([torch.zeros(1, 1, 64, 64) if int(c) == 0 else gradcam(ref, x[i:i + 1], int(c)) for i, c in enumerate(y_hat)])
    got = tools.gradcam_maps(cam.localizer, x, y_hat, chunk=1)
    np.testing.assert_allclose(got.cpu().numpy(), torch.nan_to_num(want).numpy(), atol=TOL, rtol=0)
    assert float(got[0].abs().sum()) == 0.0 and float(got[2].abs().sum()) == 0.0


def test_gradcam_errors(cam):
    with pytest.raises(NotImplementedError):
        cam(torch.zeros(1, 3, 64, 96))
    with pytest.raises(ValueError):
        cam(torch.zeros(1, 1, 64, 64))
    cam.localizer.enable_patch_level_mode()
    try:
        with pytest.raises(RuntimeError):
            cam(torch.zeros(1, 3, 64, 64))
    finally:
        cam.localizer.disable_patch_level_mode()
