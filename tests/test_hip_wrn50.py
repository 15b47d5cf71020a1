"""BASELINE.json configs[3] (WideResNet-50-2 layer1-3 feature-distance maps): HIP path against its torch-CPU restatement
(oracle/wrn50.py).  The reference has no such model: the definition is ours, the parity bar is the path's (2e-5 relative for the
features, 1e-4 for the maps)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_wrn50_features_and_distance_maps():
    from oracle import wrn50 as ow
    from self_supervised.wrn50 import FeatureDistanceScorer, WideResNet50Features
    from oracle import weights as w
    dev = torch.device("cuda:0")
    ref = ow.seeded_trunk(0)
    m = WideResNet50Features()
    m.load_state_dict(ref.state_dict(), strict=True)
    m.to(dev).eval()
    x = w.synthetic_images(2, 128, seed=21)
    with torch.no_grad():
        want = ref(x)
        got = m(x.to(dev))
    assert [tuple(f.shape) for f in got] == [(2, 32, 32, 256), (2, 16, 16, 512), (2, 8, 8, 1024)]
    for g, r in zip(got, want):
        r = r.permute(0, 2, 3, 1)
        err = (g.cpu() - r).abs().max().item()
        assert err <= 2e-5 * max(1.0, r.abs().max().item()), (tuple(r.shape), err, r.abs().max().item())
    banks = ow.seeded_banks(64)
    maps = FeatureDistanceScorer([b.to(dev) for b in banks])(got, 128)
    ref_maps = ow.distance_maps(want, banks, 128)
    assert tuple(maps.shape) == (2, 1, 128, 128)
    assert (maps.cpu() - ref_maps).abs().max().item() < 1e-4
    # a second size exercises odd maps (the stride-2 3x3 convs and the pool on 72 -> 36 -> 18 -> 9 -> 5)
    x2 = w.synthetic_images(1, 72, seed=22)
    with torch.no_grad():
        want2, got2 = ref(x2), m(x2.to(dev))
    for g, r in zip(got2, want2):
        r = r.permute(0, 2, 3, 1)
        assert (g.cpu() - r).abs().max().item() <= 2e-5 * max(1.0, r.abs().max().item())


def test_wrn50_full_size_batch_and_one_image_against_the_oracle():
    """BASELINE configs[3] at its stated size: 64 images of 512 x 512 in one call (the bench's workload).  Every map is finite and
    non-negative, the map of an image does not depend on its position in the batch or on the batch around it (same kernels,
    another grid: compared at the parity bar, not bit for bit), and one 512 x 512 image matches oracle/wrn50.py at 1e-4."""
    from oracle import wrn50 as ow
    from oracle import weights as w
    from self_supervised.wrn50 import FeatureDistanceScorer, WideResNet50Features
    dev = torch.device("cuda:0")
    ref = ow.seeded_trunk(0)
    m = WideResNet50Features()
    m.load_state_dict(ref.state_dict(), strict=True)
    m.to(dev).eval()
    banks = ow.seeded_banks(588)
    scorer = FeatureDistanceScorer([b.to(dev) for b in banks])
    x8 = w.synthetic_images(8, 512, seed=31)
    x = torch.cat([x8[i % 8:i % 8 + 1] for i in range(64)]).to(dev)          # image j sits at positions j, j + 8, ...
    with torch.no_grad():
        feats = m(x)
        maps = scorer(feats, 512)
    assert [tuple(f.shape) for f in feats] == [(64, 128, 128, 256), (64, 64, 64, 512), (64, 32, 32, 1024)]
    assert tuple(maps.shape) == (64, 1, 512, 512)
    assert bool(torch.isfinite(maps).all()) and float(maps.min()) >= 0.0
    scale = max(1.0, float(maps.abs().max()))
    assert float((maps[:8] - maps[56:]).abs().max()) <= 1e-6 * scale            # same image, another batch position
    with torch.no_grad():
        alone = scorer(m(x[3:4].contiguous()), 512)                              # the same image in a batch of one
        want = ow.distance_maps(ref(x8[3:4]), banks, 512)
    assert float((alone - maps[3:4]).abs().max()) <= 1e-5 * scale
    err = float((maps[3:4].cpu() - want).abs().max())
    assert err < 1e-4, err
