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
