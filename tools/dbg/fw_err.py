import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch, torch.nn.functional as F
from self_supervised import ops
dev = torch.device("cuda:0")
for (n, h, w, cin, cout) in [(32, 64, 64, 64, 64), (52, 32, 32, 128, 128)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g)
    w32 = torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5
    want = F.conv2d(x.double().permute(0, 3, 1, 2), w32.double().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    t32 = F.conv2d(x.permute(0, 3, 1, 2), w32.permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    xd, wd = x.to(dev), w32.to(dev)
    wp, _ = ops.conv3x3_hw_pack(wd.reshape(-1), [(0, cout, cin, False)], f32=True)
    z = ops.conv3x3_hw(xd, wp, cout).cpu().double()
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    zi = ops.conv_fwd_stats(xd, wd, 1e-5, 0.1, rm, rv, 1, 1)[0].cpu().double()
    for name, t in (("new", z), ("igemm", zi), ("torch-cpu fp32", t32.double())):
        e = (t - want)
        print((n, h, w, cin, cout), name, "rms err %.3e  max %.3e  mean(signed) %.3e  sum-over-pixels rel err %.3e" % (
            e.pow(2).mean().sqrt().item(), e.abs().max().item(), e.mean().item(),
            ((t.sum((0, 1, 2)) - want.sum((0, 1, 2))).norm() / want.sum((0, 1, 2)).norm()).item()))
