import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops
dev = torch.device("cuda:0")
for (n, h, w, cin, cout) in [(26, 64, 64, 64, 64), (52, 32, 32, 128, 128)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g).half().to(dev)
    w32 = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).to(dev)
    wp, _ = ops.conv3x3_hw_pack(w32.reshape(-1), [(0, cout, cin, False)])
    tr = tuple((torch.rand(cin, generator=g) + 0.5).to(dev) for _ in range(4))
    act = ops.bn_apply_fwd(x, tr[0], tr[1], tr[2], tr[3], None, True)
    z2, em = ops.conv3x3_hw(x, wp, cout, transform=tr, emit=True)
    bad = (em != act)
    print((n, h, w, cin, cout), "mismatches", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print("channels with mismatches:", sorted(set(idx[:, 3].tolist()))[:70])
        print("images:", sorted(set(idx[:, 0].tolist()))[:20], "rows", sorted(set(idx[:, 1].tolist()))[:20], "cols", sorted(set(idx[:, 2].tolist()))[:40])
        i = idx[0].tolist(); print(i, float(em[tuple(i)]), float(act[tuple(i)]), float(x[tuple(i)]), [float(t[i[3]]) for t in tr])
        d = (em.float() - act.float()).abs().max().item(); print("max abs diff", d)
