"""Oracle: sliding-window patches, cosine 3-NN scoring, blur + bilinear upsample.

Follows src/self_supervised/functional.py:77-82 (extract_patches),
src/self_supervised/models.py:345-370 (AnomalyDetector),
src/self_supervised/tools.py:394-399 (upsample) of the reference.
"""
import numpy as np
import torch
import torch.nn.functional as F


def extract_patches(image, dim=32, stride=4):
    """functional.py:77-82.  (B,C,H,W) -> (B,P,C,dim,dim); patch p = r*ncols + c covers
    rows [stride*r, stride*r+dim) and cols [stride*c, stride*c+dim) (column index fastest)."""
    b, c, h, w = image.shape
    nr = (h - dim) // stride + 1
    nc = (w - dim) // stride + 1
    ys = (torch.arange(nr) * stride)[:, None] + torch.arange(dim)[None, :]       # (nr, dim)
    xs = (torch.arange(nc) * stride)[:, None] + torch.arange(dim)[None, :]       # (nc, dim)
    # gather rows then cols: (B,C,nr,dim,W) -> (B,C,nr,dim,nc,dim)
    t = image[:, :, ys, :]
    t = t[:, :, :, :, xs]
    t = t.permute(0, 2, 4, 1, 3, 5)                                              # B,nr,nc,C,dim,dim
    return t.reshape(b, nr * nc, c, dim, dim)


# ---------------------------------------------------------------------------
# cosine k-NN (models.py:352-370; sklearn NearestNeighbors(metric='cosine') -> brute force)
# ---------------------------------------------------------------------------
def cosine_knn_mean(bank, queries, k=3):
    """numpy fp32 restatement of sklearn's brute cosine kneighbors + torch.mean(dim=1).

    sklearn: Xn = X/||X||, Yn = Y/||Y||; D = clip(1 - Xn @ Yn.T, 0, 2); k smallest per row,
    sorted ascending; the reference then averages the k distances (models.py:366-368).
    Returns (mean (Nq,), dists (Nq,k), idx (Nq,k)).
    """
    b = np.asarray(bank, dtype=np.float32)
    q = np.asarray(queries, dtype=np.float32)
    bn = b / np.sqrt(np.einsum("ij,ij->i", b, b))[:, None]
    qn = q / np.sqrt(np.einsum("ij,ij->i", q, q))[:, None]
    d = 1.0 - qn @ bn.T
    np.clip(d, 0, 2, out=d)
    idx = np.argsort(d, axis=1, kind="stable")[:, :k]
    dk = np.take_along_axis(d, idx, axis=1)
    return dk.mean(axis=1, dtype=np.float32), dk, idx


def sklearn_knn_mean(bank, queries, k=3):
    """The reference's exact call sequence on the library it uses (present in this image)."""
    from sklearn.neighbors import NearestNeighbors
    nbrs = NearestNeighbors(n_neighbors=k, algorithm="auto", metric="cosine").fit(np.asarray(bank))
    d = nbrs.kneighbors(np.asarray(queries))[0].squeeze()
    d = torch.tensor(d)
    return torch.mean(d, dim=1) if k > 1 else d


class OracleAnomalyDetector:
    """models.py:345-370 with the train/val split made explicit and seedable (quirk Q5)."""

    def __init__(self, patch_level=False, batch=None, num_patches=None):
        self.patch_level = patch_level
        self.batch = batch
        self.dim = int(np.sqrt(num_patches)) if num_patches else None

    def fit(self, embeddings, split=True):
        from sklearn.model_selection import train_test_split as tts
        emb = np.asarray(embeddings, dtype=np.float32)
        self.k = 3
        if split:
            train, val = tts(emb, test_size=0.3)
        else:
            train, val = emb, emb
        self.bank = train
        scores, _, _ = cosine_knn_mean(train, val, self.k)
        self.threshold = float(scores.max())

    def predict(self, x):
        scores, _, _ = cosine_knn_mean(self.bank, np.asarray(x, dtype=np.float32), self.k)
        s = torch.from_numpy(scores)
        if self.patch_level:
            s = s.reshape(self.batch, 1, self.dim, self.dim)
        return s


# ---------------------------------------------------------------------------
# upsample (tools.py:394-399)
# ---------------------------------------------------------------------------
def gaussian_kernel1d(ksize=7, sigma=None):
    """Restated, third-party (torchvision.transforms.functional.gaussian_blur):
    sigma defaults to 0.15*k + 0.35; taps exp(-0.5 (x/sigma)^2) on linspace(-(k-1)/2,(k-1)/2,k), normalised."""
    if sigma is None:
        sigma = 0.15 * ksize + 0.35
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(maps, kernel_size=7):
    """Restated, third-party: reflect-pad k//2, depth-wise conv with the outer-product kernel."""
    k1 = gaussian_kernel1d(kernel_size).to(maps.dtype)
    k2 = torch.mm(k1[:, None], k1[None, :])
    c = maps.shape[1]
    w = k2.expand(c, 1, kernel_size, kernel_size)
    p = kernel_size // 2
    x = F.pad(maps, [p, p, p, p], mode="reflect")
    return F.conv2d(x, w, groups=c)


def upsample(anomaly_maps, target_size=256):
    """tools.py:394-399: relu(gaussian_blur(k=7)) then bilinear (align_corners=False)."""
    m = F.relu(gaussian_blur(anomaly_maps, 7))
    return F.interpolate(m, target_size, mode="bilinear")


def bilinear_loops(m, out):
    """Pure-python statement of F.interpolate(mode='bilinear', align_corners=False) for one
    2-D map (small cases only): src = (dst+0.5)*in/out - 0.5 clamped at 0, neighbours clamped."""
    m = np.asarray(m, dtype=np.float32)
    h, w = m.shape
    o = np.zeros((out, out), np.float32)
    sh, sw = np.float32(h / out), np.float32(w / out)
    for y in range(out):
        sy = max(np.float32(sh * np.float32(y + 0.5) - np.float32(0.5)), np.float32(0))
        y0 = int(sy); y1 = min(y0 + 1, h - 1); ly = np.float32(sy - y0)
        for x in range(out):
            sx = max(np.float32(sw * np.float32(x + 0.5) - np.float32(0.5)), np.float32(0))
            x0 = int(sx); x1 = min(x0 + 1, w - 1); lx = np.float32(sx - x0)
            o[y, x] = (np.float32(1) - ly) * ((np.float32(1) - lx) * m[y0, x0] + lx * m[y0, x1]) \
                + ly * ((np.float32(1) - lx) * m[y1, x0] + lx * m[y1, x1])
    return o
