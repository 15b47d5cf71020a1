"""Oracle: seeded random weights and synthetic inputs.

No ImageNet weights, checkpoints or MVTec images exist offline, so
parity is proven on seeded random weights + synthetic images.  Everything here
is generated with the torch CPU generator only, so the GPU box (same image,
same torch build) regenerates bit-identical tensors instead of shipping 50 MB.
"""
import torch
from torch import nn

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def seeded_state_dict(seed=0, layer_outputs=("layer2", "layer3"), latent_space_layers=5):
    """Reference-named state_dict (feature_extractor.*, concatenator.*, latent_space.*, classifier.*).

    Convs: kaiming-normal fan_out/relu (the torchvision ResNet init, keeps the signal alive through
    20 layers); Linears: torch default; BN: gamma~U(.8,1.2), beta~N(0,.1), running_mean~N(0,.1),
    running_var~U(.5,1.5) so eval-mode BN is exercised with non-trivial statistics.
    """
    from .peranet import OraclePeraNet
    g = torch.Generator().manual_seed(seed)
    with torch.random.fork_rng():
        torch.manual_seed(seed)
        m = OraclePeraNet(layer_outputs=layer_outputs, latent_space_layers=latent_space_layers)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.Conv2d):
                fan_out = mod.out_channels * mod.kernel_size[0] * mod.kernel_size[1]
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / fan_out) ** 0.5)
            elif isinstance(mod, nn.Linear):
                bound = 1.0 / mod.in_features ** 0.5
                mod.weight.copy_((torch.rand(mod.weight.shape, generator=g) * 2 - 1) * bound)
                if mod.bias is not None:
                    mod.bias.copy_((torch.rand(mod.bias.shape, generator=g) * 2 - 1) * bound)
            elif isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
                n = mod.num_features
                mod.weight.copy_(0.8 + 0.4 * torch.rand(n, generator=g))
                mod.bias.copy_(0.1 * torch.randn(n, generator=g))
                mod.running_mean.copy_(0.1 * torch.randn(n, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(n, generator=g))
    return {k: v.clone() for k, v in m.state_dict().items()}


def synthetic_images(n, size=256, seed=1234, normalized=True):
    """uint8 noise, 3x3 box low-pass, ToTensor, ImageNet Normalize -> (n,3,size,size) fp32 NCHW."""
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (n, 3, size, size), generator=g, dtype=torch.int32).float()
    k = torch.ones(3, 1, 3, 3) / 9.0
    u8 = torch.nn.functional.conv2d(torch.nn.functional.pad(u8, [1, 1, 1, 1], mode="replicate"), k, groups=3)
    x = u8.round().clamp(0, 255) / 255.0
    if normalized:
        mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
        x = (x - mean) / std
    return x.contiguous()


def synthetic_labels(n, seed=1235, num_classes=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, num_classes, (n,), generator=g, dtype=torch.int64)


def synthetic_bank(rows=588, dim=512, seed=2):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(rows, dim, generator=g)
