"""Seeded input recipes shared by make_golden.py (reference side) and the tests / bench
(this repository's side).  Pure data generators: nothing here computes a hot-path result."""
import math
import zlib

import torch


def tag_seed(tag):
    return zlib.crc32(tag.encode()) % 100000


def crop_coords(n, Wo, Ho, rng, flip=True, scale=(0.08, 1.0)):
    """Crop records [N,10] = [x0/(Wo-1), y0/(Ho-1), x1/(Wo-1), y1/(Ho-1), j, i, w, h, Wo, Ho] with
    the sampling rule of RandomResizedCropCoord.get_params (contrast/data/transform_coord.py:156-210:
    area scale U(scale), log-uniform aspect 3/4..4/3, 10 tries, whole-image fallback) and the
    horizontal flip encoded as swapped x0/x2 (transform_coord.py:88-93).  rng: random.Random."""
    out = []
    for _ in range(n):
        area = Ho * Wo
        for _ in range(10):
            ta = rng.uniform(*scale) * area
            ar = math.exp(rng.uniform(math.log(3 / 4), math.log(4 / 3)))
            w = int(round(math.sqrt(ta * ar)))
            h = int(round(math.sqrt(ta / ar)))
            if 0 < w <= Wo and 0 < h <= Ho:
                i = rng.randint(0, Ho - h)
                j = rng.randint(0, Wo - w)
                break
        else:
            w, h = Wo, Ho
            i = j = 0
        c = [j / (Wo - 1), i / (Ho - 1), (j + w - 1) / (Wo - 1), (i + h - 1) / (Ho - 1),
             j, i, w, h, Wo, Ho]
        if flip and rng.random() < 0.5:
            c[0], c[2] = c[2], c[0]
        out.append(c)
    return torch.tensor(out, dtype=torch.float32)


def smooth_flow(shape, sigma, gen, device="cpu"):
    """N(0, sigma^2) noise smoothed by a 5x5 box filter (stride 1, pad 2) over the last two dims."""
    f = torch.randn(*shape, generator=gen, device=device) * sigma
    lead = f.shape[:-2]
    f = torch.nn.functional.avg_pool2d(f.reshape(-1, 1, *f.shape[-2:]), 5, 1, 2)
    return f.reshape(*lead, *f.shape[-2:])


def seeded_state_dict(shapes, seed):
    """Deterministic non-degenerate values for a {name: shape} map (any PixPro state dict):
    conv/linear weights ~ N(0, 2/fan_in), norm gains 1+0.1 N, biases/means 0.1 N,
    running_var 1+0.2 U, counters 0.  Iterates names in sorted order."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=torch.long)
        elif name.endswith("running_var"):
            out[name] = 1 + 0.2 * torch.rand(shape, generator=gen)
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[name] = torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_in)
        elif name.endswith("weight"):
            out[name] = 1 + 0.1 * torch.randn(shape, generator=gen)
        else:
            out[name] = 0.1 * torch.randn(shape, generator=gen)
    return out


def raft_state_dict(shapes, seed):
    """Seeded RAFT weights: `seeded_state_dict` with the last flow-head convolution scaled down so that twelve
    refinement iterations of a random network stay in a well-conditioned range (|flow| of a few 1/8-res pixels)."""
    sd = seeded_state_dict(shapes, seed)
    for k in sd:
        if "flow_head.conv2" in k:
            sd[k] = sd[k] * 0.05
    return sd


def raft_images(n, h, w, seed):
    """A smooth random frame pair in 0..255 (second frame = first shifted by a few pixels + noise)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.rand(n, 3, h // 8 + 2, w // 8 + 2, generator=g), size=(h + 8, w + 8),
                                           mode="bilinear", align_corners=True) * 255.0
    im1 = base[:, :, 4:4 + h, 4:4 + w].contiguous()
    im2 = (base[:, :, 2:2 + h, 6:6 + w] + 2.0 * torch.randn(n, 3, h, w, generator=g)).clamp(0, 255).contiguous()
    return im1, im2
