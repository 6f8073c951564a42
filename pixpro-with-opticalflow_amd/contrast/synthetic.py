"""Synthetic batches with the record layout of the reference's video loader
(contrast/data/dataset.py:485-506; coord layout contrast/data/transform_coord.py:208-210):

    [im1[N,3,S,S], im2[N,3,S,S], coord1[N,10], coord2[N,10], index[N],
     [target[N], flow_fwds[N,F-1,2,h,w], flow_bwds[N,F-1,2,h,w]], [size[N,2], num_img[N,1]]]

Definition: crops ~ N(0,1); coords by the RandomResizedCropCoord.get_params rule on a
Wo x Ho frame with p=0.5 flips; low-res flows = 5x5-box-smoothed N(0, sigma^2) at 1/8 resolution,
backward stack = negated, time-reversed forward stack.  There is no dataset or network in the build.
"""
import math
import random

import torch
import torch.nn.functional as F


def crop_coords(n, Wo, Ho, rng, flip=True, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.)):
    rows = []
    for _ in range(n):
        area = Ho * Wo
        box = None
        for _ in range(10):
            target = rng.uniform(*scale) * area
            aspect = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
            w = int(round(math.sqrt(target * aspect)))
            h = int(round(math.sqrt(target / aspect)))
            if 0 < w <= Wo and 0 < h <= Ho:
                box = (rng.randint(0, Ho - h), rng.randint(0, Wo - w), h, w)
                break
        if box is None:  # whole-frame fallback (1280x720 is inside the aspect range)
            box = (0, 0, Ho, Wo)
        i, j, h, w = box
        row = [j / (Wo - 1), i / (Ho - 1), (j + w - 1) / (Wo - 1), (i + h - 1) / (Ho - 1), j, i, w, h, Wo, Ho]
        if flip and rng.random() < 0.5:
            row[0], row[2] = row[2], row[0]
        rows.append(row)
    return torch.tensor(rows, dtype=torch.float32)


def smooth_noise(shape, sigma, gen, device):
    f = torch.randn(*shape, generator=gen, device=device) * sigma
    lead = f.shape[:-2]
    return F.avg_pool2d(f.reshape(-1, 1, *f.shape[-2:]), 5, 1, 2).reshape(*lead, *f.shape[-2:])


def make_batch(n, image_size=224, n_frames=2, frame_hw=(720, 1280), seed=1234, device="cuda", sigma=0.5,
               flow_up=True, with_flow=True, channels_last=False, with_frames=False):
    Ho, Wo = frame_hw
    gen = torch.Generator(device=device).manual_seed(seed)
    rng = random.Random(seed)
    im1 = torch.randn(n, 3, image_size, image_size, generator=gen, device=device)
    im2 = torch.randn(n, 3, image_size, image_size, generator=gen, device=device)
    if channels_last:
        im1 = im1.contiguous(memory_format=torch.channels_last)
        im2 = im2.contiguous(memory_format=torch.channels_last)
    coord1 = crop_coords(n, Wo, Ho, rng).to(device)
    coord2 = crop_coords(n, Wo, Ho, rng).to(device)
    index = torch.arange(n, device=device)
    target = torch.zeros(n, dtype=torch.long, device=device)
    size = torch.tensor([Ho, Wo], device=device).repeat(n, 1)
    num_img = torch.full((n, 1), n_frames, device=device)
    if with_frames:
        # online flow mode: the original frames (0..255, smooth, each a slightly shifted copy of the previous one) travel
        # in data[6][2:] as load_img_for_raft yields them (reference contrast/data/dataset.py:337-340,493-497)
        base = torch.nn.functional.interpolate(torch.rand(n, 3, Ho // 16 + 2, Wo // 16 + 2, generator=gen, device=device),
                                               size=(Ho + n_frames, Wo + 2 * n_frames), mode="bilinear",
                                               align_corners=True) * 255.0
        # frame t: the scene moved by (+2, -1) pixels per frame
        frames = [base[:, :, t:t + Ho, 2 * (n_frames - 1 - t):2 * (n_frames - 1 - t) + Wo].contiguous()
                  for t in range(n_frames)]
        return [im1, im2, coord1, coord2, index, target, [size, num_img] + frames]
    if not with_flow:
        return [im1, im2, coord1, coord2, index, target, [size, num_img]]
    h, w = (Ho // 8, Wo // 8) if flow_up else (Ho, Wo)
    fwd = smooth_noise((n, n_frames - 1, 2, h, w), sigma, gen, device)
    bwd = -fwd.flip(1)
    return [im1, im2, coord1, coord2, index, [target, fwd.contiguous(), bwd.contiguous()], [size, num_img]]


class SyntheticLoader:
    """Iterates `steps` copies of a small pool of pre-generated device batches (a stand-in for the
    DataLoader + H2D copies; generation is outside the timed step)."""

    def __init__(self, steps, batch_size, pool=2, **kw):
        self.steps = steps
        seed = kw.pop("seed", 1234)
        self.batches = [make_batch(batch_size, seed=seed + 7919 * i, **kw) for i in range(pool)]
        self.sampler = None
        self.dataset = range(0)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.batches[i % len(self.batches)]
