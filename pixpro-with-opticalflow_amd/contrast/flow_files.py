"""On-disk flow format feeding the hot path (the data format on the input side).

One file per video and direction: `<flow_root>/<prefix>/{forward,backward}/<video>.pth`, a serialised
`Tensor[n_video-1, 2, h, w]` at 1/8 resolution (units: 1/8-res pixels).  `forward[t]` is the flow frame t -> t+1,
`backward[s]` is the flow of the time-REVERSED clip, i.e. frame (n_video-1-s) -> (n_video-2-s).  A clip that starts
at frame `local_i` and spans `n_frames` frames therefore uses

    forward [local_i : local_i + n_frames-1]          (reference contrast/data/dataset.py:149-153)
    backward[num_flow - local_i - (n_frames-1) : num_flow - local_i]   (:151-152, contrast/data/dataset.py:357-361)

Function names follow the reference (`load_flow` dataset.py:343, `calc_bwd_idx` :357, `load_flows` :364); the path
rules are those of contrast/data/__init__.py:15-34.  `FlowClipBatch` assembles the `[N,F-1,2,h,w]` pair that
`util.apply_optical_flow` (pixpro_flow_prepare) consumes and moves it to the device in one pinned copy per direction.
Files are read with `weights_only=True`: nothing from a flow file is executed.
"""
import os

import torch

EXT = ".pth"


def flow_roots(args, prefix="train"):
    """(fwd_root, bwd_root) and the same side effects on `args` as contrast/data/__init__.py:15-34."""
    if not getattr(args, "use_flow_file", False):
        return "", ""
    root = args.flow_root
    if root is None or root == "":
        root = os.path.join(os.path.dirname(args.data_dir), "flow", "pth")
    root = os.path.join(root, prefix)
    args.flow_root = root
    args.fwd_name = args.fwd_name or "forward"
    args.bwd_name = args.bwd_name or "backward"
    return os.path.join(root, args.fwd_name), os.path.join(root, args.bwd_name)


def clip_window(local_i, n_video, n_frames):
    """Index arithmetic of the dataset item (dataset.py:134-152): returns (fwd_s, fwd_n, bwd_s, bwd_n) for a clip
    of min(n_frames, n_video) frames starting at frame local_i of a video with n_video frames."""
    n_frames = n_frames if n_video >= n_frames else n_video
    if not 0 <= local_i <= n_video - n_frames:
        raise IndexError(f"clip start {local_i} outside 0..{n_video - n_frames}")
    num_flow = n_video - 1
    span = n_frames - 1
    bwd_n = num_flow - local_i
    return local_i, local_i + span, bwd_n - span, bwd_n


def calc_bwd_idx(fwd_s_idx, fwd_n_idx, num_flow):
    span = fwd_n_idx - fwd_s_idx
    bwd_n_idx = num_flow - fwd_s_idx
    return bwd_n_idx - span, bwd_n_idx


def _read(path):
    ext = os.path.splitext(os.path.basename(path))[-1]
    if ext != EXT:
        raise NotImplementedError(f"{ext} is not supported!!")
    t = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(t, torch.Tensor) or t.dim() != 4 or t.shape[1] != 2:
        raise ValueError(f"{path}: expected a Tensor[n_frames-1, 2, h, w], got "
                         f"{tuple(t.shape) if isinstance(t, torch.Tensor) else type(t).__name__}")
    return t


def load_flow(path, s_idx, n_idx, return_num=True):
    flows = _read(path)
    if return_num:
        return flows[s_idx:n_idx], flows.shape[0]
    return flows[s_idx:n_idx]


def load_flows(fwd_pathes, bwd_pathes):
    """([fwd_path, s, n], [bwd_path, _, _]) -> (flow_fwd[n-s,2,h,w], flow_bwd[n-s,2,h,w]); the backward window is
    re-derived from the forward one and the file's own length, as the reference does (dataset.py:364-370)."""
    _, fwd_s, fwd_n = fwd_pathes
    bwd_path = bwd_pathes[0]
    flow_fwd, num_flow = load_flow(*fwd_pathes, return_num=True)
    bwd_s, bwd_n = calc_bwd_idx(fwd_s, fwd_n, num_flow)
    return flow_fwd, load_flow(bwd_path, bwd_s, bwd_n, return_num=False)


def video_flow_paths(fwd_root, bwd_root, frame_path):
    """Flow files of the video a frame belongs to (dataset.py:141-147): named after the frame's directory."""
    name = os.path.basename(os.path.dirname(frame_path)) + EXT
    fwd, bwd = os.path.join(fwd_root, name), os.path.join(bwd_root, name)
    if not os.path.isfile(fwd):
        raise FileNotFoundError(f"not exist path fwd path {fwd}")
    if not os.path.isfile(bwd):
        raise FileNotFoundError(f"not exist path bwd path {bwd}")
    return fwd, bwd


class FlowClipBatch:
    """Collates per-sample flow windows into the `[N,F-1,2,h,w]` pair of `data[5]` and uploads them with one
    pinned, non-blocking copy per direction (the reference's default collate + two `.cuda()` calls,
    main_pretrain.py:214-221)."""

    def __init__(self, n, span, h, w, pin=None):
        pin = torch.cuda.is_available() if pin is None else pin
        self.fwd = torch.empty((n, span, 2, h, w), dtype=torch.float32, pin_memory=pin)
        self.bwd = torch.empty_like(self.fwd, pin_memory=pin)
        self.count = 0

    def add(self, fwd_pathes, bwd_pathes):
        f, b = load_flows(fwd_pathes, bwd_pathes)
        if f.shape != self.fwd.shape[1:] or b.shape != self.bwd.shape[1:]:
            raise ValueError(f"flow window {tuple(f.shape)}/{tuple(b.shape)} does not match the batch "
                             f"{tuple(self.fwd.shape[1:])} (videos shorter than n_frames need their own batch)")
        if self.count >= self.fwd.shape[0]:
            raise IndexError("batch is full")
        self.fwd[self.count].copy_(f)
        self.bwd[self.count].copy_(b)
        self.count += 1

    def to(self, device):
        if self.count != self.fwd.shape[0]:
            raise ValueError(f"batch holds {self.count} of {self.fwd.shape[0]} samples")
        return self.fwd.to(device, non_blocking=True), self.bwd.to(device, non_blocking=True)
