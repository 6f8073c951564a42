"""Golden vectors for the flow preparation at output sizes ABOVE ATen's small-output threshold
(output_H + output_W > 128), generated from the reference's own Python (build container only).

Run:  python tests/golden/make_golden_up8big.py     (needs /root/reference; never runs on the GPU box)

Why a second generator: round 1's fixtures are all <= 48x64.  torch's CPU bilinear up-sampling (the reference's
`upflow8` = 8 * F.interpolate(..., align_corners=True), contrast/flow/utils/utils.py:87-89) switches kernels - and with
them the floating-point operation order - at output_H + output_W = 128, so every production size (720x1280) takes the
path those fixtures never exercised.  These pin that path, the threshold itself, and the pipeline built on it
(`util.apply_optical_flow`, contrast/util.py:174-248) for nframe 2 and 6.  Only DATA is written.
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, save  # noqa: E402
from recipes import smooth_flow  # noqa: E402


def main():
    P, util, resnet, lars = load_reference()
    torch.set_num_threads(4)
    from contrast.flow.utils.utils import upflow8
    for tag, (M, h, w) in {"big": (2, 9, 12), "edge128": (1, 8, 8), "edge136": (1, 9, 8), "edge136w": (1, 7, 10)}.items():
        g = torch.Generator().manual_seed(100 + h * w)
        lo = smooth_flow((M, 2, h, w), 0.5, g)
        save(f"upflow8_{tag}", flow=lo, out=upflow8(lo))
    for tag, (Fr, N, h, w, sig) in {"f2up_big": (2, 2, 8, 10, 0.6), "f6up_big": (6, 1, 9, 12, 0.6),
                                    "f2up_bigmotion": (2, 1, 9, 12, 2.5)}.items():
        g = torch.Generator().manual_seed(140 + Fr + h)
        fw = smooth_flow((N, Fr - 1, 2, h, w), sig, g)
        bw = -fw.flip(1) + smooth_flow((N, Fr - 1, 2, h, w), 0.15 * sig, g)
        size = torch.tensor([8 * h, 8 * w])
        data = [None, None, None, None, None, [torch.zeros(N), fw, bw], [size.repeat(N, 1), torch.full((N, 1), Fr)]]
        args = types.SimpleNamespace(alpha1=0.01, alpha2=0.5, use_flow_frames=False, use_flow_file=True, flow_up=True,
                                     flow_cat_norm=False, debug=False)
        (f, s, mf), (b, _, mb) = util.apply_optical_flow(data, None, args)
        print(tag, "valid ratio", float(mf.float().mean()), float(mb.float().mean()))
        save(f"applyflow_{tag}", flow_fwds=fw, flow_bwds=bw, flow_up=True, alpha1=0.01, alpha2=0.5,
             flow_fwd=f, flow_bwd=b, mask_fwd=mf, mask_bwd=mb, size=s)


if __name__ == "__main__":
    main()
