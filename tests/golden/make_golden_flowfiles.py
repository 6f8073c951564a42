"""Golden vectors for the on-disk flow format (build container only; needs /root/reference).

The three loader functions of the reference's contrast/data/dataset.py (`load_flow`, `calc_bwd_idx`, `load_flows`)
are taken out of the module source by name and executed here against flow files this script writes; the module
itself is not importable offline (pycocotools, torchvision).  Only data is stored: the file tensors, the clip
windows and what the reference returned for them.
"""
import ast
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from recipes import tag_seed  # noqa: E402
SRC = "/root/reference/contrast/data/dataset.py"
WANTED = ("load_flow", "calc_bwd_idx", "load_flows")


def reference_functions():
    tree = ast.parse(open(SRC).read(), SRC)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert len(body) == len(WANTED)
    ns = {"os": os, "torch": torch}
    exec(compile(ast.Module(body=body, type_ignores=[]), SRC, "exec"), ns)
    return ns


def main():
    ref = reference_functions()
    out = {}
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for vid, n_video in enumerate((2, 3, 6, 9)):
            g = torch.Generator().manual_seed(tag_seed(f"flowfile{n_video}"))
            fwd = torch.randn(n_video - 1, 2, 3, 4, generator=g)
            bwd = torch.randn(n_video - 1, 2, 3, 4, generator=g)
            fp, bp = os.path.join(tmp, f"f{vid}.pth"), os.path.join(tmp, f"b{vid}.pth")
            torch.save(fwd, fp)
            torch.save(bwd, bp)
            out[f"file_fwd_{n_video}"] = fwd.numpy()
            out[f"file_bwd_{n_video}"] = bwd.numpy()
            for n_frames in (2, 3, 6):
                nf = min(n_frames, n_video)
                for local_i in range(0, n_video - nf + 1):
                    s, n = local_i, local_i + nf - 1
                    # the reference's loader ignores the backward window it is handed (dataset.py:364-370)
                    f, b = ref["load_flows"]([fp, s, n], [bp, -1, -1])
                    bs, bn = ref["calc_bwd_idx"](s, n, n_video - 1)
                    k = len(cases)
                    cases.append([n_video, n_frames, local_i, s, n, bs, bn])
                    out[f"out_fwd_{k}"] = f.numpy()
                    out[f"out_bwd_{k}"] = b.numpy()
    out["cases"] = np.asarray(cases, dtype=np.int64)   # n_video, n_frames, local_i, fwd_s, fwd_n, bwd_s, bwd_n
    path = os.path.join(HERE, "flowfiles.npz")
    np.savez_compressed(path, **out)
    print(f"flowfiles: {len(cases)} cases, {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
