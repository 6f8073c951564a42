"""Command-line surface of the pre-training entry point: the flag set of the reference's
contrast/option.py:10-121 (pre-train stage), table-driven.  Extra flags of this build are listed
last (`--synthetic`, `--steps-per-epoch`, `--no-sync-bn`)."""
import argparse
import os

from . import resnet

model_names = sorted(n for n in resnet.__all__ if n.islower())

# (flags, kwargs)
_FLAGS = [
    # dataset
    (("--data-dir",), dict(type=str, default="./data")),
    (("--crop",), dict(type=float, default=0.2)),
    (("--crop-ratio",), dict(nargs=2, type=float, default=[3. / 4., 4. / 3.])),
    (("--aug",), dict(type=str, default="NULL",
                      choices=["NULL", "InstDisc", "MoCov2", "SimCLR", "RandAug", "BYOL", "val"])),
    (("--zip",), dict(action="store_true")),
    (("--cache-mode",), dict(type=str, default="part", choices=["no", "full", "part"])),
    (("--dataset",), dict(type=str, default="ImageNet", choices=["ImageNet", "bdd100k"])),
    (("--ann-file",), dict(type=str, default="")),
    (("--image-size",), dict(nargs=2, type=int, default=[224, 224])),
    (("--num-workers",), dict(type=int, default=4)),
    (("--n-frames",), dict(type=int, default=1)),
    # optical flow
    (("--use_flow",), dict(action="store_true")),
    (("--flow_model",), dict(default="")),
    (("--flow_up",), dict(action="store_true")),
    (("--alpha1",), dict(type=float, default=None)),
    (("--alpha2",), dict(type=float, default=None)),
    (("--flow_cat_norm",), dict(action="store_true")),
    (("--flow_bs",), dict(type=int, default=None)),
    (("--use_flow_frames",), dict(action="store_true")),
    (("--use_flow_file",), dict(action="store_true")),
    (("--flow_root",), dict(type=str, default="")),
    (("--fwd_name",), dict(type=str, default="")),
    (("--bwd_name",), dict(type=str, default="")),
    (("--batch-size",), dict(type=int, default=64)),
    # model
    (("--arch",), dict(type=str, default="resnet50", choices=model_names)),
    (("--model",), dict(type=str, required=True)),
    (("--feature-dim",), dict(type=int, default=256)),
    (("--head-type",), dict(type=str, default="mlp_head")),
    # optimisation
    (("--base-learning-rate", "--base-lr"), dict(type=float, default=0.03)),
    (("--optimizer",), dict(type=str, choices=["sgd", "lars"], default="sgd")),
    (("--lr-scheduler",), dict(type=str, default="cosine", choices=["step", "cosine"])),
    (("--warmup-epoch",), dict(type=int, default=5)),
    (("--warmup-multiplier",), dict(type=int, default=100)),
    (("--lr-decay-epochs",), dict(type=int, default=[120, 160, 200], nargs="+")),
    (("--lr-decay-rate",), dict(type=float, default=0.1)),
    (("--weight-decay",), dict(type=float, default=1e-4)),
    (("--momentum",), dict(type=float, default=0.9)),
    (("--amp-opt-level",), dict(type=str, default="O1", choices=["O0", "O1", "O2"])),
    (("--start-epoch",), dict(type=int, default=1)),
    (("--epochs",), dict(type=int, default=100)),
    (("--debug-epochs",), dict(type=int, default=None)),
    (("--verbose",), dict(action="store_true")),
    (("--debug",), dict(action="store_true")),
    (("--no_benchmark",), dict(action="store_true")),
    # misc
    (("--output-dir",), dict(type=str, default="./output")),
    (("--auto-resume",), dict(action="store_true")),
    (("--resume",), dict(default="", type=str)),
    (("--print-freq",), dict(type=int, default=100)),
    (("--save-freq",), dict(type=int, default=10)),
    (("--local_rank", "--local-rank"), dict(type=int, default=None)),
    (("--pretrained-model",), dict(type=str, default="")),
    # PixPro
    (("--pixpro-p",), dict(type=float, default=1.)),
    (("--pixpro-momentum",), dict(type=float, default=0.99)),
    (("--pixpro-pos-ratio",), dict(type=float, default=0.7)),
    (("--pixpro-neg-ratio",), dict(type=float, default=1.0)),        # parsed, never read (as in the reference)
    (("--pixpro-neg-loss-weight",), dict(type=float, default=1.0)),  # parsed, never read
    (("--pixpro-ins-loss-weight",), dict(type=float, default=0.)),
    (("--pixpro-clamp-value",), dict(type=float, default=0.)),
    (("--pixpro-transform-layer",), dict(type=int, default=0)),
    # this build
    (("--synthetic",), dict(action="store_true", help="synthetic crops/coords/flows (SURVEY.md §8d) instead of BDD100k")),
    (("--steps-per-epoch",), dict(type=int, default=20, help="iterations per epoch with --synthetic")),
    (("--num-instances",), dict(type=int, default=70000, help="dataset length used for the EMA schedule with --synthetic")),
    (("--frame-size",), dict(nargs=2, type=int, default=[720, 1280], help="synthetic original frame (H W)")),
    (("--no-sync-bn",), dict(action="store_true")),
    (("--lazy-flow",), dict(action="store_true", help="evaluate flow chain + consistency mask inside the loss kernel at "
                                                      "its sample points instead of preparing dense tensors (same results)")),
    (("--channels-last",), dict(action="store_true")),
    (("--key-bn-group",), dict(action="store_true", help="SyncBN collectives of the key (momentum) encoder on a process "
                                "group of their own (second RCCL communicator): with several ranks the two branches of "
                                "--overlap-momentum then do not queue behind each other's statistics exchanges")),
    (("--overlap-momentum",), dict(action="store_true", help="run the key (momentum) branch on a second HIP stream, "
                                                             "concurrently with the online branch")),
    (("--fused-bn",), dict(action="store_true", help="HIP (Sync)BatchNorm with the ReLU / residual add folded in "
                                                     "(needs --channels-last)")),
    (("--bn-view-batch",), dict(action="store_true", help="run both views through an encoder in ONE pass with per-view "
                                                          "BatchNorm statistics (needs --fused-bn): same statistics, half "
                                                          "the SyncBN collectives per step")),
    (("--wgrad-stream",), dict(action="store_true", help="weight gradients of the encoder convolutions on a second HIP "
                                                         "stream, overlapping the next layers' BatchNorm backward "
                                                         "(contrast/wgrad_stream.py; one rank, use with --bn-view-batch)")),
]


def build_parser(stage="pre-train"):
    if stage != "pre-train":
        raise NotImplementedError("only the pre-training stage is part of this build")
    parser = argparse.ArgumentParser(f"contrast {stage} stage", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flags, kw in _FLAGS:
        parser.add_argument(*flags, **kw)
    return parser


def finalize(args):
    """Post-parse fix-ups of the reference (option.py:107-119) + LOCAL_RANK from the torchrun env."""
    if args.local_rank is None:
        args.local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if args.flow_model != "":
        if "small" in os.path.basename(args.flow_model):
            args.small = True
        args.mixed_precision = False
    if args.image_size[0] == args.image_size[1]:
        args.image_size = args.image_size[0]
    if args.debug_epochs is None:
        args.debug_epochs = args.epochs + 1
    args.sync_bn = not args.no_sync_bn
    return args


def parse_option(stage="pre-train", argv=None):
    return finalize(build_parser(stage).parse_args(argv))
