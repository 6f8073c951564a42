"""Pre-training entry point (DDP over RCCL, one process per GPU).

Same contract as the reference's main_pretrain.py: `build_model` (:38-83), `train` (:187-371),
`main` (:130-184), checkpoint save/resume (:96-127), flags of contrast/option.py.  Launch:

    torchrun --standalone --nproc-per-node=8 main_pretrain.py --model PixPro --arch resnet50 \
        --synthetic --use_flow --use_flow_file --flow_up --alpha1 0.01 --alpha2 0.5 --n-frames 2 \
        --optimizer lars --base-lr 1.0 --weight-decay 1e-5 --amp-opt-level O0 \
        --pixpro-p 2 --pixpro-transform-layer 1 --batch-size 128

Differences from the reference, all in SURVEY.md §0.2: D1 (`calc_frame_ratio` defined), D2 (`size`
passed through unchanged), D7 (device from LOCAL_RANK; backend "nccl" is RCCL on ROCm), D9 (the step
itself issues no host sync: logging reads scalars only every --print-freq steps).  The BDD100k video
loader is CPU-side data plumbing outside this build; `--synthetic` feeds batches of the same layout.
"""
import json
import math
import os
import time
from shutil import copyfile

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel

from contrast import models, resnet, util, wgrad_stream
from contrast.lars import LARS, add_weight_decay
from contrast.lr_scheduler import get_scheduler
from contrast.option import parse_option
from contrast.synthetic import SyntheticLoader


def log(msg):
    if not dist.is_initialized() or dist.get_rank() == 0:
        print(msg, flush=True)


def build_model(args, device):
    encoder = resnet.__dict__[args.arch]
    model = models.__dict__[args.model](encoder, args).to(device)
    if getattr(args, "channels_last", False):
        model = model.to(memory_format=torch.channels_last)
    world = dist.get_world_size() if dist.is_initialized() else 1
    lr = args.batch_size * world / 256 * args.base_learning_rate
    if args.optimizer == 'sgd':
        optimizer = torch.optim.SGD(model.parameters(), lr=lr, momentum=args.momentum,
                                    weight_decay=args.weight_decay)
    elif args.optimizer == 'lars':
        optimizer = LARS(add_weight_decay(model, args.weight_decay), lr=lr, momentum=args.momentum)
    else:
        raise NotImplementedError
    if args.amp_opt_level != "O0":
        raise NotImplementedError("the optical-flow recipes train in fp32 (apex amp O0); O1/O2 are not provided")
    if dist.is_initialized():
        ids = [device.index] if device.type == "cuda" else None
        model = DistributedDataParallel(model, device_ids=ids, broadcast_buffers=False)
    return model, optimizer


def save_checkpoint(args, epoch, model, optimizer, scheduler):
    state = {'opt': vars(args), 'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
             'scheduler': scheduler.state_dict(), 'epoch': epoch}
    file_name = os.path.join(args.output_dir, f'ckpt_epoch_{epoch}.pth')
    torch.save(state, file_name)
    copyfile(file_name, os.path.join(args.output_dir, 'current.pth'))


def _safe_load(path):
    """Checkpoints are untrusted pickles (a PixPro checkpoint from upstream is whatever its author saved): load them
    with torch's restricted unpickler only.  The reference stores `opt` as an argparse.Namespace (main_pretrain.py:118),
    which is plain data and is allow-listed; anything else non-tensor is refused by torch.load."""
    import argparse
    with torch.serialization.safe_globals([argparse.Namespace]):
        return torch.load(path, map_location='cpu', weights_only=True)


def load_checkpoint(args, model, optimizer, scheduler):
    ckpt = _safe_load(args.resume)
    args.start_epoch = ckpt['epoch'] + 1
    model.load_state_dict(ckpt['model'])
    optimizer.load_state_dict(ckpt['optimizer'])
    sched = ckpt['scheduler']
    if 'after_scheduler' in sched:
        # the reference's GradualWarmupScheduler layout (contrast/lr_scheduler.py): the closed-form schedule here needs
        # only the step count, which both layouts keep as `last_epoch`
        sched = {'last_epoch': sched['last_epoch']}
        scheduler.last_epoch = sched['last_epoch']
        for g, lr in zip(optimizer.param_groups, [base * f(scheduler.last_epoch)
                                                  for base, f in zip(scheduler.base_lrs, scheduler.lr_lambdas)]):
            g['lr'] = lr
        scheduler._last_lr = [g['lr'] for g in optimizer.param_groups]
    else:
        scheduler.load_state_dict(sched)
    inner = model.module if hasattr(model, "module") else model
    inner.k = inner.steps_before_epoch(args.start_epoch)        # PixPro.py:295, the constructor's own expression
    log(f"=> resumed '{args.resume}' (epoch {ckpt['epoch']})")


def load_pretrained(model, path):
    ckpt = _safe_load(path)
    sd = model.state_dict()
    sd.update(ckpt['model'])
    model.load_state_dict(sd)


def build_flow_model(args, device):
    """main_pretrain.py:42-58: RAFT for the online flow mode (`--use_flow` without `--use_flow_file`).  With
    --synthetic and no checkpoint a randomly initialised RAFT-small stands in (there are no weights offline)."""
    if not args.use_flow or args.use_flow_file:
        return None
    if args.flow_model:
        return util.load_flow_model(args, device)
    if not args.synthetic:
        raise FileNotFoundError(f"not exit flow model path {args.flow_model}")
    from contrast.flow import RAFT
    args.small, args.mixed_precision = True, False
    log("no --flow_model given: randomly initialised RAFT-small (synthetic run)")
    model = RAFT(args).to(device).eval()
    for p in model.parameters():
        p.requires_grad = False
    return model


def prepare_step_inputs(data, args, flow_model=None):
    """main_pretrain.py:224-251: flow preparation and the coord/flow packing the model expects."""
    coord1, coord2 = data[2], data[3]
    info = {}
    if args.use_flow:
        flow_fwd, flow_bwd = util.apply_optical_flow(data, flow_model, args)
        info["mask_fwd"], info["mask_bwd"] = flow_fwd[2], flow_bwd[2]
        info["n_frames"] = data[6][1]
        coord1, coord2 = [coord1, flow_fwd], [coord2, flow_bwd]
    return coord1, coord2, info


def train_step(model, optimizer, scheduler, data, args, flow_model=None):
    """One pre-training step (main_pretrain.py:213-269) with no host synchronisation."""
    coord1, coord2, info = prepare_step_inputs(data, args, flow_model)
    loss, pos_num_list = model(data[0], data[1], coord1, coord2)
    optimizer.zero_grad()
    loss.backward()
    wgrad_stream.join()          # weight gradients computed on the side stream (--wgrad-stream); nothing to wait for otherwise
    optimizer.step()
    scheduler.step()
    return loss, pos_num_list, info


def train(epoch, loader, model, optimizer, scheduler, args, flow_model=None):
    model.train()
    loss_meter, time_meter = util.AverageMeter(), util.AverageMeter()
    is_mask_flow = args.use_flow and args.alpha1 is not None and args.alpha2 is not None
    end = time.time()
    for idx, data in enumerate(loader):
        loss, pos_num_list, info = train_step(model, optimizer, scheduler, data, args, flow_model)
        if idx % args.print_freq == 0:
            # all .item() syncs of the reference's per-step logging (main_pretrain.py:272-285) live here
            (pn1, pm1), (pn2, pm2) = pos_num_list
            pos_num = pn1.sum().item() + pn2.sum().item()
            pos_mean = (pm1.mean().item() + pm2.mean().item()) / 2.0
            loss_meter.update(loss.item(), data[0].size(0))
            extra = ""
            if is_mask_flow:
                r_f = util.calc_mask_ratio(info["mask_fwd"]).mean().item()
                r_b = util.calc_mask_ratio(info["mask_bwd"]).mean().item()
                extra += f" mask ratio {(r_f + r_b) / 2:07.3%}"
            if args.use_flow:
                mean_f, no_r, _ = util.calc_frame_ratio(info["n_frames"])
                extra += f" Mean frames {mean_f:.3f} ({no_r:07.3%})"
            if torch.cuda.is_available():
                extra += f" max_mem: {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}M"
            time_meter.update(time.time() - end)
            log(f"Train: [{epoch}/{args.epochs}][{idx}/{len(loader)}]  Time {time_meter.val:.3f}  "
                f"lr {optimizer.param_groups[0]['lr']:.3f}  loss {loss_meter.val:.3f} ({loss_meter.avg:.3f}) "
                f"[{loss_meter.val + 4.0:.3f}] pos_num {pos_num:.4g} ({pos_mean:07.3%}){extra}")
        end = time.time()


def main(args, device):
    if not args.synthetic:
        raise NotImplementedError("the BDD100k video loader is outside this build; run with --synthetic")
    size = args.image_size if isinstance(args.image_size, int) else args.image_size[0]
    world = dist.get_world_size() if dist.is_initialized() else 1
    # the EMA schedule (PixPro.py:294) counts num_instances / (world * batch) steps per epoch and the LR schedule counts
    # len(loader): with the synthetic loader the epoch length is --steps-per-epoch, so make the two agree
    args.num_instances = args.steps_per_epoch * args.batch_size * world
    loader = SyntheticLoader(args.steps_per_epoch, args.batch_size, image_size=size, n_frames=max(args.n_frames, 2),
                             frame_hw=tuple(args.frame_size), device=device, flow_up=args.flow_up,
                             with_flow=args.use_flow, with_frames=args.use_flow and not args.use_flow_file, channels_last=getattr(args, "channels_last", False), seed=1234 + (dist.get_rank() if dist.is_initialized() else 0))
    model, optimizer = build_model(args, device)
    flow_model = build_flow_model(args, device)
    scheduler = get_scheduler(optimizer, len(loader), args)
    if args.pretrained_model:
        load_pretrained(model, args.pretrained_model)
    if args.auto_resume:
        cand = os.path.join(args.output_dir, "current.pth")
        if os.path.exists(cand):
            args.resume = cand
    if args.resume:
        load_checkpoint(args, model, optimizer, scheduler)
    for epoch in range(args.start_epoch, args.epochs + 1):
        train(epoch, loader, model, optimizer, scheduler, args, flow_model)
        if epoch == args.start_epoch:
            from contrast.miopen_tuning import status
            log(f"MIOpen: {status()}")
        rank0 = not dist.is_initialized() or dist.get_rank() == 0
        if rank0 and (epoch % args.save_freq == 0 or epoch == args.epochs):
            save_checkpoint(args, epoch, model, optimizer, scheduler)
        if epoch >= args.debug_epochs:
            break


def main_prog(opt):
    os.makedirs(opt.output_dir, exist_ok=True)
    if not dist.is_initialized() or dist.get_rank() == 0:
        with open(os.path.join(opt.output_dir, "config.json"), "w") as f:
            json.dump({k: v for k, v in vars(opt).items()}, f, indent=2, default=str)
    device = torch.device("cuda", opt.local_rank)
    from contrast.miopen_tuning import use_shipped_db
    use_shipped_db()    # MIOpen's recorded find-db for the ResNet-50 shapes; other shapes are tuned by benchmark mode
    torch.backends.cudnn.benchmark = not opt.no_benchmark
    main(opt, device)


if __name__ == '__main__':
    opt = parse_option(stage='pre-train')
    torch.cuda.set_device(opt.local_rank)
    opt.num_instances = opt.num_instances
    if "RANK" in os.environ:
        dist.init_process_group(backend='nccl', init_method='env://')   # RCCL on ROCm
    main_prog(opt)
    if dist.is_initialized():
        dist.destroy_process_group()
