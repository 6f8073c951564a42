#!/usr/bin/env python
"""bench.py --gpus N --steps K --warmup W

Measures BASELINE.json's metric: pre-train images/sec (whole node) of the ResNet-50 PixPro+optical-flow
step at 224x224, nframe=2, fp32 (the flow recipes train at apex O0), synthetic inputs resident in HBM.
A step = file-mode flow preparation (HIP) -> 2 online + 2 momentum encoder passes (PyTorch-ROCm) ->
fused PPM / flow-matched pixel loss forward+backward (HIP) -> backward -> LARS step (HIP) -> LR schedule;
the EMA of the key encoder (HIP) runs inside the forward as in the reference.

For N>1 the driver starts one process per GPU (torch.distributed.run); gradients are all-reduced by DDP over
RCCL/xGMI and BatchNorm statistics by SyncBN.  Per-GPU batch is fixed (weak scaling).  Rank 0 prints one
JSON line; besides the contract keys it carries
  roofline      dominant hot-path kernel: algorithmic bytes / HIP-event duration measured inside the timed steps
  hot_kernels   the same for every hot-path kernel of the step
  cpu_baseline  the CPU oracle (port of the reference's torch-CPU path) timed on this host on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pixpro-with-opticalflow_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12        # B/s, MI355X_MICROARCH.md
MFMA_F32_PEAK = 157.3e12  # FLOP/s, MI355X_MICROARCH.md


def _pmc_file():
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return os.path.relpath(found[-1], ROOT) if found else None


PMC_FILE = _pmc_file()      # the latest round's kept summary


def pmc_traffic(kernel, a):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    passes by tools/pmc_traffic.sh, gfx950 corrections applied as the file's note says).  It is a kept summary, not a
    counter read during this run (`traffic_source` on the JSON line says so); only valid for the configuration it was
    collected on, scaled linearly with the batch.  Returns (bytes | None, source | None)."""
    try:
        with open(os.path.join(ROOT, PMC_FILE or "")) as fh:
            t = json.load(fh)
    except OSError:
        return None, None
    if a.image_size != 224 or a.n_frames != 2 or kernel not in t:
        return None, None
    return int(t[kernel]["hbm_bytes_per_image"] * a.batch), f"{PMC_FILE}, collected at N={t.get('N', 256)}, scaled to N={a.batch}"


def pmc_mfma_util(a):
    """MFMA utilisation of the PPM/loss launches from the committed PMC summaries (rocprofv3's MfmaUtil: matrix-core busy
    cycles over the kernel's cycles x 1024 SIMDs; tools/pmc_kernel.sh).  A kept figure for the configuration it was
    collected on (N = 256 at 7 x 7, N = 64 at 14 x 14), not a counter read during this run.  {kernel: percent}, source."""
    import glob
    import re
    name = "ppm_fused_n256" if a.image_size == 224 else "ppm_tiled_14x14_n64" if a.image_size == 448 else None
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{name}.txt"))) if name else []
    if not found:
        return None, None
    out, cur = {}, None
    for line in open(found[-1]):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+MFMA utilisation \(derived\)\s+([\d.]+) %", line)
            if m and float(m.group(1)) > 0:
                out[cur] = float(m.group(1))
    # only the kernels the library still launches (a kept summary can outlive a kernel's template arguments)
    live = _lib_bytes()
    if live:
        out = {k: v for k, v in out.items() if _mangled_fragment(k) in live}
    return (out or None), os.path.relpath(found[-1], ROOT)


def pmc_attainable(a, measured_ms):
    """The bound the single-launch PPM/loss kernel can reach on this chip, next to the fp32-MFMA roofline it is priced on: on
    gfx950 v_mfma_f32_16x16x4_f32 shares the vector datapath (tools/mfma_valu_overlap.hip: the times of an MFMA wave and a
    VALU wave on one SIMD ADD), so a SIMD needs at least  waves x (MFMA instructions x 32 cycles + other VALU instructions x
    4 cycles)  per item, whatever the schedule.  Instruction counts per wave from the kept PMC summary (SQ_INSTS_VALU,
    SQ_INSTS_VALU_MFMA_F32, SQ_WAVES at N = 256), 2 waves per SIMD (one 8-wave work-group per CU: LDS), 2.4 GHz."""
    import glob
    import re
    if a.image_size != 224:
        return None
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_ppm_fused_n256.txt")))
    if not found:
        return None
    vals, cur = {}, None
    for line in open(found[-1]):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\w+)\s+median\s+([\d.]+)", line)
            if m and cur in ("ppm_nhwc_kernel", "ppm_fused_kernel"):
                vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    for name in ("ppm_nhwc_kernel", "ppm_fused_kernel"):
        v = vals.get(name)
        if v and v.get("SQ_WAVES") and _mangled_fragment(name) in _lib_bytes():
            mf = v["SQ_INSTS_VALU_MFMA_F32"] / v["SQ_WAVES"]
            other = (v["SQ_INSTS_VALU"] - v["SQ_INSTS_VALU_MFMA_F32"]) / v["SQ_WAVES"]
            items_per_wave = (2 * 256 * 8) / v["SQ_WAVES"]          # persistent kernels run several items per wave
            cyc = 2 * (mf * 32 + other * 4) / items_per_wave        # per item and SIMD
            rounds = -(-2 * a.batch // 256)                          # items per CU
            floor_ms = rounds * cyc / 2.4e9 * 1e3
            return {"kernel": name, "mfma_instructions_per_wave_and_item": round(mf / items_per_wave, 1),
                    "other_valu_instructions_per_wave_and_item": round(other / items_per_wave, 1),
                    "simd_cycles_per_item": int(cyc), "items_per_cu": rounds, "floor_ms_per_launch": round(floor_ms, 4),
                    "frac_of_attainable": round(floor_ms / measured_ms, 4), "source": os.path.relpath(found[-1], ROOT),
                    "note": "MFMA issue floor + non-overlappable VALU floor (DESIGN.md 4.1); the fp32-MFMA roofline fraction "
                            "above prices the algorithmic flops only"}
    return None


_LIB_BYTES = None


def _lib_bytes():
    global _LIB_BYTES
    if _LIB_BYTES is None:
        try:
            _LIB_BYTES = open(os.path.join(ROOT, "pixpro-with-opticalflow_amd", "lib", "libpixpro_hip.so"), "rb").read()
        except OSError:
            _LIB_BYTES = b""
    return _LIB_BYTES


def _mangled_fragment(kernel):
    """'k_tl_strip<2, 0, 8>' -> b'10k_tl_stripILi2ELi0ELi8EE' (Itanium mangling of a function template with int arguments)."""
    import re
    m = re.match(r"(\w+)(?:<([^>]*)>)?", kernel)
    name, targs = m.group(1), m.group(2)
    frag = f"{len(name)}{name}"
    if targs:
        frag += "I" + "".join(f"Li{t.strip()}E" for t in targs.split(",")) + "E"
    return frag.encode()


VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4      # wave64 instructions / s: 1024 SIMD16s, one wave64 VALU instruction per 4 cycles, 2.4 GHz


def pmc_valu_instructions(a):
    """Vector-ALU wave-instructions per launch of the dense nframe > 2 flow preparation (k_chain_up8 + the mask kernels) from
    the committed PMC summary (SQ_INSTS_VALU, collected at N = 64, nframe = 6), scaled to the batch.  (count, source)."""
    import glob
    import re
    if a.n_frames != 6:
        return None, None
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_flow_nframe6_n64.txt")))
    if not found:
        return None, None
    total, cur = 0.0, None
    for line in open(found[-1]):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+SQ_INSTS_VALU\s+median\s+([\d.]+)", line)
            if m and cur and (cur.startswith("k_chain") or cur.startswith("k_fbc")):
                total += float(m.group(1))
    return (total * a.batch / 64.0 if total else None), os.path.relpath(found[-1], ROOT)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (SURVEY 8d protocol: >= 50)")
    ap.add_argument("--warmup", type=int, default=20, help="untimed warm-up steps (SURVEY 8d protocol: >= 20)")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE configs[1]: 256 on 1 GPU)")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--n-frames", type=int, default=2)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--lazy-flow", action="store_true",
                    help="skip the dense flow preparation, evaluate flow chain + mask inside the loss kernel "
                         "(same results; the default keeps the reference's dense pipeline, which BASELINE's metric names)")
    ap.add_argument("--no-fused-bn", dest="fused_bn", action="store_false",
                    help="torch's (Sync)BatchNorm + ReLU + add instead of the HIP (Sync)BatchNorm with ReLU / residual add "
                         "folded in (default: fused; same statistics exchange, tests/test_gpu_sync_bn.py)")
    ap.add_argument("--overlap-momentum", choices=("auto", "on", "off"), default="auto",
                    help="key (momentum) branch on a second HIP stream next to the online branch.  auto = on with one rank, off "
                         "with several: every SyncBN collective of a process goes through ProcessGroupNCCL's one internal stream "
                         "in host order, so with >1 rank the two branches would serialise at every layer anyway")
    ap.add_argument("--no-overlap-momentum", dest="overlap_momentum", action="store_const", const="off")
    ap.add_argument("--key-bn-group", action="store_true",
                    help="EXPERIMENTAL, with several ranks: the key encoder's SyncBN collectives on a process group of their own "
                         "(second RCCL communicator); combine with --overlap-momentum on to put the key branch on its second "
                         "stream.  Collectives of two communicators issued from two streams are not ordered against each "
                         "other across ranks and may deadlock; never run over RCCL so far, off by default")
    ap.add_argument("--bn-view-batch", action="store_true", default=None,
                    help="both views of an encoder in one pass with per-view BatchNorm statistics: half the SyncBN collectives "
                         "(162 instead of 324 per step), same statistics and results.  Default: on with several ranks (fewer, "
                         "larger collectives; on one GPU it measures the same: 208.2 against 207.9 ms per step), off with one")
    ap.add_argument("--no-bn-view-batch", dest="bn_view_batch", action="store_false",
                    help="two passes per encoder as in the reference, also with several ranks")
    ap.add_argument("--wgrad-stream", choices=("on", "off"), default="off",
                    help="weight gradients of the online encoder's convolutions on a second HIP stream, overlapping the next "
                         "layers' BatchNorm backward (contrast/wgrad_stream.py).  It wants one use of a weight per pass, so it "
                         "switches --bn-view-batch on with it; one rank only (DistributedDataParallel copies a gradient into "
                         "its bucket the moment autograd delivers it).  Measured neutral on MI355X (214.2 against 214.4 ms "
                         "per step, profiles/NOTES_r04.md): off by default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the two short legs for BASELINE configs[3] (nframe=6, N=64) and configs[4] (448^2, N=64) that the "
                         "default one-GPU run appends under `extra_configs` (same process, same model)")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--extra-warmup", type=int, default=5)
    ap.add_argument("--no-untuned-probe", action="store_true",
                    help="skip the child process that times 3 steps with MIOpen's library heuristics (PIXPRO_NO_MIOPEN_DB=1) "
                         "for `config.miopen_untuned_ms_per_step`")
    ap.add_argument("--untuned-probe", action="store_true", help=argparse.SUPPRESS)   # the child's own mode
    ap.add_argument("--no-syncbn-ab", action="store_true",
                    help="with several ranks: skip the untimed A/B of the SyncBN exchange variants (`rccl.syncbn_ab`)")
    ap.add_argument("--ab-key-bn-group", action="store_true",
                    help="EXPERIMENTAL: add the --key-bn-group variant to the A/B.  Two RCCL communicators driven from two "
                         "streams of one process may deadlock (never run over RCCL yet); off by default")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check without a GPU: start the ranks, form the process group over gloo on the CPU, run the "
                         "barrier / max-over-ranks timing reduction around an empty timed region, print the JSON line and stop")
    ap.add_argument("--cpu-flow-images", type=int, default=4, help="images per call of the CPU flow-preparation leg")
    ap.add_argument("--no-channels-last", dest="channels_last", action="store_false",
                    help="keep NCHW activations (default: channels_last, which MIOpen's fp32 implicit-GEMM kernels prefer)")
    return ap.parse_args()


def note(rank, msg):
    if rank == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def model_args(a, world):
    from contrast import option
    argv = ["--model", "PixPro", "--arch", a.arch, "--synthetic", "--use_flow", "--use_flow_file", "--flow_up",
            "--alpha1", "0.01", "--alpha2", "0.5", "--n-frames", str(a.n_frames), "--optimizer", "lars",
            "--base-lr", "1.0", "--weight-decay", "1e-5", "--amp-opt-level", "O0", "--pixpro-p", "2",
            "--pixpro-transform-layer", "1", "--pixpro-pos-ratio", "0.7", "--batch-size", str(a.batch),
            "--epochs", "2000", "--warmup-epoch", "5", "--head-type", "early_return",
            "--image-size", str(a.image_size), str(a.image_size)]
    if a.channels_last:
        argv.append("--channels-last")
    if a.lazy_flow:
        argv.append("--lazy-flow")
    if a.fused_bn:
        argv.append("--fused-bn")
    if a.bn_view_batch is None:                    # auto: the collectives' count matters only where there are collectives
        a.bn_view_batch = world > 1 and a.fused_bn
    a.wgrad_stream = a.wgrad_stream == "on" and a.fused_bn and world == 1
    if a.wgrad_stream:
        a.bn_view_batch = True
        argv.append("--wgrad-stream")
    if a.bn_view_batch:
        argv.append("--bn-view-batch")
    # auto = on with ONE rank only.  With several ranks the key branch on a second stream is pointless without --key-bn-group
    # and, with it, two communicators driven from two streams may deadlock (never run over RCCL): that combination has to be
    # asked for explicitly (--key-bn-group --overlap-momentum on), it is never switched on implicitly.
    a.overlap_momentum = a.overlap_momentum == "on" or (a.overlap_momentum == "auto" and world == 1)
    if a.key_bn_group:
        argv.append("--key-bn-group")
    if a.overlap_momentum:
        argv.append("--overlap-momentum")
    args = option.parse_option(argv=argv)
    args.num_instances = 70000
    return args


class KernelClock:
    """HIP-event pairs recorded on the launch stream immediately around the C-ABI calls of the hot path
    (contrast._native.call), inside the timed steps."""

    def __init__(self, names):
        self.pairs = {}
        self.names = names

    def install(self, native):
        inner = native.call

        def call(name, *a):
            key = self.names.get(name)
            if key is None:
                return inner(name, *a)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            inner(name, *a)
            e.record()
            self.pairs.setdefault(key, []).append((s, e))
        native.call = call

    def reset(self):
        self.pairs = {}

    def mean_ms(self):
        return {k: sum(s.elapsed_time(e) for s, e in v) / len(v) for k, v in self.pairs.items()}


class BnAudit:
    """One extra, untimed step with a HIP-event pair around every batch-norm C-ABI call and the tensor passes each call
    makes (rows * C * 4 B per pass): achieved HBM rate of the BN + ReLU + residual kernels."""

    PASSES = {   # name -> (index of rows, index of C, function(args) -> tensor passes)
        "pixpro_bn_stats": (1, 2, lambda a: 1),
        "pixpro_bn_apply": (11, 12, lambda a: 2 + (a[3] is not None)),
        "pixpro_bn_backward_reduce": (7, 8, lambda a: 2 + (a[6] != 0 and a[3] is None)),
        "pixpro_bn_backward_apply": (12, 13, lambda a: 3 + (a[11] != 0 and a[3] is None) + (a[15] is not None)),
        # the ReLU bit record (1/32 of a pass) in place of the saved output; the shortcut gradient handed over as (dy, bits)
        "pixpro_bn_apply_bits": (10, 11, lambda a: 2 + (a[3] is not None) + 1 / 32),
        "pixpro_bn_backward_reduce_bits": (5, 6, lambda a: 2 + 1 / 32),
        "pixpro_bn_backward_apply_bits": (10, 11, lambda a: 3 + 1 / 32 + (a[13] is not None)),
        "pixpro_bn_masked_add": (4, 5, lambda a: 2 + (a[0] is not None) + 1 / 32),
        # the stem in one pass (rows = N H W of the full-resolution tensor; the pooled tensors are a quarter of it)
        "pixpro_bn_apply_pool": (lambda a: a[9] * a[10] * a[11], 12, lambda a: 1.25),
        "pixpro_bn_pool_backward_reduce": (lambda a: a[5] * a[6] * a[7], 8, lambda a: 1.25),
        "pixpro_bn_pool_backward_apply": (lambda a: a[10] * a[11] * a[12], 13, lambda a: 2.25),
    }

    def __init__(self):
        self.rec = []

    def install(self, native):
        self.native, self.inner = native, native.call

        def call(name, *a):
            spec = self.PASSES.get(name)
            if spec is None:
                return self.inner(name, *a)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self.inner(name, *a)
            e.record()
            rows = spec[0](a) if callable(spec[0]) else a[spec[0]]
            self.rec.append((name, s, e, rows * a[spec[1]] * 4 * spec[2](a)))
        native.call = call

    def remove(self):
        self.native.call = self.inner

    def summary(self):
        out = {}
        for name, s, e, nbytes in self.rec:
            short = name.replace("pixpro_", "").replace("_bits", "").replace("bn_pool_backward", "bn_backward").replace("apply_pool", "apply")
            o = out.setdefault(short, [0, 0.0, 0])
            o[0] += 1
            o[1] += s.elapsed_time(e)
            o[2] += nbytes
        return out


def cpu_baseline(a, args):
    """The oracle on the host cores, by BASELINE.md section 3's protocol: (ii) file-mode flow preparation for
    `cpu_flow_images` images per call at 720x1280; (i) PPM + loss forward+backward
    for one full batch with per-sample dense flow / mask tensors (what the reference's apply_optical_flow hands the model -
    contiguous [N,2,Hf,Wf] / [N,Hf,Wf], not broadcast views); each 5 warm-up + 20 timed iterations, mean and min.  Reported
    as hot-path images/s from the two means.  (iii), the config-1 step, is cpu_cfg1_step below.)"""
    from oracle import pixpro_oracle as O
    from contrast.synthetic import make_batch
    # the box exposes every host core but a 1-GPU job owns a share of them: use the affinity mask, capped
    threads = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("PIXPRO_CPU_THREADS", "16"))))
    torch.set_num_threads(threads)
    nf = a.cpu_flow_images
    data = make_batch(nf, image_size=8, n_frames=a.n_frames, device="cpu", seed=99)
    _, lo_f, lo_b = data[5]
    size = (720, 1280)
    flow_warm, flow_reps, t_calls = 5, 20, []
    with O.aten_samplers():     # the ATen kernels the reference calls (bit-identical to the explicit restatement)
        for i in range(flow_warm + flow_reps):
            t0 = time.perf_counter()
            ff, fb = O.apply_optical_flow_file(lo_f, lo_b, size, True, 0.01, 0.5)
            if i >= flow_warm:
                t_calls.append((time.perf_counter() - t0) / nf)
    t_flow, t_flow_min = sum(t_calls) / len(t_calls), min(t_calls)
    N, C, G = a.batch, 256, a.image_size // 32
    gen = torch.Generator().manual_seed(5)
    big = make_batch(N, image_size=8, n_frames=2, device="cpu", seed=98, with_flow=False)
    c1, c2 = big[2], big[3]
    rep = (N + nf - 1) // nf        # every sample owns its maps (the nf prepared ones, tiled): N x 7.4 MB per direction
    flow_f = ff[0].repeat(rep, 1, 1, 1)[:N].contiguous()
    flow_b = fb[0].repeat(rep, 1, 1, 1)[:N].contiguous()
    mask_f = ff[2].repeat(rep, 1, 1)[:N].contiguous()
    mask_b = fb[2].repeat(rep, 1, 1)[:N].contiguous()
    xs = [torch.randn(N, C, G, G, generator=gen).requires_grad_(True) for _ in range(4)]
    ks = [torch.randn(N, C, G, G, generator=gen) for _ in range(2)]
    ppm_warm, ppm_reps, t_reps = 5, 20, []
    for i in range(ppm_warm + ppm_reps):
        t0 = time.perf_counter()
        A = [c1, [flow_f, size, mask_f]]
        B = [c2, [flow_b, size, mask_b]]
        with O.aten_samplers():
            l1 = O.ppm_loss(xs[0], xs[1], ks[1], A, B, 2.0, 0.0, 0.7, True)[0]
            l2 = O.ppm_loss(xs[2], xs[3], ks[0], B, A, 2.0, 0.0, 0.7, True)[0]
        (l1 + l2).backward()
        if i >= ppm_warm:
            t_reps.append((time.perf_counter() - t0) / N)
    t_ppm, t_ppm_min = sum(t_reps) / len(t_reps), min(t_reps)
    return {"value": round(1.0 / (t_flow + t_ppm), 2), "unit": "images/s (hot path only: flow prep + PPM/loss fwd+bwd)",
            "cores": threads, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"oracle/pixpro_oracle.py (ATen samplers), torch-CPU fp32, {threads} threads: flow prep nframe={a.n_frames} "
                      f"720x1280 on {nf} images, {flow_warm} warm-up + {flow_reps} timed calls (mean {t_flow * 1e3:.1f}, min "
                      f"{t_flow_min * 1e3:.1f} ms/img) + PPM/loss fwd+bwd on {N} images with per-sample dense maps, {ppm_warm} "
                      f"warm-up + {ppm_reps} timed (mean {t_ppm * 1e3:.3f}, min {t_ppm_min * 1e3:.3f} ms/img); encoders excluded",
            "flow_ms_per_image": round(t_flow * 1e3, 3), "flow_ms_per_image_min": round(t_flow_min * 1e3, 3),
            "ppm_ms_per_image": round(t_ppm * 1e3, 4), "ppm_ms_per_image_min": round(t_ppm_min * 1e3, 4),
            "protocol": {"flow": [flow_warm, flow_reps], "ppm": [ppm_warm, ppm_reps]}}


def cpu_cfg1_step():
    """BASELINE configs[0] on the host: ResNet-18, 64x64 2-frame crops + identity flow on a 320x180 frame, batch 8,
    one process, no GPU.  The product has no CPU route, so - exactly as tests/test_dist_gloo.py does - the two HIP
    entry points the step needs are substituted here by the oracle; everything else is the shipped host code."""
    import importlib
    import types
    from oracle import pixpro_oracle as O
    from contrast import ops, resnet
    pixpro_mod = importlib.import_module("contrast.models.PixPro")

    def oracle_ppm_loss(dirs, p, clamp, pos_ratio, normalize_k=True, want_pred=False, generic=False):
        total, nums, means = 0, [], []
        for d in dirs:
            a_ = [d.coord_q, [d.flow, d.size, d.mask]] if d.flow is not None else d.coord_q
            b_ = [d.coord_k, None] if d.flow is not None else d.coord_k
            with O.aten_samplers():
                loss, _, pn, pm = O.ppm_loss(d.x, d.v, d.k, a_, b_, p, clamp, pos_ratio, normalize_k)
            total = total + loss
            nums.append(pn)
            means.append(pm)
        return total, torch.stack(nums), torch.stack(means), None

    def oracle_ema(pq, pk, m):
        for q, k in zip(pq, pk):
            k.data.mul_(m).add_(q.data, alpha=1. - m)

    saved = (ops.ppm_loss, pixpro_mod.ema_update)
    ops.ppm_loss, pixpro_mod.ema_update = oracle_ppm_loss, oracle_ema
    try:
        margs = types.SimpleNamespace(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7, pixpro_clamp_value=0.0,
                                      pixpro_transform_layer=1, pixpro_ins_loss_weight=0.0, output_dir="/tmp",
                                      num_instances=70000, batch_size=8, epochs=2000, start_epoch=1, feature_dim=256,
                                      head_type="early_return", sync_bn=False)
        torch.manual_seed(0)
        model = pixpro_mod.PixPro(resnet.resnet18, margs).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.03, momentum=0.9, weight_decay=1e-5)
        from contrast.synthetic import make_batch
        d = make_batch(8, image_size=64, n_frames=2, frame_hw=(180, 320), device="cpu", seed=7, with_flow=False)
        zf = torch.zeros(8, 2, 180, 320)
        c1, c2 = [d[2], [zf, (180, 320), None]], [d[3], [zf, (180, 320), None]]
        ts, warm, reps = [], 5, 20           # >= 5 warm-up + >= 20 timed iterations
        for i in range(warm + reps):
            t0 = time.perf_counter()
            loss, _ = model(d[0], d[1], c1, c2)
            opt.zero_grad()
            loss.backward()
            opt.step()
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return 8.0 / (sum(ts) / len(ts))
    finally:
        ops.ppm_loss, pixpro_mod.ema_update = saved


def hot_kernel_lines(a, ms, ms_single=None, ms_lazy=None, bn=None):
    """`hot_kernels` of a JSON line from the HIP-event clocks of the C-ABI calls (KernelClock.mean_ms): algorithmic bytes /
    flops per launch over the measured duration, against the HBM / fp32-MFMA peaks.  `a` carries batch,
    image_size, n_frames, arch, overlap_momentum of the configuration that was timed."""
    P = (a.image_size // 32) ** 2
    C = 256
    A = 720 * 1280 * 4
    F1 = a.n_frames - 1
    # algorithmic bytes per image: dense flow prep B_flow; PPM+loss 10*C*P*4; flops 24*C*P^2
    b_flow = (2 * F1 * 2 * A / 64 + 2 * F1 * 2 * A * 2 + 4 * A + A / 2)
    b_ppm, f_ppm = 10 * C * P * 4, 24 * C * P * P
    kernels = {}
    if "flow_prepare" in ms:
        t = ms["flow_prepare"] * 1e-3
        # SURVEY 8d gives two per-image byte counts for the dense flow preparation: B_flow, what the reference's
        # unfused pipeline moves (up-sampled maps written and read back), and the fused lower bound (read the
        # 1/8-resolution maps once, write flows + masks once).  The launches here ARE fused, so the roofline is priced
        # on the lower bound - what they must move; the B_flow ratio is kept beside it (it exceeds 1 for a fused
        # kernel by construction and says nothing about kernel quality).
        b_fused = (F1 * A / 16 + 4.5 * A)
        kernels["flow_prepare"] = {"bound": "hbm", "achieved": round(b_fused * a.batch / t / 1e9, 1), "peak": HBM_PEAK / 1e9,
                                   "unit": "GB/s", "frac": round(b_fused * a.batch / t / HBM_PEAK, 4),
                                   "traffic": pmc_traffic("flow_prepare", a)[0],
                                   "traffic_source": pmc_traffic("flow_prepare", a)[1],
                                   "ms_per_launch": round(ms["flow_prepare"], 4),
                                   "algorithmic_bytes_per_launch": int(b_fused * a.batch),
                                   "unfused_pipeline_bytes_per_launch": int(b_flow * a.batch),
                                   "ratio_to_unfused_pipeline_at_peak": round(b_flow * a.batch / t / HBM_PEAK, 4),
                                   "note": "algorithmic bytes = SURVEY 8d fused lower bound ((F-1)A/16 + 4.5A per image: "
                                           "low-res maps in, dense flows + masks out); SURVEY's B_flow prices the "
                                           "reference's unfused pipeline and is reported as "
                                           "ratio_to_unfused_pipeline_at_peak; DESIGN.md 4.3"}
        vi, vsrc = pmc_valu_instructions(a)
        if vi:
            # the chain of nframe > 2 is bound by vector-ALU ISSUE, not by HBM (bit-exact tap arithmetic, ~136 instructions
            # per pixel and chain step): price it on that as well, so that the HBM fraction does not stand in for quality
            kernels["flow_prepare"]["valu_issue"] = {
                "bound": "valu_issue", "wave_instructions_per_launch": int(vi), "source": vsrc,
                "peak_wave_instructions_per_s": VALU_ISSUE_PEAK, "achieved": round(vi / t, 1),
                "frac": round(vi / t / VALU_ISSUE_PEAK, 4),
                "note": "SQ_INSTS_VALU of the launch's kernels (kept PMC summary, scaled to the batch) over the measured "
                        "duration, against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction; tools/ubench/valu_rate.hip "
                        "measures 2.8 (mul/add) to 5 (fma) cycles per instruction on this chip, so ~0.8-0.9 is the practical ceiling"}
    if "ppm_loss" in ms:
        t = ms["ppm_loss"] * 1e-3
        util, util_src = pmc_mfma_util(a)
        kernels["ppm_loss"] = {"bound": "mfma", "achieved": round(f_ppm * a.batch / t / 1e12, 3),
                               "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                               "frac": round(f_ppm * a.batch / t / MFMA_F32_PEAK, 4),
                               "traffic": pmc_traffic("ppm_loss", a)[0],
                               "traffic_source": pmc_traffic("ppm_loss", a)[1],
                               "ms_per_launch": round(ms["ppm_loss"], 4),
                               "algorithmic_flops_per_launch": int(f_ppm * a.batch),
                               "hbm_frac": round(b_ppm * a.batch / t / HBM_PEAK, 4),
                               "mfma_util_pmc_percent": util, "mfma_util_source": util_src}
        att = pmc_attainable(a, ms["ppm_loss"])
        if att:
            kernels["ppm_loss"]["attainable"] = att
    for extra, nbytes in (("lars_sgd_step", 7 * 33023552 * 4), ("ema_update", 3 * 32957888 * 4)):
        if extra in ms:
            t = ms[extra] * 1e-3
            kernels[extra] = {"bound": "hbm", "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK / 1e9,
                              "unit": "GB/s", "frac": round(nbytes / t / HBM_PEAK, 4), "traffic": None,
                              "ms_per_launch": round(ms[extra], 4), "algorithmic_bytes_per_launch": nbytes}
            if extra == "ema_update" and a.overlap_momentum and ms_single and extra in ms_single:
                # in the timed steps this launch shares the GPU with the online branch (second stream): the figure
                # that prices the kernel is the single-stream audit step's
                t1 = ms_single[extra] * 1e-3
                kernels[extra].update({"achieved": round(nbytes / t1 / 1e9, 1), "frac": round(nbytes / t1 / HBM_PEAK, 4),
                                       "ms_per_launch": round(ms_single[extra], 4),
                                       "ms_per_launch_overlapped": round(ms[extra], 4),
                                       "note": "achieved/frac from the single-stream audit step; in the timed steps the "
                                               "launch runs on the second stream next to the online branch"})
    if ms_lazy and "ppm_loss" in ms_lazy:
        t = ms_lazy["ppm_loss"] * 1e-3
        kernels["ppm_loss_lazy"] = {"bound": "mfma", "achieved": round(f_ppm * a.batch / t / 1e12, 3),
                                    "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                                    "frac": round(f_ppm * a.batch / t / MFMA_F32_PEAK, 4), "traffic": None,
                                    "ms_per_launch": round(ms_lazy["ppm_loss"], 4),
                                    "algorithmic_flops_per_launch": int(f_ppm * a.batch),
                                    "note": "sparse variant (--lazy-flow), one extra untimed step: no flow_prepare launch, "
                                            "flow chain + consistency mask evaluated at the loss sample points inside "
                                            "this launch; never mixed with the dense figures"}
    if bn:
        n_l = sum(v[0] for v in bn.values())
        t_ms = sum(v[1] for v in bn.values())
        nb = sum(v[2] for v in bn.values())
        kernels["batch_norm"] = {"bound": "hbm", "achieved": round(nb / (t_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9,
                                 "unit": "GB/s", "frac": round(nb / (t_ms * 1e-3) / HBM_PEAK, 4),
                                 "traffic": pmc_traffic("batch_norm", a)[0] if a.arch == "resnet50" else None,
                                 "traffic_source": pmc_traffic("batch_norm", a)[1] if a.arch == "resnet50" else None,
                                 "ms_per_step": round(t_ms, 3), "launch_pairs_per_step": n_l,
                                 "algorithmic_bytes_per_step": int(nb),
                                 "by_call": {k: {"n": v[0], "ms": round(v[1], 3),
                                                 "GB/s": round(v[2] / (v[1] * 1e-3) / 1e9, 1)} for k, v in bn.items()},
                                 "note": "fused (Sync)BatchNorm + ReLU + residual add, one audited step with both branches "
                                         "on one stream; bytes = tensor passes x rows x C x 4 (DESIGN.md 4.7)"}
    return kernels


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a):
    """`python bench.py --gpus N` with no torch.distributed.run around it: start one child process per GPU (the reference's
    launch is one MPI rank per GPU, main_pretrain_mpirun.py:17-49) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, pass rank 0's JSON line through and exit non-zero when any rank fails.  This parent makes no GPU query at
    all (not even a device count: without amdsmi torch counts devices through hipGetDeviceCount, which initialises HIP);
    every rank checks for itself that it has a device of its own and fails the launch otherwise."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PIXPRO_LAUNCHED_BY="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // a.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:      # a dead rank leaves the others waiting in a collective: stop them (exact PIDs)
                        q.terminate()
            time.sleep(0.2)
    except KeyboardInterrupt:
        rc = 130
        for p in procs:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    if rc:
        print(f"[bench] a rank failed (exit code {rc})", file=sys.stderr, flush=True)
    sys.exit(rc)


def dry_launch(a, world, rank):
    """The N>1 control flow of main() with nothing in the timed region (tests/test_bench_launcher.py)."""
    seed = 1234 + rank
    comm = None
    if os.environ.get("PIXPRO_TEST_FAIL_RANK") == str(rank):
        sys.exit(7)
    if world > 1:
        dist.init_process_group(backend="gloo", init_method="env://")
        ones = torch.ones(1)
        dist.all_reduce(ones)
        seeds = [None] * world
        dist.all_gather_object(seeds, seed)
        comm = {"world": dist.get_world_size(), "backend": dist.get_backend(), "allreduce_of_ones": int(ones.item()),
                "seeds": seeds, "launcher": os.environ.get("PIXPRO_LAUNCHED_BY", "torch.distributed.run")}
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "elapsed_s": round(elapsed, 4), "rccl": comm}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def timed_steps(step_fn, steps, world, dev):
    """EXACTLY `steps` calls of step_fn between barrier + synchronize pairs; the MAX over ranks of the wall time."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step_fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed, out


def extra_config_leg(a, name, image_size, n_frames, batch, model, optimizer, scheduler, args, clock, dev, rank):
    """One short leg of another BASELINE config in the same process and on the same model (the encoders are
    size-agnostic): `--extra-warmup` untimed + `--extra-steps` timed steps, the same HIP-event clocks around the hot calls."""
    import copy
    import main_pretrain
    from contrast.synthetic import SyntheticLoader
    b = copy.copy(a)
    b.image_size, b.n_frames, b.batch = image_size, n_frames, batch
    loader = SyntheticLoader(a.extra_steps + a.extra_warmup, batch, pool=2, image_size=image_size, n_frames=n_frames,
                             frame_hw=(720, 1280), device=dev, seed=4321 + rank, channels_last=a.channels_last)
    it = iter(loader)
    for _ in range(a.extra_warmup):
        main_pretrain.train_step(model, optimizer, scheduler, next(it), args)
    torch.cuda.synchronize()
    clock.reset()
    elapsed, _ = timed_steps(lambda: main_pretrain.train_step(model, optimizer, scheduler, next(it), args), a.extra_steps, 1, dev)
    ms = clock.mean_ms()
    # the sparse variant beside the dense one (never mixed): one more untimed step
    clock.reset()
    args.lazy_flow = True
    main_pretrain.train_step(model, optimizer, scheduler, next(iter(loader)), args)
    torch.cuda.synchronize()
    args.lazy_flow = False
    ms_lazy = clock.mean_ms()
    del loader, it
    torch.cuda.empty_cache()
    note(rank, f"extra config {name}: {batch * a.extra_steps / elapsed:.1f} images/s; hot kernels (ms): {ms}")
    return {"workload": f"{a.arch} PixPro+optical-flow pre-train step, {image_size}x{image_size} crops, nframe={n_frames}, "
                        f"per-GPU batch {batch}, otherwise as config.workload",
            "baseline_config": name, "value": round(batch * a.extra_steps / elapsed, 2), "unit": "images/s",
            "ms_per_step": round(elapsed / a.extra_steps * 1e3, 3), "steps": a.extra_steps, "warmup": a.extra_warmup,
            "hot_kernels": hot_kernel_lines(b, ms, None, ms_lazy, None)}


def untuned_probe(a):
    """config.miopen_untuned_ms_per_step: the same step with MIOpen's library heuristics instead of the shipped find-db, in
    a child process (MIOpen reads its db paths once per process): PIXPRO_NO_MIOPEN_DB=1, 2 warm-up + 3 timed steps."""
    import subprocess
    env = dict(os.environ, PIXPRO_NO_MIOPEN_DB="1")
    for k in ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--untuned-probe", "--steps", "3", "--warmup", "2", "--batch", str(a.batch),
           "--image-size", str(a.image_size), "--n-frames", str(a.n_frames), "--arch", a.arch, "--no-cpu-baseline"]
    if not a.channels_last:
        cmd.append("--no-channels-last")
    if not a.fused_bn:
        cmd.append("--no-fused-bn")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)["untuned_ms_per_step"]
    except Exception as exc:      # informative only
        return f"not measured: {exc!r}"


def syncbn_ab(a, model, optimizer, scheduler, args, loader, world, dev, rank):
    """N > 1 only, after the timed region: the SyncBN exchange variants on the same model, 3 warm-up + 5 timed steps each
    (max over ranks), so that the first run on real xGMI says which setting the next one should use.
      default        one collective per layer, view and direction (324 per step), both branches on one stream
      bn_view_batch  both views of an encoder in one pass, per-view statistics: 162 collectives per step
      key_bn_group   (only with --ab-key-bn-group: EXPERIMENTAL, may deadlock over RCCL) the key encoder's collectives on a
                     second communicator, the key branch on its second stream"""
    import main_pretrain
    inner = model.module if hasattr(model, "module") else model
    saved = (inner.view_batch, inner.overlap_momentum)
    out = {}

    def run(name):
        import itertools
        it = itertools.cycle(loader.batches)        # the loader itself only holds the headline's steps
        for _ in range(3):
            main_pretrain.train_step(model, optimizer, scheduler, next(it), args)
        el, _ = timed_steps(lambda: main_pretrain.train_step(model, optimizer, scheduler, next(it), args), 5, world, dev)
        out[name] = round(el / 5 * 1e3, 3)
        note(rank, f"syncbn A/B {name}: {out[name]} ms/step")

    inner.view_batch, inner.overlap_momentum = False, False
    run("default")
    if a.fused_bn:
        inner.view_batch = True
        run("bn_view_batch")
        inner.view_batch = False
    if a.ab_key_bn_group and a.fused_bn:
        from contrast.sync_bn import FusedSyncBatchNorm
        grp = dist.new_group()
        mods = [m for enc in (inner.encoder_k, inner.projector_k) for m in enc.modules() if isinstance(m, FusedSyncBatchNorm)]
        prev = [m.process_group for m in mods]
        for m in mods:
            m.process_group = grp
        inner.overlap_momentum = True
        run("key_bn_group+second_stream")
        for m, g in zip(mods, prev):
            m.process_group = g
    inner.view_batch, inner.overlap_momentum = saved
    out["note"] = ("ms per step, 3 warm-up + 5 timed steps each after the headline region (never part of `value`); the headline "
                   "ran with " + ("bn_view_batch" if a.bn_view_batch else "default")
                   + (" + key_bn_group" if a.key_bn_group else ""))
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or with "
                 f"torch.distributed.run --nproc-per-node N ... --gpus N")
    if a.dry_launch:
        return dry_launch(a, world, rank)
    # PIXPRO_REHEARSAL=1: all ranks share the visible GPU(s) and talk over gloo.  Only for rehearsing the N>1 control
    # flow (DDP buckets, SyncBN exchanges, shard seeds, max-over-ranks timing) on a one-GPU box; never a measurement.
    rehearsal = os.environ.get("PIXPRO_REHEARSAL") == "1"
    if rehearsal:
        local %= max(torch.cuda.device_count(), 1)
    elif torch.cuda.device_count() < world:
        sys.exit(f"[bench] {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("PIXPRO_FORCE_DDP") == "1":
        # RCCL.  PIXPRO_FORCE_DDP=1 exercises the DDP wrapper with a one-rank group on a single GPU.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="gloo" if rehearsal else "nccl", init_method="env://")
        # what the collective library itself saw: a sum of ones over the ranks and every rank's device
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        devs = [None] * dist.get_world_size()
        dist.all_gather_object(devs, f"{os.uname().nodename}:cuda{local}")
        comm = {"world": dist.get_world_size(), "backend": dist.get_backend(), "allreduce_of_ones": int(ones.item()),
                "devices": devs, "launcher": os.environ.get("PIXPRO_LAUNCHED_BY", "torch.distributed.run")}
        if comm["world"] != a.gpus or comm["allreduce_of_ones"] != a.gpus:
            sys.exit(f"[bench] the process group has {comm['world']} ranks (all-reduce of ones = "
                     f"{comm['allreduce_of_ones']}), --gpus says {a.gpus}")
        if not rehearsal and len(set(devs)) != len(devs):
            sys.exit(f"[bench] two ranks share a GPU: {devs}")
    else:
        comm = None
    import main_pretrain
    from contrast import ops, util
    from contrast.lr_scheduler import get_scheduler
    from contrast.synthetic import SyntheticLoader

    args = model_args(a, world)
    # MIOpen immediate mode: this image ships no gfx950 find-db, and an exhaustive find on a fresh box
    # compiles for many minutes; the heuristic picks the prebuilt CK implicit-GEMM kernels at once
    from contrast.miopen_tuning import use_shipped_db, status as miopen_status
    tuned = use_shipped_db()      # MIOpen's own find-db for these shapes, recorded on an MI355X (contrast/miopen_tuning.py)
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("PIXPRO_MIOPEN_FIND", "0")))
    note(rank, f"MIOpen: shipped find-db {'in use' if tuned else 'not used'}, find mode {torch.backends.cudnn.benchmark}")
    torch.manual_seed(1234 + rank)
    loader = SyntheticLoader(a.steps + a.warmup, a.batch, pool=2, image_size=a.image_size, n_frames=a.n_frames,
                             frame_hw=(720, 1280), device=dev, seed=1234 + rank, channels_last=a.channels_last)
    model, optimizer = main_pretrain.build_model(args, dev)
    scheduler = get_scheduler(optimizer, 70000 // (a.batch * world), args)
    model.train()

    from contrast import _native
    clock = KernelClock({"pixpro_flow_prepare": "flow_prepare", "pixpro_flow_prepare_ws": "flow_prepare",
                         "pixpro_ppm_loss": "ppm_loss",
                         "pixpro_lars_sgd_step": "lars_sgd_step", "pixpro_ema_update": "ema_update"})
    clock.install(_native)

    it = iter(loader)
    note(rank, f"model built ({a.arch}, per-GPU batch {a.batch}); warm-up")
    for i in range(a.warmup):
        tw = time.perf_counter()
        loss, _, _ = main_pretrain.train_step(model, optimizer, scheduler, next(it), args)
        torch.cuda.synchronize()
        note(rank, f"warm-up step {i}: {time.perf_counter() - tw:.2f} s")
    torch.cuda.synchronize()
    miopen_note = miopen_status()
    note(rank, f"MIOpen: {miopen_note}")
    clock.reset()
    elapsed, (loss, _, _) = timed_steps(lambda: main_pretrain.train_step(model, optimizer, scheduler, next(it), args),
                                        a.steps, world, dev)
    final_loss = loss.item()
    ms = clock.mean_ms()
    note(rank, f"timed {a.steps} steps in {elapsed:.2f} s; hot kernels (ms): {ms}")
    if a.untuned_probe:           # child of untuned_probe(): only the step time is wanted
        if rank == 0:
            print(json.dumps({"untuned_ms_per_step": round(elapsed / a.steps * 1e3, 3), "miopen": miopen_note}), flush=True)
        return
    ab = None
    if world > 1 and not a.no_syncbn_ab:
        ab = syncbn_ab(a, model, optimizer, scheduler, args, loader, world, dev, rank)
    # Two extra, untimed steps (rank 0, one rank only) for figures the timed region cannot give cleanly:
    #  (1) everything on ONE stream: with the key branch on its second stream an event pair spans shared-GPU time, so the
    #      per-call clocks of the BN kernels and of the EMA launch are only meaningful here (reproduces the rocprof sums);
    #  (2) the sparse ("lazy") flow variant of the loss kernel, reported separately from the dense one.
    bn = ms_single = ms_lazy = None
    if rank == 0 and world == 1:
        inner = model.module if hasattr(model, "module") else model
        saved_overlap, inner.overlap_momentum = inner.overlap_momentum, False
        from contrast import wgrad_stream
        wgrad_stream.set_enabled(False)           # (the audit step: every kernel on the one stream)
        clock.reset()
        audit = BnAudit() if a.fused_bn else None
        if audit:
            audit.install(_native)
        main_pretrain.train_step(model, optimizer, scheduler, next(iter(loader)), args)
        torch.cuda.synchronize()
        if audit:
            audit.remove()
            bn = audit.summary()
        ms_single = clock.mean_ms()
        if not a.lazy_flow:
            clock.reset()
            args.lazy_flow = True
            main_pretrain.train_step(model, optimizer, scheduler, next(iter(loader)), args)
            torch.cuda.synchronize()
            args.lazy_flow = False
            ms_lazy = clock.mean_ms()
        inner.overlap_momentum = saved_overlap
        wgrad_stream.set_enabled(True)
        note(rank, f"single-stream audit step (ms): {ms_single}; lazy-flow step (ms): {ms_lazy}")

    extra = None
    if rank == 0 and world == 1 and not a.no_extra_configs and (a.image_size, a.n_frames) == (224, 2) and a.arch == "resnet50":
        extra = {}
        for name, size_, nf_, nb_ in (("configs[2] per-GPU share: nframe=2, global batch 1024 over 8 GPUs", 224, 2, 128),
                                      ("configs[3] per-GPU share: nframe=6, global batch 512 over 8 GPUs", 224, 6, 64),
                                      ("configs[4] per-GPU share: 448x448 crops (14x14 grid), N=64", 448, 2, 64)):
            key = f"nframe{nf_}_n{nb_}" if (nf_, size_) != (2, 448) else f"crop{size_}_n{nb_}"
            try:
                extra[key] = extra_config_leg(a, name, size_, nf_, nb_, model, optimizer, scheduler, args, clock, dev, rank)
            except Exception as exc:      # never lose the headline line to an extra leg
                extra[key] = {"error": repr(exc)}
                torch.cuda.synchronize()
    untuned = None
    if rank == 0 and world == 1 and not a.no_untuned_probe and not a.no_cpu_baseline:
        note(rank, "timing 3 steps with MIOpen's library heuristics in a child process")
        untuned = untuned_probe(a)
    if rank == 0:
        kernels = hot_kernel_lines(a, ms, ms_single, ms_lazy, bn)
        cand = [k for k in ("flow_prepare", "ppm_loss") if k in kernels]
        dominant = max(cand, key=lambda k: kernels[k]["ms_per_launch"]) if cand else None
        out = {
            "metric": f"pretrain images/sec (whole node), ResNet-50 PixPro+OF {a.image_size}^2, nframe={a.n_frames}",
            "value": round(a.batch * world * a.steps / elapsed, 2), "unit": "images/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.arch} PixPro+optical-flow pre-train step, {a.image_size}x{a.image_size} crops, "
                                   f"nframe={a.n_frames}, 720x1280 frames, file-mode flow (flow_up, alpha1=0.01, alpha2=0.5), "
                                   f"p=2, transform_layer=1, LARS, fp32",
                       "batch_norm": ("fused HIP SyncBN+ReLU" + (", two views per pass" if a.bn_view_batch else ""))
                                     if a.fused_bn else "torch (Sync)BatchNorm",
                       "weight_gradients": ("second HIP stream, both views per encoder pass (contrast/wgrad_stream.py)"
                                            if a.wgrad_stream else "backward stream"),
                       "key_branch": ("second HIP stream" + (", own SyncBN process group" if a.key_bn_group and world > 1 else ""))
                                     if a.overlap_momentum else "same stream",
                       "miopen": f"immediate mode; {miopen_note}",
                       **({"miopen_untuned_ms_per_step": untuned} if untuned is not None else {}),
                       "flow_mode": "lazy (flow + mask evaluated at the loss sample points)" if a.lazy_flow else "dense",
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world,
                       "parallelism": f"dp{world}" if world > 1 else "single", "final_loss": round(final_loss, 5),
                       **({"rehearsal": "gloo, ranks share one GPU: not a measurement"} if rehearsal else {})},
            "roofline": dict(kernel=dominant, **kernels[dominant]) if dominant else None,
            "hot_kernels": kernels,
        }
        if extra:
            out["extra_configs"] = extra
        if comm:
            if ab:
                comm["syncbn_ab"] = ab
            out["rccl"] = comm
        if world == 1 and not a.no_cpu_baseline:
            note(rank, "timing the CPU oracle on a bounded sample")
            cb = cpu_baseline(a, args)
            try:
                cb["cfg1_whole_step_images_per_s"] = round(cpu_cfg1_step(), 2)
                cb["cfg1_note"] = ("BASELINE configs[0]: ResNet-18, 64x64 crops, identity flow, batch 8, whole step on the "
                                   "host cores (hot path = oracle, encoders = torch CPU)")
            except Exception as exc:      # the baseline is informative; never fail the bench line for it
                cb["cfg1_note"] = f"cfg1 CPU step not timed: {exc!r}"
            out["cpu_baseline"] = cb
            hot_ms = ms.get("flow_prepare", 0.0) + ms.get("ppm_loss", 0.0)
            out["hot_path_gpu"] = {"images_per_s": round(a.batch / (hot_ms * 1e-3), 1),
                                   "ms_per_batch_fwd": round(hot_ms, 4),
                                   "note": "flow prep + fused PPM/loss fwd+bwd on the GPU, same unit as cpu_baseline"}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
