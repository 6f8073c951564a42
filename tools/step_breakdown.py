"""Per-kernel breakdown of ONE steady-state pre-training step from a rocprofv3 --kernel-trace CSV
(`*_kernel_trace.csv`): the kernels between the last two k_lars_update launches.
usage: python tools/step_breakdown.py <kernel_trace.csv> [header text]"""
import collections
import csv
import sys

path = sys.argv[1]
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_lars_update" in r[2]]
if len(marks) < 2:
    sys.exit("fewer than two k_lars_update launches in the trace")
step = rows[marks[-2] + 1: marks[-1] + 1]
wall = (step[-1][1] - step[0][0]) / 1e6
agg = collections.OrderedDict()
for s, e, n in step:
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += e - s
total = sum(v[1] for v in agg.values()) / 1e6
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print(f"# wall {wall:.2f} ms, sum of kernel time {total:.2f} ms, {len(step)} launches")
print(f"{'kernel':110s} {'n':>5s} {'ms':>9s} {'%':>6s}")
for n, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:110]:110s} {c:5d} {ns / 1e6:9.3f} {100 * ns / 1e6 / total:6.2f}")
groups = collections.OrderedDict([("convolutions (MIOpen / CK)", ("igemm", "conv", "gemm")),
                                  ("batch norm + ReLU + residual (pixpro::k_stats/k_apply/k_bwd_*)", ("k_stats", "k_apply", "k_bwd_", "k_merge", "BatchNorm")),
                                  ("hot path (flow prep, PPM loss, EMA, LARS)", ("k_prepare", "ppm_fused", "k_ema", "k_lars", "k_chain", "k_fbc", "k_finalize", "k_scale")),
                                  ("MIOpen tensor fills (SubTensorOpWithScalar)", ("SubTensorOp",)),
                                  ("element-wise / other torch kernels", ("",))])
left = dict(agg)
print("# groups:")
for g, keys in groups.items():
    ms = 0.0
    for n in list(left):
        if any(k in n for k in keys):
            ms += left.pop(n)[1] / 1e6
    print(f"#   {g:70s} {ms:8.2f} ms  {100 * ms / total:5.1f} %")
