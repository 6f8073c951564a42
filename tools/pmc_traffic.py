#!/usr/bin/env python
"""Turn the per-kernel PMC summaries of tools/pmc_kernel.sh into the kept traffic file bench.py reads
(profiles/rNN_pmc_traffic.json).  Usage: tools/pmc_traffic.py OUT.json N FLOW_DIR PPM_DIR [PREVIOUS.json]

Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of 16 B/lane coalesced streaming reads -> doubled
for the single-launch PPM kernels (16 B/lane streams, through registers or LDS-DMA); the flow kernels read only the 1/8-resolution maps (4-byte accesses of a
few KB per tile, uncalibrated width), reported uncorrected; WRITE_SIZE is exact for both.  Counters are KiB per dispatch
(median over dispatches), collected in separate --pmc passes.  batch_norm entries are carried over from PREVIOUS (those
kernels did not change)."""
import json
import re
import sys


def read(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+median\s+([\d.]+)", line)
            if m:
                out[cur][m.group(1)] = float(m.group(2))
    return out


def main():
    out_path, n, flow_dir, ppm_dir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    flow, ppm = read(f"{flow_dir}/pmc_summary.txt"), read(f"{ppm_dir}/pmc_summary.txt")
    res = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "N": n}
    fk = {k: v for k, v in flow.items() if k.startswith("k_prepare") or k.startswith("k_chain") or k.startswith("k_fbc")}
    fetch = sum(v.get("FETCH_SIZE", 0.0) for v in fk.values())
    write = sum(v.get("WRITE_SIZE", 0.0) for v in fk.values())
    res["flow_prepare"] = {"kernels": sorted(fk), "fetch_kib": fetch, "write_kib": write,
                           "hbm_bytes_per_image": (fetch + write) * 1024 / n,
                           "valu_wave_instructions": sum(v.get("SQ_INSTS_VALU", 0.0) for v in fk.values()),
                           "salu_wave_instructions": sum(v.get("SQ_INSTS_SALU", 0.0) for v in fk.values())}
    pk = {k: v for k, v in ppm.items() if k.startswith("ppm_nhwc")} or {k: v for k, v in ppm.items() if k.startswith("ppm_fused")}
    if pk:
        v = list(pk.values())[0]
        res["ppm_loss"] = {"fetch_kib": v.get("FETCH_SIZE", 0.0), "write_kib": v.get("WRITE_SIZE", 0.0),
                           "hbm_bytes_per_image": (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 / n}
    if len(sys.argv) > 5:
        prev = json.load(open(sys.argv[5]))
        if "batch_norm" in prev:
            res["batch_norm"] = dict(prev["batch_norm"], carried_over_from=sys.argv[5])
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
