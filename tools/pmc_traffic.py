#!/usr/bin/env python3
"""Aggregate rocprofv3 PMC passes into the HBM-traffic figures bench.py reports.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out prefix>

The two passes are separate runs of `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` over
`bench.py --steps 3 --warmup 1 --no-cpu-baseline` (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and, on gfx950, FETCH_SIZE tallies a wide
coalesced read at half its size (MI355X_MICROARCH.md, HBM / rocprofv3 section).
Writes <prefix>_hbm_traffic_pmc.json (per-kernel averages) and <prefix>_roofline_traffic.json (the conv3x3 class)."""
import csv
import json
import sys


def short(name: str) -> str:
    return name.replace("(anonymous namespace)::", "").replace("spdm::", "").replace("void ", "").split("(")[0]


def per_kernel(path: str, counter: str):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        n, s = acc.get(k, (0, 0.0))
        acc[k] = (n + 1, s + float(r["Counter_Value"]))
    return {k: [n, s / n] for k, (n, s) in acc.items()}


def is_conv3x3(k: str) -> bool:
    return k.startswith("conv3x3_wide_kernel") or k.startswith("conv_gemm_kernel<true")


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    prefix = sys.argv[3]
    json.dump({"fetch": fetch, "write": write}, open(prefix + "_hbm_traffic_pmc.json", "w"), indent=1)
    n = tot = 0.0
    all_bytes = 0.0
    for k, (cnt, avg) in fetch.items():
        if k in write:
            all_bytes += cnt * (2.0 * avg + write[k][1]) * 1024.0
        if is_conv3x3(k) and k in write:
            n += cnt
            tot += cnt * (2.0 * avg + write[k][1]) * 1024.0
    steps = max(fetch.get("out_step_kernel", [1, 0])[0], 1)        # one out_step_kernel launch per denoise step
    json.dump({
        "kernel_class": "conv3x3_wide_kernel<...> + conv_gemm_kernel<HALO=true,...> (all 3x3/3x1 implicit-GEMM launches of a denoise step)",
        "traffic_bytes_per_launch": tot / max(n, 1), "launches_profiled": int(n),
        "hbm_bytes_per_step": all_bytes / steps, "steps_profiled": int(steps),
        "config": {"batch": int(sys.argv[4]) if len(sys.argv) > 4 else 4096, "horizon": 32, "state_dim": 3, "kind": "ddpm", "attention": True},
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1`; "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of a wide coalesced read, "
                  "MI355X_MICROARCH.md HBM section); aggregated by tools/pmc_traffic.py; raw per-kernel averages in "
                  + prefix.split("/")[-1] + "_hbm_traffic_pmc.json"}, open(prefix + "_roofline_traffic.json", "w"), indent=1)
    print(open(prefix + "_roofline_traffic.json").read())


if __name__ == "__main__":
    main()
