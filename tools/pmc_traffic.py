#!/usr/bin/env python3
"""Aggregate rocprofv3 PMC passes into the HBM-traffic figures bench.py reports.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out prefix> [tag]
                      [--batch B --horizon H --state-dim D --kind ddpm|ddim --no-attention]

The two passes are separate runs of `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` over
`bench.py --steps 3 --warmup 1 --no-cpu-baseline` (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and, on gfx950, FETCH_SIZE tallies a wide
coalesced read at half its size (MI355X_MICROARCH.md, HBM / rocprofv3 section).
Writes <prefix>_hbm_traffic_pmc[_tag].json (per-kernel averages) and <prefix>_roofline_traffic[_tag].json (the conv3x3
class + the whole step); bench.py finds the latter by the geometry recorded in its "config"."""
import argparse
import csv
import json


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
    return (k.startswith("conv3x3_wide_kernel") or k.startswith("conv_gemm_kernel<true") or k.startswith("conv_skinny_kernel") or
            k.startswith("conv_reg64_kernel"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch"); ap.add_argument("write"); ap.add_argument("prefix"); ap.add_argument("tag", nargs="?", default="")
    ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--state-dim", type=int, default=3); ap.add_argument("--kind", default="ddpm")
    ap.add_argument("--no-attention", action="store_true")
    a = ap.parse_args()
    fetch = per_kernel(a.fetch, "FETCH_SIZE")
    write = per_kernel(a.write, "WRITE_SIZE")
    sfx = ("_" + a.tag) if a.tag else ""
    json.dump({"fetch": fetch, "write": write}, open(a.prefix + "_hbm_traffic_pmc" + sfx + ".json", "w"), indent=1)
    n = tot = 0.0
    all_bytes = 0.0
    for k, (cnt, avg) in fetch.items():
        if k in write:
            all_bytes += cnt * (2.0 * avg + write[k][1]) * 1024.0
        if is_conv3x3(k) and k in write:
            n += cnt
            tot += cnt * (2.0 * avg + write[k][1]) * 1024.0
    steps = max(fetch.get("out_step_kernel", [1, 0])[0], 1)        # one out_step_kernel launch per denoise step
    outp = a.prefix + "_roofline_traffic" + sfx + ".json"
    json.dump({
        "kernel_class": "conv3x3_wide_kernel<...> + conv_reg64_kernel<...> + conv_gemm_kernel<HALO=true,...> + conv_skinny_kernel<...> (all 3x3/3x1 implicit-GEMM launches of a denoise step)",
        "traffic_bytes_per_launch": tot / max(n, 1), "launches_profiled": int(n),
        "hbm_bytes_per_step": all_bytes / steps, "steps_profiled": int(steps),
        "config": {"batch": a.batch, "horizon": a.horizon, "state_dim": a.state_dim, "kind": a.kind, "attention": not a.no_attention},
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1`; "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of a wide coalesced read, "
                  "MI355X_MICROARCH.md HBM section); aggregated by tools/pmc_traffic.py; raw per-kernel averages in "
                  + (a.prefix.split("/")[-1] + "_hbm_traffic_pmc" + sfx + ".json")}, open(outp, "w"), indent=1)
    print(open(outp).read())


if __name__ == "__main__":
    main()
