#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to a per-kernel, per-launch summary (JSON + markdown table).

usage: tools/pmc_summary.py OUT.json DIR [DIR ...]     (each DIR = one rocprofv3 -d output of a separate --pmc pass)

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B.  Per MI355X_MICROARCH.md (HBM section)
gfx950's FETCH_SIZE tallies 128-B requests at 64 B, so `fetch_bytes_corrected` = 2 x FETCH_SIZE (an upper bound for
the narrow gathers, exact for wide streaming reads); WRITE_SIZE is exact for 16-B stores and float atomics.
"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict


def kernel_source_sha(root=None):
    """sha256 over the CODE of the rasterizer's kernel sources (comments and white space stripped): bench.py compares it with the tree it runs from and marks counter figures taken from another state of the kernels
    as stale."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import re
    h = hashlib.sha256()
    # the rasterizer's sources: the counter figures bench.py reads are those of its tile backward (the headline's dominant kernel);
    # the neural / loss / optimizer / densify kernels are not on that path
    names = ("preprocess.hip", "binning.hip", "render.hip", "capi.hip", "gs_layout.h", "kernels.h", "project_gaussian.h", "sh_color.h")
    for f in [os.path.join(root, "segs-slam_amd", "csrc", n) for n in names]:
        h.update(os.path.basename(f).encode())
        # code only: comments and white space do not make a counter file stale
        text = open(f, "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            for row in csv.DictReader(open(f)):
                k = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[k] += float(row["Counter_Value"])  # rows may be split per XCD/instance
                names[row["Dispatch_Id"]] = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            # The first step of a bench run is the calibrating one (reference-shaped forward with its full instance lists):
            # everything up to and including the first tile backward is left out, the averages describe resident steps.
            first_bwd = [int(d) for d, n in names.items() if n.endswith("render_bwd_kernel")]
            cut = min(first_bwd) if (first_bwd and os.environ.get("PMC_KEEP_FIRST_STEP") is None) else -1
            for (disp, cname), v in per_dispatch.items():
                if int(disp) <= cut:
                    continue
                a = acc[names[disp]][cname]
                a[0] += v
                a[1] += 1
    res = {}
    for kern, cs in sorted(acc.items()):
        e = {"launches": max(n for _, n in cs.values())}
        for cname, (tot, n) in cs.items():
            e[cname] = tot / n
        if "FETCH_SIZE" in e:
            e["fetch_bytes_corrected"] = 2.0 * e["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024.0
        if "fetch_bytes_corrected" in e and "write_bytes" in e:
            e["hbm_bytes"] = e["fetch_bytes_corrected"] + e["write_bytes"]
        res[kern] = e
    res["_kernel_source_sha"] = kernel_source_sha()
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("| kernel | launches | fetch MB (2x corrected) | write MB | other |")
    print("|---|---|---|---|---|")
    for k, e in res.items():
        if not isinstance(e, dict):
            continue
        other = {c: round(v, 1) for c, v in e.items() if c not in ("launches", "FETCH_SIZE", "WRITE_SIZE", "fetch_bytes_corrected", "write_bytes", "hbm_bytes")}
        print(f"| {k} | {e['launches']} | {e.get('fetch_bytes_corrected', 0)/1e6:.2f} | {e.get('write_bytes', 0)/1e6:.2f} | {other} |")


if __name__ == "__main__":
    main()
