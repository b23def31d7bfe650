"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-kernel-class HBM bytes per launch and per
STEADY-STATE step.  A step is delimited by its softmax_ce kernel: the dispatches between the last two of them are the backward
of one step and the forward of the next -- exactly one recurring step, with none of the first step's one-off fills and
allocations in it.  Every recurring kernel is counted, the torch-side fills included (class `fill`); whatever still lands in
`other` must stay under 1 % of the bytes or the tool fails.
usage: traffic_agg.py <dir with the two passes> [source-tree hash]"""
import collections
import csv
import glob
import json
import sys

CLASSES = [  # substring of the kernel name -> bench.py stage name (first match wins)
    # the fused stem tail (sfk_bn_maxpool_*): bench.py books it under the BatchNorm stages it replaces
    ("bn_pool_bwd_kernelIDF16bLb0", "bn_bwd_reduce"), ("bn_pool_bwd_kernelIfLb0", "bn_bwd_reduce"), ("bn_pool_bwd_kernel", "bn_bwd_apply"),
    ("E, true>((anonymous namespace)::FM, ELi, unsigned char*", "bn_apply"),      # maxpool_fwd_kernel<..., BN = true>
    ("conv_igemm", "conv_igemm"), ("conv_halo", "conv_igemm"), ("conv_pw_fused", "conv_igemm"), ("pw_dual", "conv_igemm"),
    ("conv_wgrad", "conv_wgrad"), ("wgrad_reduce", "conv_wgrad"), ("band_reduce", "conv_wgrad"),
    ("bn_tail", "bn_finalize"), ("stem_fwd", "stem_fwd"), ("stem_wgrad", "stem_wgrad"),
    ("bn_bwd_reduce", "bn_bwd_reduce"), ("bn_bwd_apply", "bn_bwd_apply"), ("bn_apply", "bn_apply"),
    ("bn_fold", "bn_finalize"), ("bn_finalize", "bn_finalize"), ("bn_bwd_finalize", "bn_finalize"), ("relu_bits", "bn_bwd_reduce"),
    ("maxpool", "pool"), ("head_pool", "pool"), ("fc_", "pool"), ("softmax_ce", "pool"),
    ("adam", "adam"), ("filter_", "filter_refresh"), ("cast_kernel", "filter_refresh"),
    ("fill_zero", "fill"), ("FillFunctor", "fill"), ("at::native", "fill"),   # torch-side fills / counters (seed bump, zero_grad)
]
STEP_MARK = "softmax_ce"


def classify(name: str) -> str:
    for sub, cls in CLASSES:
        if sub in name:
            return cls
    return "other"


def main():
    d = sys.argv[1]
    tree = sys.argv[2] if len(sys.argv) > 2 else None
    byts = collections.defaultdict(lambda: collections.defaultdict(float))   # class -> counter -> KiB in the window
    launches = collections.Counter()
    others = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        marks = sorted({int(r["Dispatch_Id"]) for r in rows if STEP_MARK in r["Kernel_Name"]})
        if len(marks) < 2:
            sys.exit(f"{f}: fewer than two {STEP_MARK} dispatches -- profile at least two steady-state steps")
        lo, hi = marks[-2], marks[-1]
        seen = set()
        for r in rows:
            i = int(r["Dispatch_Id"])
            if not (lo < i <= hi):
                continue
            cls = classify(r["Kernel_Name"])
            byts[cls][r["Counter_Name"]] += float(r["Counter_Value"])
            if cls == "other":
                others[r["Kernel_Name"][:80]] += 1
            if r["Counter_Name"] == "FETCH_SIZE" and i not in seen:
                seen.add(i)
                launches[cls] += 1
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python bench.py",
           "correction": "bytes = 2 * FETCH_SIZE * 1024 (gfx950 counts a 128-B read request as 64 B) + WRITE_SIZE * 1024",
           "window": "the dispatches between the last two softmax_ce kernels = one steady-state step (no first-step one-offs)",
           "tree": tree, "classes": {}}
    total = 0.0
    for cls, c in sorted(byts.items()):
        n = max(launches[cls], 1)
        rd, wr = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0, c.get("WRITE_SIZE", 0.0) * 1024.0
        total += rd + wr
        out["classes"][cls] = {"launches": launches[cls], "launches_per_step": launches[cls],
                               "read_bytes_per_launch": round(rd / n), "write_bytes_per_launch": round(wr / n),
                               "hbm_bytes_per_launch": round((rd + wr) / n), "hbm_bytes_per_step": round(rd + wr)}
    out["hbm_bytes_per_step"] = round(total)
    oth = out["classes"].get("other", {}).get("hbm_bytes_per_step", 0)
    out["other_frac"] = round(oth / max(total, 1.0), 5)
    if others:
        out["other_kernels"] = dict(others.most_common(12))
    json.dump(out, sys.stdout, indent=1)
    if oth > 0.01 * total:
        sys.exit(f"recurring kernels outside every class move {oth / total:.1%} of the step's bytes: extend CLASSES ({list(others)[:6]})")


if __name__ == "__main__":
    main()
