"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-kernel-class HBM bytes per launch.
usage: traffic_agg.py <dir with the two passes> <steps profiled>"""
import collections
import csv
import glob
import json
import sys

CLASSES = [  # substring of the kernel name -> bench.py stage name (first match wins)
    # the fused stem tail (sfk_bn_maxpool_*): bench.py books it under the BatchNorm stages it replaces
    ("bn_pool_bwd_kernelIDF16bLb0", "bn_bwd_reduce"), ("bn_pool_bwd_kernelIfLb0", "bn_bwd_reduce"), ("bn_pool_bwd_kernel", "bn_bwd_apply"),
    ("E, true>((anonymous namespace)::FM, ELi, unsigned char*", "bn_apply"),      # maxpool_fwd_kernel<..., BN = true>
    ("conv_igemm", "conv_igemm"), ("conv_pw_fused", "conv_igemm"), ("conv_wgrad", "conv_wgrad"), ("wgrad_reduce", "conv_wgrad"),
    ("bn_tail", "bn_finalize"), ("stem_fwd", "stem_fwd"), ("stem_wgrad", "stem_wgrad"),
    ("bn_bwd_reduce", "bn_bwd_reduce"), ("bn_bwd_apply", "bn_bwd_apply"), ("bn_apply", "bn_apply"),
    ("bn_fold", "bn_finalize"), ("bn_finalize", "bn_finalize"), ("bn_bwd_finalize", "bn_finalize"),
    ("maxpool", "pool"), ("head_pool", "pool"), ("adam", "adam"), ("filter_", "filter_refresh"), ("cast_kernel", "filter_refresh"),
]


def classify(name: str) -> str:
    for sub, cls in CLASSES:
        if sub in name:
            return cls
    return "other"


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    kib = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            cls = classify(r["Kernel_Name"])
            kib[cls][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), r["Counter_Name"])
            if r["Counter_Name"] == "FETCH_SIZE" and key not in seen:
                seen.add(key)
                launches[cls] += 1
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python bench.py",
           "correction": "bytes = 2 * FETCH_SIZE * 1024 (gfx950 counts a 128-B read request as 64 B) + WRITE_SIZE * 1024",
           "steps_profiled": steps, "classes": {}}
    for cls, c in sorted(kib.items()):
        n = max(launches[cls], 1)
        rd, wr = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0, c.get("WRITE_SIZE", 0.0) * 1024.0
        out["classes"][cls] = {"launches": launches[cls], "launches_per_step": round(launches[cls] / steps, 1),
                               "read_bytes_per_launch": round(rd / n), "write_bytes_per_launch": round(wr / n),
                               "hbm_bytes_per_launch": round((rd + wr) / n),
                               "hbm_bytes_per_step": round((rd + wr) / steps)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
