"""Per-layer roofline report of one training step (SURVEY.md section 8d / north_star: "achieved HBM GB/s on the BN/pool stages
and MFMA utilisation on the conv stages against chip peak").

usage: per_layer_report.py <per_layer.json> [out.txt]
<per_layer.json> is what `SFK_PER_LAYER=<path> python bench.py` writes: every kernel of the step under a HIP event pair,
SERIAL (one stream, each kernel alone on the chip: the tuning view); <path>.lanes holds the same for the PRODUCTION 4-lane
schedule (a kernel's time then includes what it loses to the other lanes).  Peaks: 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E
(MI355X_MICROARCH.md).  A layer is MFMA-bound when flops / 2.5 PF > bytes / 8 TB/s (the 18 + backward layers of res4 / res5)."""
import json
import os
import sys
from collections import defaultdict

PF, BW = 2.5e15, 8.0e12


def short(name: str) -> str:
    return (name.replace("multipathway_blocks.", "p").replace("res_blocks.", "r").replace("blocks.", "b")
            .replace("branch2.", "").replace("multipathway_fusion.", "fuse."))


def load(path):
    with open(path) as f:
        return json.load(f)


def table(rows, title, out):
    out.append(f"== {title}")
    kinds = defaultdict(lambda: [0.0, 0.0, 0.0, 0.0, 0])
    lanes = defaultdict(float)
    for r in rows:
        k = kinds[r["kind"]]
        f, b = r.get("flops", 0.0), r.get("bytes", 0.0)
        k[0] += r["ms"]; k[1] += f; k[2] += b; k[3] += max(f / PF, b / BW) * 1e3; k[4] += 1
        lanes[r.get("lane", 0)] += r["ms"]
    out.append(f"{'class':16s} {'launches':>8s} {'ms':>8s} {'roof ms':>8s} {'frac':>6s} {'TFLOP/s':>9s} {'/2.5PF':>7s} {'GB/s':>8s} {'/8TB/s':>7s}")
    for name, (ms, f, b, roof, n) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
        tf, gb = f / ms / 1e9 if ms else 0.0, b / ms / 1e6 if ms else 0.0
        out.append(f"{name:16s} {n:8d} {ms:8.3f} {roof:8.3f} {roof / ms if ms else 0:6.2f} {tf:9.1f} {tf / 2500:7.3f} {gb:8.1f} {gb / 8000:7.3f}")
    out.append("lane sums (ms): " + ", ".join(f"lane {k}: {v:.2f}" for k, v in sorted(lanes.items()))
               + f"; all: {sum(lanes.values()):.2f}")
    out.append("")


def main():
    path = sys.argv[1]
    serial = load(path)
    out = [f"per-layer report of one training step ({os.path.basename(path)}); peaks 2.5 PFLOP/s bf16 MFMA, 8 TB/s HBM", ""]
    table(serial, "SERIAL schedule (every kernel alone on the chip)", out)
    lanes = None
    if os.path.exists(path + ".lanes"):
        lanes = load(path + ".lanes")
        table(lanes, "PRODUCTION schedule (4 lanes; a kernel's time includes what it loses to the other lanes)", out)
    lane_ms = {}
    if lanes is not None:                       # same op order in both dumps
        for a, b in zip(serial, lanes):
            if a.get("layer") == b.get("layer") and a["kind"] == b["kind"]:
                lane_ms[id(a)] = b["ms"]
    conv = [r for r in serial if r["kind"] in ("conv_fwd", "conv_dgrad", "conv_wgrad", "stem_fwd", "stem_wgrad")]
    mf = [r for r in conv if r.get("flops", 0) / PF > r.get("bytes", 0) / BW]
    out.append("== MFMA-bound conv layers (flops / 2.5 PF > bytes / 8 TB/s): MFMA utilisation = TFLOP/s / 2500")
    out.append(f"{'kind':11s} {'layer':44s} {'GFLOP':>8s} {'MB':>8s} {'us serial':>10s} {'TFLOP/s':>8s} {'util':>6s} {'us lanes':>9s}")
    for r in mf:
        tf = r["flops"] / r["ms"] / 1e9
        lm = lane_ms.get(id(r))
        out.append(f"{r['kind']:11s} {short(r.get('layer', '')):44s} {r['flops'] / 1e9:8.1f} {r['bytes'] / 1e6:8.1f} {r['ms'] * 1e3:10.1f} "
                   f"{tf:8.1f} {tf / 2500:6.3f} {'' if lm is None else f'{lm * 1e3:9.1f}'}")
    tot = sum(r["ms"] for r in mf)
    out.append(f"total {tot:.3f} ms serial for {sum(r['flops'] for r in mf) / 1e12:.3f} TFLOP = "
               f"{sum(r['flops'] for r in mf) / tot / 1e9:.1f} TFLOP/s = {sum(r['flops'] for r in mf) / tot / 1e9 / 2500:.3f} of peak; "
               f"roof {sum(r['flops'] for r in mf) / PF * 1e3:.3f} ms")
    out.append("")
    out.append("== HBM-bound stages: achieved GB/s / 8000 (BatchNorm / pool / head and the HBM-bound convs), 40 largest by time")
    hb = [r for r in serial if r.get("bytes", 0) > 0 and r not in mf]
    out.append(f"{'kind':14s} {'layer':44s} {'MB':>8s} {'us serial':>10s} {'GB/s':>8s} {'/8TB/s':>7s} {'us lanes':>9s}")
    for r in sorted(hb, key=lambda r: -r["ms"])[:40]:
        gb = r["bytes"] / r["ms"] / 1e6
        lm = lane_ms.get(id(r))
        out.append(f"{r['kind']:14s} {short(r.get('layer', '')):44s} {r['bytes'] / 1e6:8.1f} {r['ms'] * 1e3:10.1f} {gb:8.1f} {gb / 8000:7.3f} "
                   f"{'' if lm is None else f'{lm * 1e3:9.1f}'}")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
