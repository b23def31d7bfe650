#!/usr/bin/env python
"""Benchmark of BASELINE.json's metric: clips/sec of a full training step (forward, cross-entropy, backward,
gradient all-reduce, Adam) of SlowFast-R50 8x8 on synthetic 3 x 32 x 224^2 bf16 clips, 32 clips per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment); weak scaling (32 clips per GPU); inputs
are resident in HBM before the timed region; rank 0 prints ONE JSON line.  Besides the contract keys the line holds
  roofline      the dominant kernel class (by device time in one step): algorithmic bytes (FLOPs) / summed launch
                durations, measured live with HIP events on each kernel's own launch stream in instrumented steps of the
                PRODUCTION 4-lane schedule right after the timed region (the same thing rocprofv3 --kernel-trace of this
                command sees: profiles/rNN_class_stats.json holds its per-class average durations)
  chip          all HBM bytes of a step (committed PMC run) / this run's step time, vs 8 TB/s and vs the fused ideal
  stages        the same accounting for every kernel class (MFMA TFLOP/s for convs, HBM GB/s for BN / pool stages)
  cpu_baseline  the oracle (torch.nn restatement of the reference path) running the same training step on the host
                cores, on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_BF16_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md (vendor figure without sparsity)
PEAK_HBM_GBS = 8000.0            # HBM3E spec; ~6300 achievable (same guide)
MFMA_KINDS = ("conv_fwd", "conv_dgrad", "conv_wgrad")


FUSED_IDEAL_GB_PER_STEP = 0.791 * 3 * 32   # fully fused ideal 0.791 GB/clip forward, x3 for a training step, 32 clips


def pmc_summary():
    """The newest committed PMC summary (tools/gpu_traffic.sh: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes over
    this very command, gfx950 correction applied) -> (dict, path relative to the repo) or (None, None).  Newest = highest
    (round, version) of profiles/rNN_vM_pmc_traffic.json; the summary records the source-tree hash it was measured on
    (tools/tree_hash.py) and the caller labels it stale when that is not the tree this run comes from."""
    import glob
    import re

    def key(path):
        m = re.search(r"r(\d+)_v(\d+)_pmc_traffic", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=key)
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError):
        return None, None


def pmc_is_stale(summ) -> bool:
    """True when the PMC summary was measured on another source tree than the one this process runs (or does not say)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from tree_hash import tree_hash
        return summ.get("tree") != tree_hash()
    except Exception:
        return True


def pmc_traffic(kind: str):
    """HBM bytes per launch of kernel class `kind` from that summary."""
    c, src = pmc_summary()
    c = (c or {}).get("classes", {}).get(kind)
    if not c or not c.get("launches"):
        return None, None
    return int(c["hbm_bytes_per_launch"]), src


def event_pair_overhead_us(n: int = 32) -> float:
    """what an event pair reads with NOTHING between its records on an idle stream (marker packet processing): subtracted
    from every kernel's reading"""
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        b.record()
    torch.cuda.synchronize()
    xs = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return xs[len(xs) // 2]


def instrumented_step(step, pl, frames, labels, idx, schedule="lanes", only=None, overhead_us=0.0):
    """One step with a HIP event pair around every kernel of the schedule, recorded on the stream the kernel is launched
    on.  schedule = "lanes": the PRODUCTION schedule (4 concurrent lanes, trunk on its high-priority stream) -- a kernel's
    duration includes what it loses to the other lanes' kernels, which is what `rocprofv3 --kernel-trace` of this command
    reports too; "serial": everything on one stream (each kernel alone on the chip; the per-layer tuning view).
    only: instrument just the kernels of these meta kinds (the others run un-timed): with ~500 instead of ~3000 event
    records per step the host stays ahead of the GPU, as it does in the timed region, so no launch bubble leaks into a reading.
    Returns {kind: {'ms', 'flops', 'bytes', 'launches', 'roof_ms'}}."""
    from video_classification_amd.engine import Wait
    eng = step.eng
    ops = step._build(pl, labels)
    ev = []
    per_layer = bool(os.environ.get("SFK_PER_LAYER"))
    lanes = schedule == "lanes" and eng.two_streams

    def timed(oplist):
        streams = eng.lane_streams() if lanes else [torch.cuda.current_stream()]
        handles = [s_.cuda_stream for s_ in streams]
        for op, meta, lane in zip(oplist, oplist.meta, oplist.lane):
            li = lane if lanes else 0
            if isinstance(op, Wait):
                if lanes:
                    e = torch.cuda.Event()
                    e.record(streams[op.on])
                    streams[op.lane].wait_event(e)
                continue
            if (meta is None and not per_layer) or (only is not None and (meta is None or meta["kind"] not in only)):
                op(handles[li])
                continue
            if meta is None:
                meta = {"kind": "misc", "layer": getattr(op, "sfk_name", "?")}   # finalize / pool / head kernels: per-layer dump only
            meta = dict(meta, lane=lane)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(streams[li])
            op(handles[li])
            b.record(streams[li])
            ev.append((meta, a, b))

    def body():
        st = eng._stream()
        eng.drop_seed.add_(1)
        timed(pl.fwd)
        ops["zero_loss"](st); ops["loss"](st); ops["zero_grad"](st)
        timed(pl.bwd)
        ops["adam"](st)

    torch.cuda.synchronize()
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    if lanes and step._trunk is not None:
        cur = torch.cuda.current_stream()
        step._trunk.wait_stream(cur)
        with torch.cuda.stream(step._trunk):
            body()
        cur.wait_stream(step._trunk)
    else:
        body()
    torch.cuda.synchronize()
    out = {}
    if per_layer and only is None:          # (the dominant-class-only pass must not overwrite the full dump)
        rows = [dict(meta, ms=a.elapsed_time(b), t_ms=base.elapsed_time(a)) for meta, a, b in ev]   # t_ms: start since the step began
        path = os.environ["SFK_PER_LAYER"] + ("" if schedule == "serial" else ".lanes")
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "w") as f:
            json.dump(rows, f)
    for meta, a, b in ev:
        kind = meta["kind"]
        if kind == "misc":
            continue
        if kind in ("conv_fwd", "conv_dgrad"):
            kind = "conv_igemm"
        d = out.setdefault(kind, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "roof_ms": 0.0})
        d["ms"] += max(a.elapsed_time(b) - overhead_us * 1e-3, 0.0)
        d["roof_ms"] += 1e3 * max(meta.get("flops", 0.0) / (PEAK_MFMA_BF16_TFLOPS * 1e12),
                                  meta.get("bytes", 0.0) / (PEAK_HBM_GBS * 1e9))
        d["flops"] += meta.get("flops", 0.0)
        d["bytes"] += meta.get("bytes", 0.0)
        d["launches"] += 1
    return out


def cpu_model_string() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline():
    """the oracle's training step (forward, CE, backward, Adam; reference train.py:225-231) at the
    metric geometry, fp32, on the host cores of THIS node, in the same job as the GPU measurement.  Bounded sample: N = 2
    clips per step, 1 warm-up + 2 timed steps; clips/s = 2 N / dt.  The line carries torch.get_num_threads(), os.cpu_count()
    and the CPU model string."""
    from oracle import my_slowfast as o
    torch.manual_seed(1234)
    model = o.canonical_slowfast_8x8(400)
    optim = torch.optim.Adam(model.parameters(), lr=2e-4)
    n, timed = 2, 2
    frames = torch.randn(n, 3, 32, 224, 224)
    labels = torch.randint(0, 400, (n,))
    x = o.pack_pathway(frames)
    o.train_step(model, optim, x, labels)
    t0 = time.time()
    for _ in range(timed):
        o.train_step(model, optim, x, labels)
    dt = time.time() - t0
    return {"value": round(timed * n / dt, 4), "unit": "clips/sec", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model_string(), "os_cpu_count": os.cpu_count(),
            "sample": f"{timed} training steps of {n} clips (3x32x224^2, fp32) after 1 warm-up step, {dt:.1f} s; "
                      f"torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--classes", type=int, default=400)
    ap.add_argument("--graph", action="store_true", help="replay the step as one hipGraph (serialises the two pathway streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="one GPU: run the N>1 backward schedule (segments + comm stream) with a local stand-in for the all-reduce; "
                         "a diagnostic of what the fifth stream costs, the line is marked and is not the metric")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--serial-stages", action="store_true", help="also report every kernel class timed alone on one stream")
    ap.add_argument("--ablate", default="", help="TIMING DIAGNOSTIC: comma-separated kernel classes (stage names, or laneN) the lane "
                    "scheduler skips -- the step's results are garbage and the line says so (tools/gpu_ablate.sh)")
    args = ap.parse_args()

    from video_classification_amd import dist as sdist
    from video_classification_amd.engine import EngineOptions
    from video_classification_amd.slowfast import pack_pathway_index, slowfast_r50_8x8
    from video_classification_amd.train import TrainStep

    rank, world, local = sdist.init_process_group_from_env("nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # experiment switches: read from the environment HERE, explicitly (tools/gpu_ab_env.sh); the engine itself never does
    options = EngineOptions.from_env()
    if args.ablate:
        options.ablate_kinds = frozenset(k for k in args.ablate.split(",") if k)
    model = slowfast_r50_8x8(args.classes, dtype=torch.bfloat16, device=dev, seed=0, options=options)
    model.train()
    eng = model.engine
    gen = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    frames = torch.randn(B, 3, 32, 224, 224, generator=gen).to(torch.bfloat16).to(dev)   # N,C,T,H,W resident in HBM
    labels = torch.randint(0, args.classes, (B,), generator=gen).to(dev)
    idx = pack_pathway_index(32, 4, dev)                   # slow pathway = frames [0,4,8,13,17,22,26,31]
    reducer = sdist.GradReducer(eng.G, bucket_mb=32.0) if world > 1 else None
    if world == 1 and args.rehearse_comm:
        reducer = sdist.LoopbackReducer(eng.G, world=8, bucket_mb=32.0)     # NOT the metric: see --rehearse-comm
    step = TrainStep(eng, lr=2e-4, use_graph=args.graph, reducer=reducer)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2 if step.use_graph else 0)):
        step(frames, frames, labels, slow_t_index=idx)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(frames, frames, labels, slow_t_index=idx)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
    final_loss = float(step.loss[0])

    line = {
        "metric": "clips/sec SlowFast-R50 8x8, 32x224^2 bf16 training step", "value": round(args.steps * B * world / dt, 2),
        "unit": "clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "config/slowfast-Torso.yaml path at the metric geometry: SlowFast-R50 8x8, "
                               "3x32x224^2 clips, fwd+CE+bwd+Adam", "clips_per_gpu": B, "global_batch": B * world,
                   "classes": args.classes, "parallelism": f"dp{world}", "hipgraph": bool(step.use_graph), "pathway_streams": 2 if eng.two_streams else 1,
                   "params": eng.num_parameters()},
        "loss_after": round(final_loss, 4),
    }
    if options.non_default():
        line["engine_options"] = {k_: (sorted(v) if isinstance(v, frozenset) else v) for k_, v in options.non_default().items()}
    if options.ablate_kinds:
        line["WARNING"] = ("TIMING DIAGNOSTIC, NOT THE METRIC: the scheduler skipped the kernel classes " + ",".join(sorted(options.ablate_kinds))
                           + " -- the step computed garbage; only ms_per_step means anything")
    if world == 1 and args.rehearse_comm:
        line["rehearsal"] = ("NOT the metric: the N = 8 backward schedule on one GPU with a local stand-in for every all-reduce "
                             "bucket (dist.LoopbackReducer); GPU_MAX_HW_QUEUES=" + os.environ.get("GPU_MAX_HW_QUEUES", "default"))
    if rank == 0 and not args.no_roofline:
        pl = eng._plan_for(frames, frames, idx, True)
        # the roofline numbers come from the PRODUCTION schedule (4 lanes); SFK_PER_LAYER additionally dumps the serial view
        ovh = event_pair_overhead_us()
        for _ in range(2):
            stages = instrumented_step(step, pl, frames, labels, idx, "lanes", overhead_us=ovh)   # (first pass warms the event pool)
        if os.environ.get("SFK_PER_LAYER") or args.serial_stages:
            serial = instrumented_step(step, pl, frames, labels, idx, "serial")
            line["stages_serial_ms"] = {k_: round(v["ms"], 3) for k_, v in serial.items()}
        rep = {}
        for kind, d in stages.items():
            sec = d["ms"] * 1e-3
            r = {"ms_per_step": round(d["ms"], 3), "launches": d["launches"]}
            if d["flops"] > 0:
                r["tflops"] = round(d["flops"] / sec / 1e12, 1)
            if d["bytes"] > 0:
                r["algorithmic_GBps"] = round(d["bytes"] / sec / 1e9, 1)
            rep[kind] = r
        line["stages"] = rep
        line["stages_schedule"] = "production: 4 concurrent lanes, HIP events on each kernel's own stream (kernels of different lanes overlap, so the class times add up to more than ms_per_step)"
        dom = max(stages, key=lambda k: stages[k]["ms"])
        # the dominant class again, alone under the event pairs (see instrumented_step): these are the roofline's durations
        dom_kinds = ("conv_fwd", "conv_dgrad") if dom == "conv_igemm" else (dom,)
        d = instrumented_step(step, pl, frames, labels, idx, "lanes", only=dom_kinds, overhead_us=ovh)[dom]
        sec = d["ms"] * 1e-3
        # which roof bounds the class as a whole: the larger of its aggregate MFMA time and its aggregate HBM time
        # (the conv classes mix 18 MFMA-bound layers with 93 HBM-bound ones; summed, the bytes dominate)
        t_mfma = d["flops"] / (PEAK_MFMA_BF16_TFLOPS * 1e12)
        t_hbm = d["bytes"] / (PEAK_HBM_GBS * 1e9)
        common = {"avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2), "launches_per_step": d["launches"],
                  "event_pair_overhead_us_subtracted": round(ovh, 2),
                  "algorithmic_bytes_per_launch": int(d["bytes"] / d["launches"]),
                  "algorithmic_flops_per_launch": int(d["flops"] / d["launches"]),
                  "schedule": "production 4-lane step (same as rocprofv3 --kernel-trace of this command: profiles/rNN_class_stats.json)",
                  # sum over launches of max(flops/peak, bytes/peak) / measured time: the per-layer roofline
                  "layerwise_frac": round(d["roof_ms"] / d["ms"], 4),
                  "mfma_frac": round(d["flops"] / sec / 1e12 / PEAK_MFMA_BF16_TFLOPS, 4),
                  "hbm_frac": round(d["bytes"] / sec / 1e9 / PEAK_HBM_GBS, 4)}
        if t_mfma > t_hbm:
            ach = d["flops"] / sec / 1e12
            line["roofline"] = dict({"kernel": dom, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_MFMA_BF16_TFLOPS,
                                     "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_BF16_TFLOPS, 4), "traffic": None}, **common)
        else:
            ach = d["bytes"] / sec / 1e9
            line["roofline"] = dict({"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                                     "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None}, **common)
        line["roofline"]["traffic"], src = pmc_traffic(dom)
        summ, _ = pmc_summary()
        stale = pmc_is_stale(summ) if summ else None
        if src:
            line["roofline"]["traffic_source"] = src + " (rocprofv3 --pmc passes over this command; not re-measured in this run)"
            line["roofline"]["traffic_stale"] = stale      # True: measured on another source tree than this run's
        # chip level: all HBM bytes of a step (PMC) over the step time, against the 8 TB/s spec and the fused-ideal bytes
        summ, src = pmc_summary()
        if summ:
            cls = summ.get("classes", {})
            # EVERY recurring kernel of a steady-state step (tools/traffic_agg.py: the window between two softmax_ce kernels; the
            # torch-side fills are class `fill`, anything unclassified is `other` and fails the tool above 1 % of the bytes).
            # Summaries of rounds 1-3 averaged over the profiled steps instead and carried first-step one-offs in `other`: there
            # `other` stays excluded.
            per_step = sum(c["hbm_bytes_per_step"] for k_, c in cls.items() if k_ != "other" or "window" in summ)
            line["chip"] = {"pmc_hbm_bytes_per_step": int(per_step), "pmc_source": src, "pmc_stale": stale,
                            "hbm_frac_of_step": round(per_step / (dt / args.steps) / (PEAK_HBM_GBS * 1e9), 4),
                            "bytes_vs_fused_ideal": round(per_step / (FUSED_IDEAL_GB_PER_STEP * 1e9), 3),
                            "fused_ideal_GB_per_step": round(FUSED_IDEAL_GB_PER_STEP, 1),
                            "note": "bytes from the committed PMC run of this command, ms from THIS run's timed region"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()          # rank 0's instrumented step is over: every rank leaves together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
