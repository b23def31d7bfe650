#!/usr/bin/env python3
"""Do HIP streams that only meet through event waits run independently between the waits?  A replica of the step's backward
hand-offs with spin kernels, host far ahead of the GPU (run under `rocprofv3 --kernel-trace`; the trace is the result):
   per stage:  lane A: NA kernels | lane B: NB kernels | B waits for A's position after its NA kernels | lane B: NF kernels
A never waits for B, so A's kernels should run back to back across stages.
   python tools/probe/lane_coupling.py <variant>
variants: ab      the A -> B hand-off only
          abc     + after every 2nd kernel of B, lane C waits for B's position and runs one kernel (the filter-gradient hand-off)
          abcd    + the same from A to lane D
          none    no hand-offs at all (control)"""
import sys
import torch

variant = sys.argv[1] if len(sys.argv) > 1 else "ab"
dev = torch.device("cuda", 0)
A = torch.cuda.Stream(priority=-1)
B, C, D = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1, device=dev)
CA, CB, CW = 250000, 150000, 200000
NA, NB, NF, STAGES, STEPS = 8, 8, 3, 4, 5


def spin(stream, cycles):
    with torch.cuda.stream(stream):
        torch.cuda._sleep(cycles)


def hand(src, dst):
    ev = torch.cuda.Event()
    ev.record(src)
    dst.wait_event(ev)


torch.cuda.synchronize()
for step in range(STEPS):
    for si in range(STAGES):
        for k in range(NA):
            spin(A, CA)
            if variant == "abcd" and k % 2 == 1:
                hand(A, D)
                spin(D, CW)
        for k in range(NB):
            spin(B, CB)
            if variant in ("abc", "abcd") and k % 2 == 1:
                hand(B, C)
                spin(C, CW)
        if variant != "none":
            hand(A, B)
        for _ in range(NF):
            spin(B, CB)
    for s in (B, C, D):                          # join, as before the optimiser
        A.wait_stream(s)
    with torch.cuda.stream(A):
        x.add_(1)
torch.cuda.synchronize()
print("done", variant, float(x[0]))
