#!/usr/bin/env python3
"""Pipeline.run_pair on C3 with the image and the canvas in page-locked arrays: median wall clock of 40 passes, canvas checked against a
pageable pass; APAP_PIPE_EXP=events adds three passes with HIP events at the stage boundaries (Pipeline.trace).  (Round 4 also ran it
with experimental orderings - solve first, an own main stream - that are no longer in the pipeline: profiles/r04_pipeline_overlap.txt.)"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from cvx_proj_amd.pipeline import Pipeline  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

p = config_pair("C3")
m = p.vertices.shape[0]
pipe = Pipeline()
pipe.trace = "events" in os.environ.get("APAP_PIPE_EXP", "")
img = pipe.pinned_array(p.img.shape)
np.copyto(img, p.img)
canvas = pipe.pinned_array((p.final_h, p.final_w, 3))
args = (p.src, p.dst, p.Hg, p.shape, p.shape, m, p.gamma, p.sigma)
with contextlib.redirect_stdout(io.StringIO()):
    flat0, ref = pipe.run_pair(*args, other_img=p.img)
    ts, tp = [], []
    for i in range(45):
        canvas[:64] = 0
        t0 = time.perf_counter()
        flat, _ = pipe.run_pair(*args, other_img=img, canvas_out=canvas)
        ts.append(time.perf_counter() - t0)
    for i in range(15):
        t0 = time.perf_counter()
        pipe.run_pair(*args, other_img=p.img)
        tp.append(time.perf_counter() - t0)
    if "events" in os.environ.get("APAP_PIPE_EXP", ""):
        for _ in range(3):
            pipe.run_pair(*args, other_img=img, canvas_out=canvas)
            print("   device marks (us from the first enqueue): " + "  ".join(f"{n} {t:.0f}" for n, t in pipe.device_marks), file=sys.stderr)
ts, tp = sorted(ts[5:]), sorted(tp[3:])
print(f"{os.environ.get('APAP_PIPE_EXP', '') or 'default':28s} pinned: median {ts[len(ts) // 2] * 1e3:.3f} ms  min {ts[0] * 1e3:.3f}   pageable: median "
      f"{tp[len(tp) // 2] * 1e3:.3f} ms   same canvas {bool((canvas == ref).all())} same flat {bool((flat == flat0).all())}  {pipe.timeline}")
if "events" in os.environ.get("APAP_PIPE_EXP", ""):
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(2):
            pipe.run_pair(*args, other_img=p.img)
            print("   pageable pass, device marks: " + "  ".join(f"{n} {t:.0f}" for n, t in pipe.device_marks) + f"   host {pipe.timeline}", file=sys.stderr)
