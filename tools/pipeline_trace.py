#!/usr/bin/env python3
"""Device timeline of Pipeline.run_pair (C3, image and canvas in page-locked arrays).

  cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $ROOT/tools/pipeline_trace.py
  python3 tools/pipeline_trace.py --read $OUT      # kernels and copies of the LAST pass, microseconds from its first activity

Without --read: 6 passes, host clock of each on stderr (a 30 ms pause before the last one separates it in the trace)."""
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read(out):
    rows = []
    for f in glob.glob(os.path.join(out, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel " + r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]))
    for f in glob.glob(os.path.join(out, "**", "*_memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            extra = " ".join(f"{k}={r[k]}" for k in r if k.lower() in ("bytes", "size", "direction", "stream_id"))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + extra))
    rows.sort()
    # the last pass: everything after the longest gap of the second half
    gaps = [(rows[i + 1][0] - rows[i][1], i + 1) for i in range(len(rows) // 2, len(rows) - 1)]
    first = max(gaps)[1] if gaps else 0
    t0 = rows[first][0]
    for s, e, what in rows[first:]:
        print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  {what}")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        return read(sys.argv[2])
    import contextlib
    import io
    import numpy as np
    from cvx_proj_amd.pipeline import Pipeline
    from cvx_proj_amd.synth import config_pair
    p = config_pair("C3")
    m = p.vertices.shape[0]
    pipe = Pipeline()
    img = pipe.pinned_array(p.img.shape)
    np.copyto(img, p.img)
    canvas = pipe.pinned_array((p.final_h, p.final_w, 3))
    args = (p.src, p.dst, p.Hg, p.shape, p.shape, m, p.gamma, p.sigma)
    for i in range(6):
        if i == 5:
            time.sleep(0.03)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            pipe.run_pair(*args, other_img=img, canvas_out=canvas)
        print(f"pass {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms  {pipe.timeline}", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
