#!/usr/bin/env python3
"""Does a small page-locked upload on one torch stream wait for a 25 MB page-locked upload on another?  (tools/copy_overlap.hip: plain
HIP streams do overlap.)  Events, microseconds from the first enqueue."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
big_h = torch.empty(24883200, dtype=torch.uint8, pin_memory=True)
small_h = torch.empty(512000, dtype=torch.uint8, pin_memory=True)
small_pageable = torch.empty(512000, dtype=torch.uint8)
big_d = torch.empty_like(big_h, device=dev)
small_d = torch.empty_like(small_h, device=dev)
x = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()


def run(tag, main, side, small, pause_us=0, n=5):
    for i in range(n):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        e[0].record(side)
        with torch.cuda.stream(side):
            big_d.copy_(big_h, non_blocking=True)
            e[1].record(side)
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e6 < pause_us:
            pass
        with torch.cuda.stream(main):
            e[2].record(main)
            small_d.copy_(small, non_blocking=True)
            e[3].record(main)
            x.add_(1.0)
            e[4].record(main)
        torch.cuda.synchronize()
        if i >= n - 2:
            print(f"{tag:70s} big copy ends {e[0].elapsed_time(e[1]) * 1e3:6.0f}   small copy {e[0].elapsed_time(e[2]) * 1e3:6.0f} .. "
                  f"{e[0].elapsed_time(e[3]) * 1e3:6.0f}   kernel after it ends {e[0].elapsed_time(e[4]) * 1e3:6.0f}")


default = torch.cuda.default_stream(dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for pause in (0, 100):
    run(f"main = default stream, small page-locked, host pause {pause} us", default, s1, small_h, pause)
    run(f"main = pool stream,    small page-locked, host pause {pause} us", s2, s1, small_h, pause)
    run(f"main = default stream, small pageable,    host pause {pause} us", default, s1, small_pageable, pause)
    run(f"main = pool stream,    small pageable,    host pause {pause} us", s2, s1, small_pageable, pause)
print("streams:", default, s1, s2, file=sys.stderr)

# ---- consecutive passes of the pipeline's shape: big upload on `side`, small upload + kernel on `main`, big download on `down` ----
down_h = torch.empty(26965161, dtype=torch.uint8, pin_memory=True)
down_d = torch.empty_like(down_h, device=dev)


def passes(tag, main, side, down, n=6):
    out = []
    for i in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        e[0].record(side)
        with torch.cuda.stream(side):
            big_d.copy_(big_h, non_blocking=True)
            e[1].record(side)
        with torch.cuda.stream(main):
            e[2].record(main)
            small_d.copy_(small_h, non_blocking=True)
            e[3].record(main)
            x.add_(1.0)
            main.wait_event(e[1])
            e[4].record(main)
        with torch.cuda.stream(down):
            down.wait_event(e[4])
            e[5].record(down)
            down_h.copy_(down_d, non_blocking=True)
            e[6].record(down)
        down.synchronize()
        main.synchronize()
        out.append(f"{e[0].elapsed_time(e[3]) * 1e3:.0f}/{e[0].elapsed_time(e[1]) * 1e3:.0f}/{e[0].elapsed_time(e[6]) * 1e3:.0f}")
    print(f"{tag:60s} small copy done / big upload done / download done, per pass: " + "  ".join(out))


s3 = torch.cuda.Stream(dev)
passes("main = down = default, side = pool stream", default, s1, default)
passes("main = default, side = pool stream, down = another pool stream", default, s1, s3)
passes("main = down = pool stream s2, side = pool stream s1", s2, s1, s2)
passes("main = default, side = down = pool stream s1", default, s1, s1)
ext = [torch.cuda.ExternalStream(torch.cuda.Stream(dev, priority=-1).cuda_stream) for _ in range(2)]
passes("main = default, side = high-priority pool stream", default, ext[0], default)

# ---- which operation of a pipeline pass makes the NEXT pass's small upload wait?  add them one at a time ----
flat_d = torch.zeros(360000, dtype=torch.float64, device=dev)
import numpy as np  # noqa: E402


def passes2(tag, flat_on_side=False, flat_pageable=True, side_waits_main=False, n=5):
    main, side = default, s1
    out = []
    flat_pin = torch.empty(360000, dtype=torch.float64, pin_memory=True)
    for i in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        e[0].record(side)
        with torch.cuda.stream(side):
            big_d.copy_(big_h, non_blocking=True)
            e[1].record(side)
        e[2].record(main)
        small_d.copy_(small_h, non_blocking=True)
        e[3].record(main)
        x.add_(1.0)
        ready = torch.cuda.Event()
        ready.record(main)
        main.wait_event(e[1])
        x.add_(1.0)
        e[4].record(main)
        if flat_on_side:
            with torch.cuda.stream(side):
                if side_waits_main:
                    side.wait_event(ready)
                if flat_pageable:
                    a = np.empty(360000, np.float64)
                    torch.from_numpy(a).copy_(flat_d)
                else:
                    flat_pin.copy_(flat_d, non_blocking=True)
        down_h.copy_(down_d, non_blocking=True)
        e[6].record(main)
        main.synchronize()
        side.synchronize()
        out.append(f"{e[0].elapsed_time(e[3]) * 1e3:.0f}/{e[0].elapsed_time(e[1]) * 1e3:.0f}/{e[0].elapsed_time(e[6]) * 1e3:.0f}")
    print(f"{tag:76s} " + "  ".join(out))


print("small copy done / big upload done / download done, per pass (main = default stream, side = pool stream):")
passes2("nothing else")
passes2("+ page-locked 2.9 MB download on side, no wait", True, False, False)
passes2("+ page-locked 2.9 MB download on side after side.wait_event(main's event)", True, False, True)
passes2("+ pageable 2.9 MB download on side, no wait", True, True, False)
passes2("+ pageable 2.9 MB download on side after side.wait_event(main's event)", True, True, True)

# ---- what could a banded canvas download buy the page-locked pipeline pass?  upload in `chunks` pieces on side; from `start_us` on (the solve
#      is done, the row ranges are known) download in `bands` pieces on a third stream, each after "its" chunk; no kernels in between ----
def banded(chunks, bands, start_us, n=6):
    side, down = s1, s3
    res = []
    cb, db = big_h.numel() // chunks, down_h.numel() // bands
    for i in range(n):
        torch.cuda.synchronize()
        e0, e_up, e_dn = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        landed = []
        t0 = time.perf_counter()
        e0.record(side)
        with torch.cuda.stream(side):
            for c in range(chunks):
                big_d[c * cb:(c + 1) * cb].copy_(big_h[c * cb:(c + 1) * cb], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
                landed.append(ev)
            e_up.record(side)
        while (time.perf_counter() - t0) * 1e6 < start_us:
            pass
        with torch.cuda.stream(down):
            for b in range(bands):
                down.wait_event(landed[min(chunks - 1, (b + 1) * chunks // bands)])      # band b reads up to the chunk below it
                down_h[b * db:(b + 1) * db].copy_(down_d[b * db:(b + 1) * db], non_blocking=True)
            e_dn.record(down)
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e_up) * 1e3, e0.elapsed_time(e_dn) * 1e3, (time.perf_counter() - t0) * 1e6))
    res.sort(key=lambda r: r[2])
    u, d, w = res[len(res) // 2]
    print(f"chunks {chunks:2d} bands {bands:2d} downloads from {start_us:4d} us: upload done {u:5.0f}  download done {d:5.0f}  wall {w:5.0f} us")


print("banded download against the upload's tail (page-locked, no kernels):")
banded(1, 1, 0)
for ch, bd in ((4, 4), (4, 8), (8, 8), (2, 4)):
    banded(ch, bd, 370)
