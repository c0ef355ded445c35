#!/usr/bin/env python3
"""The `call_level` and `pipeline` objects of bench.py alone (numpy in -> numpy out through the host-buffer entry points),
several times in one process:  python tools/call_level.py [config] [repeats]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    c = bench.call_level(cfg)
    c.pop("note")
    print(json.dumps(c))
p = bench.pipeline_level(cfg)
p.pop("note")
print(json.dumps(p))
