"""Wall time of the drop-in command, fresh process per run: one C1 and one C3 pair through
``python -m cvx_proj_amd.apap``, as it is now (host-buffer entry points, no torch in the process), as it was (torch imported
for the library's sake: APAP_HIP_PRELOAD_TORCH=1) and through the resident pipeline (--resident), with the command's own
break-down (--timing: imports, runtime initialisation + code-object load, inputs, compute, savemat); then run_all.sh's pattern -
4 cases x 4 pictures - as 16 processes against one process (--cases 1-4 --imgs 1,2,4,5).
    python tools/cli_wall.py [repeats]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def run(args, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, "-m", "cvx_proj_amd.apap"] + args + ["--out-prefix", tmp + "/", "--timing"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise SystemExit(f"{' '.join(cmd)} failed:\n{r.stderr[-2000:]}")
        stages = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{")][-1])
        mats = sorted(f for _, _, fs in os.walk(tmp) for f in fs if f.endswith(".mat"))
    return wall, stages, mats


def report(tag, args, env_extra=None):
    walls, last = [], None
    for _ in range(reps):
        w, st, mats = run(args, env_extra)
        if not walls or w < min(walls):
            last = st              # the break-down shown is the fastest run's own
        walls.append(w)
    p = last["pairs"]
    comp = sum(q["compute_ms"] for q in p)
    print(f"{tag:58s} wall {min(walls) * 1e3:7.0f} ms (min of {reps}; median {sorted(walls)[len(walls) // 2] * 1e3:.0f})  | python + numpy start "
          f"{(min(walls) * 1e3 - last['total_ms']):5.0f}  imports {last['imports_ms']:5.0f}  runtime init {last['runtime_init_ms']:6.0f}  "
          f"inputs {sum(q['inputs_ms'] for q in p):6.0f}  compute {comp:7.1f} (first pair {p[0]['compute_ms']:.1f}"
          f"{', last ' + format(p[-1]['compute_ms'], '.1f') if len(p) > 1 else ''})  savemat {sum(q['save_ms'] for q in p):5.0f}  [{len(mats)} .mat]",
          flush=True)
    return min(walls)


for cfg in ("C1", "C3"):
    report(f"{cfg} one pair, host-buffer entry points (no torch)", ["1", "1", "--synth", cfg])
    report(f"{cfg} one pair, torch imported first (rounds 1-5)", ["1", "1", "--synth", cfg], {"APAP_HIP_PRELOAD_TORCH": "1"})
    report(f"{cfg} one pair, --resident (pipeline, torch)", ["1", "1", "--synth", cfg, "--resident"])
one = report("C1 one pair again (the unit of run_all.sh)", ["1", "1", "--synth", "C1"])
loop = report("C1 4 cases x 4 pictures in ONE process", ["--cases", "1-4", "--imgs", "1,2,4,5", "--synth", "C1"])
print(f"run_all.sh's pattern: 16 processes = {16 * one * 1e3:.0f} ms, one process = {loop * 1e3:.0f} ms")
loop3 = report("C3 4 cases x 4 pictures in ONE process", ["--cases", "1-4", "--imgs", "1,2,4,5", "--synth", "C3"])
