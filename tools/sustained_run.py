"""Sustained-clock evidence: runs bench.py for >= 10 s of GPU time while sampling shader / memory clocks and
power with rocm-smi, and writes one JSON record.   python tools/sustained_run.py <out.json> [--steps 500] [-- extra bench args]

The default bench window (20 steps = 0.5 s) says nothing about what the part sustains under its power limit; this does."""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sample(stop, rows):
    while not stop.is_set():
        t = time.time()
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--showuse", "--showtemp"], capture_output=True, text=True,
                                 timeout=10).stdout
        except Exception as e:  # noqa: BLE001
            out = f"error {e}"
        row = {"t": t}
        m = re.search(r"sclk clock level: \d+:? \((\d+)Mhz\)", out)
        if m:
            row["sclk_mhz"] = int(m.group(1))
        m = re.search(r"mclk clock level: \d+:? \((\d+)Mhz\)", out)
        if m:
            row["mclk_mhz"] = int(m.group(1))
        m = re.search(r"fclk clock level: \d+:? \((\d+)Mhz\)", out)
        if m:
            row["fclk_mhz"] = int(m.group(1))
        m = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
        if m:
            row["power_w"] = float(m.group(1))
        m = re.search(r"GPU use \(%\): (\d+)", out)
        if m:
            row["gpu_use_pct"] = int(m.group(1))
        m = re.search(r"Temperature \(Sensor junction\) \(C\): ([\d.]+)", out)
        if m:
            row["temp_junction_c"] = float(m.group(1))
        if len(row) == 1:
            row["raw"] = out[:400]
        rows.append(row)
        stop.wait(0.4)


def main():
    out_path = sys.argv[1]
    steps = 500
    extra = []
    args = sys.argv[2:]
    if "--steps" in args:
        steps = int(args[args.index("--steps") + 1])
    if "--" in args:
        extra = args[args.index("--") + 1:]
    rows, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(stop, rows), daemon=True)
    th.start()
    time.sleep(1.0)  # idle samples first
    t0 = time.time()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "20", "--no-cpu-baseline", "--no-forward-only", "--no-kernel-table", "--no-config5", "--no-bigmlp"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True)
    t1 = time.time()
    time.sleep(1.0)
    stop.set()
    th.join(timeout=15)
    line = None
    for ln in res.stdout.splitlines():
        if ln.startswith("{"):
            line = json.loads(ln)
    busy = [r for r in rows if t0 <= r["t"] <= t1 and r.get("gpu_use_pct", 0) > 50]
    summ = {}
    for k in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "temp_junction_c"):
        v = [r[k] for r in busy if k in r]
        if v:
            summ[k] = {"min": min(v), "max": max(v), "mean": round(sum(v) / len(v), 1), "n": len(v)}
    rec = {"what": "bench.py sustained run with rocm-smi sampled every ~0.4 s (samples with GPU use > 50 % summarised)", "cmd": " ".join(cmd[1:]),
           "steps": steps, "wall_s": round(t1 - t0, 1), "bench": None if line is None else {k: line[k] for k in ("ms_per_step", "value", "steps", "kernels") if k in line},
           "busy_summary": summ, "samples": rows, "stderr_tail": res.stderr[-1500:] if line is None else ""}
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({k: rec[k] for k in ("wall_s", "bench", "busy_summary")})[:3000])


if __name__ == "__main__":
    main()
