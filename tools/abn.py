"""N-way A/B on ONE GPU box:  python tools/abn.py [--rounds R] name=lib.so[,ENV=VAL...] ...
Runs bench.py's default pass for every variant in turn, R rounds (A B C A B C ...), and prints ms per frame and the
launch times of loop C's solver and of the density pass.  `lib.so` may be `-` for the product library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = 2
if args and args[0] == "--rounds":
    rounds = int(args[1])
    args = args[2:]
variants = []
for a in args:
    name, spec = a.split("=", 1)
    parts = spec.split(",")
    env = {}
    if parts[0] != "-":
        env["ARAH_LIB_PATH"] = os.path.abspath(parts[0])
    for kv in parts[1:]:
        k, v = kv.split("=", 1)
        env[k] = v
    variants.append((name, env))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-gpu-baseline", "--passes", "default",
                              "--no-train", "--steps", "6", "--warmup", "2", "--streams", os.environ.get("ABN_STREAMS", "1")], env=dict(os.environ, **env), capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[name].append((d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_k_density"]["avg_launch_ms"]))
        except Exception as e:   # noqa
            print("failed", name, out.stderr[-600:])
for name, _ in variants:
    print("%-14s" % name, "  ".join("%.2f/%.2f/%.2f" % x for x in res[name]), " (ms per frame / loop C / density)")
