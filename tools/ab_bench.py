#!/usr/bin/env python
"""Same-box A/B of library builds on the headline workload (boxes of the pool differ by 3-5 %, so only numbers from ONE gpurun
call compare).  python tools/ab_bench.py [--rounds 2] [--args "--dtype f16"] tagA tagB ...   tag = libie_<tag>.so in the package dir
("hip" = the default build); "tag@VAR=value,VAR2=value" runs that build with the environment variables set ("ARGS=..." = extra bench.py arguments for that tag).  Prints per round frames/s and the per-kernel ms per 64-frame step, interleaved A B A B."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--args", default="")
ap.add_argument("tags", nargs="+")
a = ap.parse_args()
res = {t: [] for t in a.tags}
for r in range(a.rounds):
    for t in a.tags:
        lib, _, extra = t.partition("@")
        env = dict(os.environ, IE_HIP_LIB=os.path.join(ROOT, "etss-07-imageenhance_amd", f"libie_{lib}.so"), IE_HIP_LIB_ABI_OK="1")
        kvs = dict(kv.split("=", 1) for kv in extra.split(",") if kv)
        more = kvs.pop("ARGS", "").split()          # "tag@ARGS=--variant 256": extra bench.py arguments for this tag only
        env.update(kvs)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-extras", "--no-cpu-baseline"]
                             + a.args.split() + more, env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            k = j["roofline"]["kernel_ms_per_step"]
            res[t].append(j["value"])
            print(f"round {r} {t:24s} {j['value']:8.1f} frames/s  " + "  ".join(f"{n.split('_')[1]} {v:.3f}" for n, v in k.items()) +
                  f"  err {(j.get('parity') or {}).get('max_rel_err')}", flush=True)
        except Exception as e:      # noqa: BLE001
            print(f"round {r} {t}: failed ({e}): {out.stderr[-400:]}", flush=True)
for t, v in res.items():
    if v:
        print(f"{t:12s} median {sorted(v)[len(v) // 2]:8.1f}  max {max(v):8.1f}")
