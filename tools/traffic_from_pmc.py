#!/usr/bin/env python
"""profiles/traffic.json from a tools/pmc_summary.py listing of the FETCH_SIZE / WRITE_SIZE pass: HBM bytes per launch and kernel
(bench.py reports them as roofline.traffic).  Kernel names lose their dtype suffix so they match bench.py's KERNEL_WORK keys."""
import ast, json, re, sys
out = {}
parts = {}
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = ast.literal_eval(line)
    if "zdce" not in d["kernel"]:
        continue
    name = re.sub(r"(_e|_ea|_u8)?_(f16|bf16|mix)$", "", d["kernel"].split(" grid=")[0])      # (zdce_k67n_*_e_*: the form compiled for `enhanced` alone)
    if "fetch_MB_x2" in d:
        parts.setdefault(name, {})["read"] = int(d["fetch_MB_x2"] * 1024 * 1024)
    if "write_MB" in d and d.get("write_MB"):
        parts.setdefault(name, {})["write"] = int(d["write_MB"] * 1024 * 1024)
for name, rw in parts.items():
    if "read" in rw and "write" in rw:
        out[name] = rw["read"] + rw["write"]
        out["_" + name + "_read_write"] = [rw["read"], rw["write"]]
out["_source"] = "rocprofv3 --pmc FETCH_SIZE WRITE_SIZE on `bench.py --steps 2 --warmup 1 --no-extras` (launches of 16 frames 1080p); FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KB -> bytes"
try:      # the per-workload entries (tools/pmc_models.sh passes of configs 3 / 4) live in the same file: carry them over
    import os
    old = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")))
    if "workloads" in old:
        out["workloads"] = old["workloads"]
except Exception:
    pass
print(json.dumps(out, indent=1))
