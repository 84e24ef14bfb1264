#!/usr/bin/env python
"""Command-line twin of project/runml/predict.py: python tools/predict.py --model zero_dce_re --weights w.pt --source dir --save-dir out"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mon_amd

ap = argparse.ArgumentParser()
ap.add_argument("--model", required=True)
ap.add_argument("--weights", default=None)
ap.add_argument("--source", required=True)
ap.add_argument("--save-dir", required=True)
ap.add_argument("--image-size", type=int, default=512)
ap.add_argument("--resize", action="store_true")
ap.add_argument("--compute-dtype", default=None)
ap.add_argument("--save-debug", action="store_true")
a = ap.parse_args()
cfg = {"name": a.model}
if a.weights:
    cfg["weights"] = a.weights
if a.compute_dtype:
    cfg["compute_dtype"] = a.compute_dtype
print(json.dumps(mon_amd.image.predict(cfg, a.source, a.save_dir, a.image_size, a.resize, save_debug=a.save_debug)))
