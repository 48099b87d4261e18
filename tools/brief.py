#!/usr/bin/env python3
"""print a one-line digest of bench.py's JSON line read from stdin"""
import json, sys
lines = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
r = d.get("roofline", {})
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), d["config"].get("split"), "group", d["config"].get("group"),
      "ok" if d.get("results_ok") else "BAD", "path_frac %.3f" % r.get("path", {}).get("frac", 0),
      {k: round(v, 3) for k, v in r.get("kernel_ms_per_step", {}).items()})
