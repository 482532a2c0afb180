#!/usr/bin/env python3
"""Print the float32-against-float64 and device-against-numpy disagreement rates per (rule, code) as
JSON (the measurement itself: tests/precision_rates.py).

    python tools/precision_rates.py [frames_per_case] > gpurun_out/precision.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import precision_rates  # noqa: E402

print(json.dumps(precision_rates.measure(want=int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17, verbose=True), indent=1))
