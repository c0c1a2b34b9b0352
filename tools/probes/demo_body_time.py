"""bench.py's demo_render_body_frame measurement on its own"""
import json, os, sys
from types import SimpleNamespace as NS
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
print(json.dumps(bench.demo_render_body_frame(NS(seed=0)), indent=1))
