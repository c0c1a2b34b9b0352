"""bench.py's eval_loop measurement on its own (serial against pipelined evaluation loop on the ZJU-sized frame)"""
import json, os, sys
from types import SimpleNamespace as NS
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
print(json.dumps(bench.eval_loop_wall(NS(seed=0), frames=int(sys.argv[1]) if len(sys.argv) > 1 else 12), indent=1))
