"""bench.py's eval_loop measurement on its own (serial against pipelined evaluation loop on the ZJU-sized frame)"""
import json, os, sys
from types import SimpleNamespace as NS
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fill = sys.argv[2] if len(sys.argv) > 2 else "survey"
reserve = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (0,)
print(json.dumps(bench.eval_loop_wall(NS(seed=0), frames=frames, fill=fill, reserve=reserve), indent=1))
