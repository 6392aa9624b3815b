"""Development aid: bench.py's C5 multi-device leg alone (wfa_hip_multi_align_batch over every visible device: host ASCII in -> host
results and op bytes out)."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
print(json.dumps(bench.multi_leg(int(sys.argv[1]) if len(sys.argv) > 1 else 8192)))
