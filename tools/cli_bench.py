"""GPU box: the CLI leg of bench.py alone (the three product CLIs on a synthetic Lyft-shaped KITTI tree)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

a = bench.parse(sys.argv[1:])
print(json.dumps(bench.cli_bench(a, 0), indent=1))
