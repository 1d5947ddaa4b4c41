import json, sys
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.flat_area_sweep(0), indent=1))
