"""print selected fields of a bench.py JSON line read from stdin:  python tools/pick.py value ms_per_step config.stage_ms"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
out = []
for key in sys.argv[1:]:
    v = d
    for part in key.split("."):
        v = v.get(part) if isinstance(v, dict) else None
    out.append("%s=%s" % (key, round(v, 4) if isinstance(v, float) else v))
print(" ".join(out))
