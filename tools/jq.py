"""Development aid: print chosen keys of the last JSON line on stdin.  python bench.py ... | python tools/jq.py value ms_per_step"""
import json, sys
lines = [l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")]
j = json.loads(lines[-1]) if lines else {}
print({k: j.get(k) for k in sys.argv[1:]} if len(sys.argv) > 1 else json.dumps(j, indent=1))
