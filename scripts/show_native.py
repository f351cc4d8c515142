import json, sys
d = json.loads(open(sys.argv[1]).read())
print(round(d["ms_per_step"], 3), {k: round(v["ms_per_launch"] * 1e3, 1) for k, v in d["native"].items()})
