import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
print({k:v["ms_per_step"] for k,v in list(d["kernels"].items())[:10]})
