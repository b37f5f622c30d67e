import json, sys
d = json.load(open(sys.argv[1]))
print("ops/s %.4g  ms/step %.2f  dominant %s frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]))
print({k: v["ms"] for k, v in d["kernels"].items() if v["ms"] > 0.8})
