import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f, "ops/s %.4g  ms/step %.2f  dominant %s frac %.4f  kernels %.1f ms" % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("kernel_ms_total", 0)))
