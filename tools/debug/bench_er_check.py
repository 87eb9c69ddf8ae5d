import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "value %.4g frac %.3f" % (d["value"], d["roofline"]["frac"]), "er %.4g" % d["value_early_reject"],
          "%.3f ms" % d["early_reject"]["kernel_ms_per_launch"], d["early_reject"]["kernel"],
          "piecewise %.4g direct %.4g" % (d["value_piecewise"], d["value_direct_form"]))
