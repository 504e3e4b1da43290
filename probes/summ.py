import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); c=d["config"]
        print(f.split("/")[-1], "value %.3g ms/step %.2f" % (d["value"], d["ms_per_step"]), {k: round(v,2) for k,v in c["kernel_ms"].items()}, c.get("regions_on_ring_kernels"), "groups", c["result_groups"], "roof %.3f" % d["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
