import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, signalalign_amd as sa, sa_cases as cases
pm = sa.Model.load(cases.MODEL_6MER); p = sa.default_params()
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 48
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
jobs = cases.realistic_anchor_jobs(cases.MODEL_6MER, NR, NE)
VAR = sys.argv[3] if len(sys.argv) > 3 else "SA_WIDE_KERNEL"   # or SA_WIDE_BWD
res = {}
for v in ("0", "1"):
    os.environ[VAR] = v
    b = sa.Batch(pm, p, jobs); b.run(); b.run()
    res[v] = [b.pairs(j) for j in range(len(jobs))]
    print(VAR, v, "pairs", sum(len(x) for x in res[v]), "fwd ms %.3f bwd ms %.3f" % (b.stats().ms_forward, b.stats().ms_backward))
    b.close()
same = all(np.array_equal(a, c) for a, c in zip(res["0"], res["1"]))
print("identical:", same)
if not same:
    for j, (a, c) in enumerate(zip(res["0"], res["1"])):
        if not np.array_equal(a, c):
            print("first differing read", j, len(a), len(c)); break
