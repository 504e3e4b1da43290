import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import signalalign_amd as sa
import sa_cases as cases
pm = sa.Model.load(cases.MODEL_CPG)
amb = sa.default_ambig({"X": "CE"})
names = ["m>m","m>x","m>y","x>m","x>x","x>y","y>m","y>x","y>y"]
jobs = cases.synthetic_jobs(cases.MODEL_CPG, 3, 1400, 20, cpg_ambiguous=True)
sparse = cases.realistic_anchor_jobs(cases.MODEL_CPG, 2, 1200, 77)
jobs += [dict(j, ref=j["ref"].replace("CG", "XG")) for j in sparse]
jobs += cases.synthetic_jobs(cases.MODEL_CPG, 1, 60, 5, cpg_ambiguous=True)
def run(js, kw, tag):
    p = sa.default_params(**kw)
    a, la, _ = sa.expect_batch(pm, p, js, ambig=amb)
    st = sa.expect_last_stats()
    b, lb, _ = sa.expect_batch(pm, p, js, ambig=amb, flags=sa.FLAG_FORCE_GENERIC)
    for i in range(len(js)):
        d = a[i] - b[i]
        print(tag, i, len(js[i]["events"]), "ring", st.n_ring_regions, "segments", st.n_segments,
              " ".join("%s %+.4g" % (nm, v) for nm, v in zip(names, d) if abs(v) > 1e-9))
for kw in (dict(), dict(expansion=20, trace_back=30, min_diags=150)):
    run(jobs, kw, "all")
    run(jobs[3:5], kw, "sparse2")
    run(jobs[:3], kw, "dense3")
    run(jobs[4:5], kw, "one")
