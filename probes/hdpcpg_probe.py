import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import signalalign_amd as sa
import sa_cases as cases
pm = sa.Model.load(cases.MODEL_R73, cases.NHDP); pm.set_to_hdp_expected_values()
jobs = cases.hdp_jobs(300, 5000, 0, table5=pm.table5())
for j in jobs: j["ref"] = j["ref"].replace("CG", "XG")
p = sa.default_params(threshold=0.1)
amb = sa.default_ambig({"X": "CE"})
for it in range(3):
    t=time.time(); b = sa.Batch(pm, p, jobs, ambig=amb); st=b.stats(); print("create %.3f s device_bytes %.2f GB f_bytes %.2f GB regions %d ring %d"%(time.time()-t, st.device_bytes/1e9, st.f_bytes/1e9, st.n_regions, st.n_ring_regions))
    for r in range(2):
        t=time.time(); b.run(); print("  run %.1f ms  fwd %.2f bwd %.2f fold %.2f"%((time.time()-t)*1e3, b.stats().ms_forward, b.stats().ms_backward, b.stats().ms_fold))
    n=sum(b.n_pairs(j) for j in range(len(jobs))); print("  pairs", n, "per event %.3f"%(n/sum(len(j["events"]) for j in jobs)))
    b.close()
