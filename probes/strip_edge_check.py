import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import signalalign_amd as sa
from signalalign_amd import synth
import sa_cases as cases
from oracle import sa_oracle_py as oracle
alpha,k,t10,tab=synth.parse_model_table(cases.MODEL_6MER)
pm=sa.Model.load(cases.MODEL_6MER); om=oracle.Model(alpha,k,t10,tab)
jobs=cases.realistic_anchor_jobs(cases.MODEL_6MER, 4, 2500, 600)
for n_ev, idx in ((40, 7), (130, 8), (700, 9)):
    r=synth.make_read(idx,n_ev,alpha,k,tab); jobs.append(dict(r, ax=np.zeros(0,dtype=np.int64), ay=np.zeros(0,dtype=np.int64)))
for thr in (0.01, 0.0, 0.9):
    p=sa.default_params(threshold=thr); op=cases.oracle_params(oracle,p)
    b=sa.Batch(pm,p,jobs); b.run(); st=b.stats()
    first=[b.pairs(j) for j in range(len(jobs))]
    b.run()
    second=[b.pairs(j) for j in range(len(jobs))]
    b.close()
    worst=0
    for j,job in enumerate(jobs):
        assert np.array_equal(first[j], second[j]), ("rerun differs", thr, j)
        exp=cases.oracle_pairs(oracle, om, job, op)
        keys=[(int(r["x"]),int(r["y"])) for r in first[j]]
        if len(set(keys))!=len(keys):
            from collections import Counter
            dup=[k_ for k_,c in Counter(keys).items() if c>1]
            print("DUP thr",thr,"job",j,"events",len(job["events"]),"ndup",len(dup),"first dups",dup[:6],"x+y+2",[a+b+2 for a,b in dup[:6]], "npairs",len(keys))
            continue
        w,lonely=cases.compare_pairs(first[j], exp, 100, p.threshold); worst=max(worst,w)
        assert cases.same_order(first[j], exp)
    print("threshold", thr, "strip regions", st.n_strip_regions, "of", st.n_regions, "worst", worst, "pairs", sum(len(f) for f in first))
print("ok")
