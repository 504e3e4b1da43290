"""How far the exact totals of a traceback lie from its speculative total (SA_SPEC_DEBUG=1 prints the range per run):
Gaussian CpG reads, HDP reads with CpG ambiguity, HDP reads with sparse anchors, Gaussian reads with sparse anchors."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ["SA_SPEC_DEBUG"] = "1"
import numpy as np
import signalalign_amd as sa
import sa_cases as cases

def run(name, pm, p, jobs, amb=None):
    print("==", name, flush=True)
    b = sa.Batch(pm, p, jobs, ambig=amb); b.run()
    st = b.stats(); print("   regions %d ring %d strip %d pairs %d" % (st.n_regions, st.n_ring_regions, st.n_strip_regions, sum(b.n_pairs(j) for j in range(len(jobs)))), flush=True)
    b.close()

pg = sa.Model.load(cases.MODEL_CPG)
run("gaussian cpg", pg, sa.default_params(), cases.synthetic_jobs(cases.MODEL_CPG, 40, 5000, 20, cpg_ambiguous=True), sa.default_ambig({"X": "CE"}))
pm = sa.Model.load(cases.MODEL_R73, cases.NHDP); pm.set_to_hdp_expected_values()
hj = cases.hdp_jobs(40, 5000, 0, table5=pm.table5())
hx = [dict(j, ref=j["ref"].replace("CG", "XG")) for j in hj]
run("hdp cpg thr 0.1", pm, sa.default_params(threshold=0.1), hx, sa.default_ambig({"X": "CE"}))
def thin(j, step):
    q = dict(j); keep = np.zeros(len(q["ax"]), dtype=bool); keep[::step] = True; q["ax"], q["ay"] = q["ax"][keep], q["ay"][keep]; return q
run("hdp sparse anchors", pm, sa.default_params(threshold=0.1), [thin(j, 29) for j in hj])
p6 = sa.Model.load(cases.MODEL_6MER)
run("gaussian realistic anchors", p6, sa.default_params(), cases.realistic_anchor_jobs(cases.MODEL_6MER, 40, 5000, 0))
