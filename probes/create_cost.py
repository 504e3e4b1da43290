"""sa_batch_create alone, repeatedly, on the headline and the configs[2] reads: python probes/create_cost.py [run]"""
import sys, time
sys.path.insert(0, ".")
import signalalign_amd as sa
from signalalign_amd import synth
import os
G = "tests/golden/models/"
do_run = len(sys.argv) > 1
for name, model, kw, amb in (("gaussian", G + "testModelR9.4_450bps.nucleotide.6mer.template.model", {}, None),
                             ("cpg", G + "testModelR9.4_450bps.cpg.6mer.template.model", {"cpg_ambiguous": True}, {"X": "CE"})):
    alpha, k, t10, tab = synth.parse_model_table(model)
    pm = sa.Model.load(model)
    arrs = [sa.JobArray([synth.make_read(i + 2000 * q, 5000, alpha, k, tab, **kw) for i in range(2000)]) for q in range(3)]
    ambig = sa.default_ambig(amb) if amb else None
    p = sa.default_params()
    ts = []
    for rep in range(12):
        t0 = time.perf_counter()
        b = sa.Batch(pm, p, arrs[rep % 3], ambig=ambig)
        t1 = time.perf_counter()
        if do_run:
            b.run()
        b.close()
        ts.append((t1 - t0) * 1e3)
    print(name, "create ms:", " ".join("%.1f" % t for t in ts))
