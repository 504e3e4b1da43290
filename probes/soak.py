"""Soak: thousands of create / run / read / destroy cycles (alignment, expectation pass, HDP) on changing read sets; prints the
process's resident memory and the device's free memory along the way -- neither may creep.  Usage: python probes/soak.py [cycles]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import signalalign_amd as sa
from signalalign_amd import synth
import sa_cases as cases


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0
    return 0.0


cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
pm = sa.Model.load(cases.MODEL_6MER)
ph = sa.Model.load(cases.MODEL_R73, cases.NHDP)
ph.set_to_hdp_expected_values()
sets = [sa.JobArray(synth.make_jobs(200, 2000, alpha, k, tab, first_index=1000 * q)) for q in range(4)]
hsets = [sa.JobArray(cases.hdp_jobs(100, 1500, 500 * q, table5=ph.table5())) for q in range(2)]
# the same reads in page-locked blocks of the caller (SA_FLAG_INPUTS_IN_HOST_BLOCK), anchors and events apart / interleaved
bsets = [sa.JobArray(synth.make_jobs(200, 2000, alpha, k, tab, first_index=1000 * q), host_block=True, interleaved=bool(q & 1)) for q in range(4)]
p = sa.default_params()
ph_p = sa.default_params(threshold=0.1)
t0 = time.time()
pairs = 0
for c in range(cycles):
    b = sa.Batch(pm, p, sets[c % 4])
    b.run()
    pairs += b.n_pairs(c % 200)
    if c % 7 == 0:
        b.mea()
    b.close()
    if c % 5 == 0:
        sa.expect_batch(pm, p, sets[(c + 1) % 4])
    if c % 2 == 0:
        b = sa.Batch(pm, p, bsets[c % 4], flags=sa.FLAG_INPUTS_IN_HOST_BLOCK, deferred=(c % 4 == 0))
        if c % 8 == 6:
            b.close()          # created, never run: the event records may still be on their way
        else:
            b.start()
            b.wait()
            pairs += b.n_pairs(c % 200)
            b.close()
    if c % 400 == 399:         # blocks come and go
        q = (c // 400) % 4
        bsets[q] = sa.JobArray(synth.make_jobs(200, 2000, alpha, k, tab, first_index=1000 * q + c), host_block=True, interleaved=bool(q & 1))
    if c % 3 == 0:
        h = sa.Batch(ph, ph_p, hsets[c % 2])
        h.run()
        pairs += h.n_pairs(0)
        h.close()
    if c % 250 == 0 or c == cycles - 1:
        free, total = sa.device_memory(0)
        print("cycle %5d  %.0f s  rss %.0f MB  device free (parked blocks count as free) %.2f GB of %.0f  pairs seen %d"
              % (c, time.time() - t0, rss_mb(), free / 1e9, total / 1e9, pairs), flush=True)
