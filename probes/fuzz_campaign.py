"""An extended run of tests/test_gpu_fuzz.py::test_random_shapes with seeds the suite does not use (the suite keeps three fixed
seeds so that it stays fast): random shapes x three parameter sets x EXACT / default / memory-resident kernels against the CPU
restatement.  usage (GPU box): python probes/fuzz_campaign.py [first_seed] [n_seeds]  ->  one line per seed, a summary line"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sa_oracle_py as oracle
import sa_cases as cases
import test_gpu_fuzz as fz
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
t0 = time.time()
for seed in range(first, first + n):
    for model, amb in ((cases.MODEL_6MER, False), (cases.MODEL_5MER, False), (cases.MODEL_CPG, True)):
        try:
            fz.test_random_shapes(oracle, model, amb, seed, require_strips=False)
            print("seed %d %s ambiguous=%s ok (%.0f s)" % (seed, os.path.basename(model)[:28], amb, time.time() - t0), flush=True)
        except AssertionError as ex:
            import traceback
            bad += 1
            tb = traceback.extract_tb(ex.__traceback__)[-1]
            print("seed %d %s ambiguous=%s FAILED at %s:%d `%s`: %s" % (seed, os.path.basename(model), amb, os.path.basename(tb.filename),
                                                                         tb.lineno, tb.line, str(ex)[:300]), flush=True)
for seed in range(first, first + n, 4):   # the HDP / expectation passes (with and without ambiguity letters): every fourth seed
    try:
        fz.test_random_shapes_hdp_and_expectations(oracle, seed0=seed)
        print("seed %d hdp + expectations ok (%.0f s)" % (seed, time.time() - t0), flush=True)
    except AssertionError as ex:
        import traceback
        bad += 1
        tb = traceback.extract_tb(ex.__traceback__)[-1]
        print("seed %d hdp + expectations FAILED at %s:%d `%s`: %s" % (seed, os.path.basename(tb.filename), tb.lineno, tb.line,
                                                                      str(ex)[:300]), flush=True)
print("fuzz campaign: seeds %d..%d, %d failures" % (first, first + n - 1, bad))
sys.exit(1 if bad else 0)
