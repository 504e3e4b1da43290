#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X pair-HMM path.

One "step" = one batch of FRESH synthetic reads through the whole C-ABI boundary: sa_batch_create (input checks, planning --
band, split, traceback schedule, on the device when the batch allows it -- and upload), sa_batch_run (forward,
backward/posterior, fold, finalisation, result copy to the host), results read, sa_batch_destroy.  The next batch is created
while the current one is on the GPU (sa_batch_start / sa_batch_wait), as a pipeline that sees every read once would do it;
consecutive steps take different read sets.  The kernels-only rate on a planned, HBM-resident batch (what round 1
reported as its headline) is measured first and reported beside it (config.kernels_only_resident_inputs); the roofline
figures come from that phase's HIP-event times.

Workload at every N: BASELINE.json configs[1] per GPU -- R9.4 6-mer template Gaussian HMM, 2000 synthetic 5k-event reads,
band (diagonal expansion) 50, threshold 0.01, traceBackDiagonals 100 -- i.e. weak scaling: reads are independent, each rank
aligns its own (different seeds), no collective on the data path.  `--gpus N` without a launcher starts the N ranks itself.

metric: DP cell updates per second: sum over reads, traceback segments and anti-diagonals of
width x paths, forward sweep plus backward sweep actually executed.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

T_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

MODEL = os.path.join(ROOT, "tests", "golden", "models", "testModelR9.4_450bps.nucleotide.6mer.template.model")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_CELL = 24.0     # 3 fp64 states of every band cell, written once, read once
ISSUE_PEAK_VALU_PER_S = 1024 / 1.83e-9   # 256 CUs x 4 SIMDs, one fp64 VALU wave-instruction per 1.83 ns each (probes/issue_probe.hip)


def host_cpu_info():
    """Socket model, physical cores per socket and the logical CPUs of socket 0's physical cores (first hardware thread of
    each core), from /proc/cpuinfo; plus what this process may actually use (affinity mask, cgroup quota)."""
    model, phys = "unknown", {}
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k_, v_ = [t.strip() for t in line.split(":", 1)]
                cur[k_] = v_
            elif cur:
                model = cur.get("model name", model)
                key = (int(cur.get("physical id", 0)), int(cur.get("core id", 0)))
                phys.setdefault(key, []).append(int(cur["processor"]))
                cur = {}
    except Exception:
        pass
    sockets = sorted({k_[0] for k_ in phys}) or [0]
    s0 = sorted(min(v_) for k_, v_ in phys.items() if k_[0] == sockets[0])
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        allowed = list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    return dict(model=model, sockets=len(sockets), cores_per_socket=len(s0) or None, socket0_cpus=s0, allowed=allowed,
                cgroup_cpu_quota=quota)


def cpu_baseline(om, oparams, make_sample, n_events, reads_per_thread, first_index, ambig=None, what=""):
    """The CPU restatement (oracle, 'port') on a bounded sample of the same workload, one read per thread, threads pinned
    to the physical cores of ONE socket (north_star: single-socket baseline); a one-thread figure beside it.
    om / oparams / ambig: the oracle's model (HDP loaded where the workload has one), parameters and ambiguity table;
    make_sample(n, first_index): n reads of the workload."""
    from oracle import sa_oracle_py as oracle
    info = host_cpu_info()
    pin = [c for c in info["socket0_cpus"] if c in set(info["allowed"])] or info["allowed"]
    cores = len(pin)
    if info["cgroup_cpu_quota"]:
        cores = max(1, min(cores, int(info["cgroup_cpu_quota"])))
    cores = min(cores, int(os.environ.get("SA_CPU_BASELINE_THREADS", "64")))
    pin = pin[:cores]
    old = None
    try:
        old = os.sched_getaffinity(0)
        os.sched_setaffinity(0, pin)
    except Exception:
        old = None
    try:
        one = make_sample(max(4, min(8, reads_per_thread)), first_index)
        t0 = time.perf_counter()
        _, c1 = oracle.align_batch_mt(om, one, oparams, 1, ambig=ambig)
        dt1 = time.perf_counter() - t0
        n = cores * reads_per_thread
        jobs = make_sample(n, first_index + 1000)
        t0 = time.perf_counter()
        npairs, cells = oracle.align_batch_mt(om, jobs, oparams, cores, ambig=ambig)
        dt = time.perf_counter() - t0
    finally:
        if old is not None:
            try:
                os.sched_setaffinity(0, old)
            except Exception:
                pass
    return dict(value=float(cells.sum() / dt), unit="cell_updates/s", cores=cores, kind="port",
                sample="%d reads x %d events%s (same generator and parameters as the GPU workload), oracle/sa_oracle.c, "
                       "%d threads pinned to physical cores of socket 0, %.1f s wall; 1 thread: %d reads, %.1f s"
                       % (n, n_events, what, cores, dt, len(one), dt1),
                events_per_s=float(sum(len(j["events"]) for j in jobs) / dt),
                one_thread_value=float(c1.sum() / dt1),
                # the GPU box grants this job a CPU quota below one socket: the full-socket figure is bounded from above by
                # linear scaling of the one-thread rate over the socket's physical cores
                single_socket_linear_extrapolation=float(c1.sum() / dt1) * float(info["cores_per_socket"] or cores),
                socket_model=info["model"], sockets=info["sockets"], physical_cores_per_socket=info["cores_per_socket"],
                logical_cpus_allowed=len(info["allowed"]), cgroup_cpu_quota=info["cgroup_cpu_quota"])


def bench_event_align(args):
    """the event <-> k-mer pre-alignment (adaptive banded Viterbi), same reads as the headline
    workload.  One step = one sa_event_align_batch call (upload, kernel, traceback, download)."""
    import signalalign_amd as sa
    from signalalign_amd import synth
    alpha, k, t10, tab = synth.parse_model_table(MODEL)
    pm = sa.Model.load(MODEL)
    jobs = []
    for i in range(args.reads):
        r = synth.make_read(i, args.events, alpha, k, tab)
        ev = np.ascontiguousarray(np.asarray(r["events4"])[:, 0])
        sh, sc = sa.scalings_mom(pm, r["ref"], ev)
        jobs.append(dict(sequence=r["ref"], event_mean=ev, scale=sc, shift=sh, var=1.0))
    stats = {}
    for _ in range(args.warmup):
        sa.event_align_batch(pm, jobs, stats=stats)
    t0 = time.perf_counter()
    kms = cms = 0.0
    for _ in range(args.steps):
        out = sa.event_align_batch(pm, jobs, stats=stats)
        kms += stats["kernel_ms"]
        cms += stats["call_ms"]
    dt = time.perf_counter() - t0
    cells = float(stats["cells"].sum())
    K = args.steps
    res = {"metric": "event_align_band_cell_updates_per_s", "value": cells * K / dt, "unit": "cell_updates/s", "n_gpus": 1,
           "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 scores / f64 emissions", "data": "synthetic",
           "config": {"workload": "adaptive banded event alignment (impl/eventAligner.c:899-1235), %d synthetic %d-event reads, "
                                  "bandwidth 100" % (args.reads, args.events),
                      "kernel_ms": kms / K, "c_call_ms": cms / K, "kernel_cell_updates_per_s": cells / (kms / K * 1e-3),
                      "reads_aligned": int(sum(1 for o in out if o[2] == 0)), "cells_per_read": cells / max(len(jobs), 1)}}
    if not args.no_cpu_baseline:
        from oracle import sa_oracle_py as oracle          # the CPU restatement: timed here as the baseline, nothing else
        from concurrent.futures import ThreadPoolExecutor
        om = oracle.Model(alpha, k, t10, tab)
        cores = min(os.cpu_count() or 1, 16)
        sample = jobs[:cores * 4]

        def one(j):
            m = oracle.Model(alpha, k, t10, tab)   # the read parameters live in the model object: one per thread call
            m.set_read_params(j["scale"], j["shift"], 1.0)
            return len(oracle.event_align(m, j["event_mean"], oracle.kmer_ids_of(m, j["sequence"]))[0])
        t1 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, sample))
        dtc = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": cells / len(jobs) * len(sample) / dtc, "unit": "cell_updates/s", "cores": cores,
                               "kind": "port", "sample": "%d reads, oracle/sa_oracle.c:sao_event_align, %d threads, %.1f s "
                                                        "wall (includes the Python k-mer id loop)" % (len(sample), cores, dtc)}
        del om
    print(json.dumps(res))


def bench_mea(args):
    """the maximum-expected-accuracy path over the posteriors of the headline workload's reads
    (src/signalalign/mea_algorithm.py:25-264).  The aligner produces the posteriors first (untimed); one step = one
    sa_mea_batch call (upload of the sparse matrices, kernel, traceback, download of the paths)."""
    import signalalign_amd as sa
    from signalalign_amd import synth
    alpha, k, t10, tab = synth.parse_model_table(MODEL)
    pm = sa.Model.load(MODEL)
    reads = synth.make_jobs(args.reads, args.events, alpha, k, tab)
    b = sa.Batch(pm, sa.default_params(), reads)
    b.run()
    jobs = []
    for j in range(len(reads)):
        pr = b.pairs(j)
        ev, rf, po, sh = sa.mea_params(pr["x"], pr["y"], np.round(pr["prob_e7"] / 1e7, 6))
        jobs.append(dict(event_idx=ev, ref_idx=rf, posterior=po, shortest=sh))
    # the chained form: matrices built on the device from the pairs the batch left in HBM (sa_batch_mea)
    cst = {}
    for _ in range(max(args.warmup, 1)):
        b.mea(stats=cst)
    ckms = ccms = 0.0
    for _ in range(args.steps):
        chained = b.mea(stats=cst)
        ckms += cst["kernel_ms"]
        ccms += cst["call_ms"]
    b.close()
    entries = float(sum(len(j["posterior"]) for j in jobs))
    stats = {}
    for _ in range(args.warmup):
        sa.mea_batch(jobs, stats=stats)
    kms = cms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = sa.mea_batch(jobs, stats=stats)
        kms += stats["kernel_ms"]
        cms += stats["call_ms"]
    dt = time.perf_counter() - t0
    K = args.steps
    res = {"metric": "mea_matrix_entries_per_s", "value": entries * K / dt, "unit": "entries/s", "n_gpus": 1, "steps": K,
           "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "maximum expected accuracy path (src/signalalign/mea_algorithm.py:25-264) over the posteriors "
                                  "of %d synthetic %d-event reads" % (args.reads, args.events),
                      "kernel_ms": kms / K, "c_call_ms": cms / K, "kernel_entries_per_s": entries / (kms / K * 1e-3),
                      "chained_kernel_ms": ckms / K, "chained_c_call_ms": ccms / K,
                      "chained_paths_found": int(sum(1 for o in chained if o[2] == 0)),
                      "entries_per_read": entries / max(len(jobs), 1),
                      "paths_found": int(sum(1 for o in out if o[2] == 0)),
                      "mean_path_length": float(np.mean([len(o[0]) for o in out]))}}
    if not args.no_cpu_baseline:
        from oracle import sa_oracle_py as oracle          # the CPU restatement: timed here as the baseline, nothing else
        from concurrent.futures import ThreadPoolExecutor
        cores = min(os.cpu_count() or 1, 16)
        reps = max(1, int(2e7 // max(entries / len(jobs) * min(len(jobs), cores * 8), 1)))
        sample = jobs[:cores * 8]

        def one(j):
            for _ in range(reps):
                oracle.mea(j["event_idx"], j["ref_idx"], j["posterior"], j["shortest"])
        t1 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, sample))
        dtc = time.perf_counter() - t1
        n_s = float(sum(len(j["posterior"]) for j in sample)) * reps
        res["cpu_baseline"] = {"value": n_s / dtc, "unit": "entries/s", "cores": cores, "kind": "port",
                               "sample": "%d reads x %d repetitions, oracle/sa_mea_oracle.c:sao_mea, %d threads, %.1f s wall"
                                         % (len(sample), reps, cores, dtc)}
    print(json.dumps(res))


def bench_expectations(args, compact=False):
    """The expectation pass (getExpectationsUsingAnchors, impl/pairwiseAligner.c:2164-2184: the inner loop of trainModels.py's
    EM): one step = one sa_expect_batch call over the headline workload's reads -- planning, upload, forward sweep storing all
    three states, backward sweep with the transition expectations fused in (k_bwd_fast_expect), fold, host rescale.  Cells are
    counted as for the alignment (forward + backward cell updates)."""
    import signalalign_amd as sa
    from signalalign_amd import synth
    cpg = args.workload == "expectations_cpg"   # EM training of a methylation model: every CpG cytosine C/E (configs[2]'s reads)
    model_path = os.path.join(ROOT, "tests", "golden", "models", "testModelR9.4_450bps.cpg.6mer.template.model") if cpg else MODEL
    ambig = sa.default_ambig({"X": "CE"}) if cpg else None
    alpha, k, t10, tab = synth.parse_model_table(model_path)
    pm = sa.Model.load(model_path)
    params = sa.default_params(threshold=args.threshold, expansion=50, trace_back=100)
    jobs = job_list = synth.make_reads_parallel(dict(kind="gauss", model=model_path, events=args.events,
                                                     kw={"cpg_ambiguous": True} if cpg else {}), range(args.reads))
    b = sa.Batch(pm, params, jobs, ambig=ambig)
    st = b.stats()
    cells = st.cells_forward + st.cells_backward
    b.close()
    jobs = sa.JobArray(jobs)   # marshalled once: a C caller holds sa_job_t arrays anyway
    for _ in range(max(1, args.warmup)):
        sa.expect_batch(pm, params, jobs, ambig=ambig)
    est = sa.expect_last_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trans, lik, _ = sa.expect_batch(pm, params, jobs, ambig=ambig)
    dt = (time.perf_counter() - t0) / args.steps
    dtg = float("nan")
    if not compact:   # (the memory-resident checker kernels beside it)
        tg0 = time.perf_counter()
        for _ in range(max(1, args.steps // 4)):
            sa.expect_batch(pm, params, jobs, ambig=ambig, flags=sa.FLAG_FORCE_GENERIC)
        dtg = (time.perf_counter() - tg0) / max(1, args.steps // 4)
    res = {"metric": "dp_cell_updates_per_s", "value": cells / dt, "unit": "cell_updates/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "expectation pass (sa_expect_batch) over BASELINE configs[%d]'s reads: %d synthetic %d-event reads%s, "
                                  "band=50" % (2 if cpg else 1, args.reads, args.events, ", every CpG cytosine C/E" if cpg else ""),
                      "regions": {"register": int(est.n_fast_regions), "ring": int(est.n_ring_regions), "all": int(est.n_regions)},
                      "step": "one sa_expect_batch call: create (host planner) + forward (all three states stored) + backward with "
                              "fused transition expectations + fold + host rescale + destroy",
                      "memory_resident_kernels_ms_per_step": None if compact else dtg * 1e3,
                      "memory_resident_kernels_value": None if compact else cells / dtg,
                      "mean_match_to_match_expectation": float(trans[:, 0].mean()), "mean_log_likelihood": float(lik.mean())},
           # 48 B per cell update: the forward sweep writes and the backward sweep reads all three states of every cell
           "roofline": {"bound": "issue", "bound_of_the_formula": "hbm", "kernel": "k_bwd_ring<EXPECT>" if cpg else "k_bwd_fast_expect", "achieved": None,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "note": "whole-call rate; the per-kernel split is in profiles/ (rocprofv3 --kernel-trace --stats)"}}
    if not args.no_cpu_baseline:
        from oracle import sa_oracle_py as oracle          # the CPU restatement: timed here as the baseline, nothing else
        from concurrent.futures import ThreadPoolExecutor
        info = host_cpu_info()
        cores = max(1, min(len(info["allowed"]), int(info["cgroup_cpu_quota"] or 64), 16))
        om_p = oracle.default_params()
        sample = job_list[:cores * 4]

        def one(j):
            m_ = oracle.Model(alpha, k, t10, tab)
            m_.set_read_params(j["scale"], j["shift"], j["var"])
            return oracle.expectations(m_, j["ref"], j["events"], j["ax"], j["ay"], om_p,
                                       **({"ambig": oracle.ambig_map({"X": "CE"})} if cpg else {}))[1]
        t1 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, sample))
        dtc = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": cells / len(job_list) * len(sample) / dtc, "unit": "cell_updates/s", "cores": cores,
                               "kind": "port", "sample": "%d reads, oracle/sa_oracle.c:sao_expectations, %d threads, %.1f s wall"
                                                         % (len(sample), cores, dtc)}
    # roofline of the dominant kernel from the per-kernel record of the last profiling session (profiles/kernel_times.json:
    # average duration of k_bwd_fast_expect over `bench.py --workload expectations` under rocprofv3 --kernel-trace --stats;
    # sa_expect_batch keeps no stage events of its own)
    try:
        kt = json.load(open(os.path.join(ROOT, "profiles", "kernel_times.json"))).get(args.workload, {})
        kms = kt.get("k_bwd_ring" if cpg else "k_bwd_fast_expect", {}).get("ms_per_step")
        if kms and args.reads == 2000 and args.events == 5000:
            # 48 B per backward cell update: the backward sweep reads all three forward states of every cell
            ach = 48.0 * st.cells_backward / (kms * 1e-3) / 1e9
            res["roofline"].update({"achieved": ach, "frac": ach / HBM_PEAK_GBS, "stage_ms": kms,
                                    "algorithmic_bytes_per_step": 48.0 * st.cells_backward,
                                    "source": "profiles/kernel_times.json (rocprofv3 --kernel-trace --stats), " + str(kt.get("_meta"))})
    except Exception:
        pass
    if not compact:
        print(json.dumps(res))
    return res


def dense_nhdp():
    """--workload hdp_dense: an .nhdp with a row of its own for most k-mers, built here by this repository's buildHdpUtil (a child
    process) from the bundled assignments of an R9.4 read (tests/golden/hdp/d6160b0b-...: 17 350 template rows): flat ACGT 6-mer model,
    3183 observed processes x 400 grid points -- no row that most k-mers share (the bundled templateSingleLevelFixed.nhdp has 352
    observed processes of 46 657; the reference's own comment puts a real HDP at 200 MB, tests/stateMachineTests.c:904, and
    dir_proc_density walks to the parent only for unobserved processes, impl/hdp.c:2588-2612).  Returns (nhdp path, path of a file
    holding the random ACGT sequence the reads' references are cut from); built once per box."""
    import gzip
    import subprocess
    import tempfile
    d = os.path.join(os.environ.get("SA_SYNTH_CACHE") or tempfile.gettempdir(), "sa_hdp_dense")
    os.makedirs(d, exist_ok=True)
    path, pool = os.path.join(d, "dense_flat_acgt_6mer.nhdp"), os.path.join(d, "dense_ref_pool.txt")
    if not (os.path.exists(path) and os.path.exists(pool)):
        asg = os.path.join(d, "assignments.tsv")
        src = os.path.join(ROOT, "tests", "golden", "hdp", "d6160b0b-a35e-43b5-947f-adaa1abade28.sm.assignments.tsv.gz")
        with open(asg, "w") as f:
            f.write(gzip.open(src, "rt").read())
        tool = os.path.join(ROOT, "signalalign_amd", "bin", "buildHdpUtil")
        tmp = path + ".tmp%d" % os.getpid()
        cmd = [tool, "-p", "14", "-v", tmp, "-l", asg, "-a", "6", "-n", "200", "-I", "20000", "-t", "100", "-s", "40", "-e", "140", "-k", "400",
               "--oneD", "-T", MODEL, "-B", "1", "-L", "1", "--seed", "7"]
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if pr.returncode != 0:
            raise RuntimeError("buildHdpUtil failed: " + pr.stderr[-400:])
        os.replace(tmp, path)
        rng = np.random.Generator(np.random.PCG64(0x44454e5345))
        with open(pool + ".tmp", "w") as f:
            f.write("".join("ACGT"[i] for i in rng.integers(0, 4, size=200000)) + "\n")
        os.replace(pool + ".tmp", pool)
    return path, pool


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks, one per GPU, as children of this process --
    which itself never touches the GPU (no HIP call, no torch.cuda call before or after) -- through
    torch.distributed.run on 127.0.0.1, relay rank 0's JSON line and exit with the launcher's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def lane_use_of_ring_regions(sa, pm, params, job_list, ambig, k, options_of):
    """Busy-lane fraction of the ring kernels on a sample of reads, from the plan's band geometry (sa_plan_describe: first and
    last cell of every diagonal) and the path counts of the reference windows: a diagonal of np cell-paths keeps ceil(np / 64)
    waves busy (waves beyond its last cell-path skip it), so the lanes that hold a cell-path are sum(np) / sum(64 ceil(np / 64))."""
    tot_np = tot_slots = tot_cells = 0
    for job in job_list:
        ref = job["ref"]
        lx = len(ref) - (k - 1)
        opt = np.array([options_of(c) for c in ref], dtype=np.int64)
        logp = np.concatenate([[0.0], np.cumsum(np.log2(opt))])
        paths = np.ones(lx + 2, dtype=np.int64)              # cell x = 0 is the NULL k-mer; x >= 1 holds window x-1 .. x+k-2
        paths[1:lx + 1] = np.rint(2.0 ** (logp[k:k + lx] - logp[0:lx])).astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(paths)])
        info, reg, rows, segs = sa.plan_describe(pm, params, job, ambig=ambig)
        d = np.arange(len(rows), dtype=np.int64)             # (one region per read here: diagonals 0 .. N in order)
        x0 = (d + rows[:, 1]) // 2
        w = (rows[:, 2] - rows[:, 1]) // 2 + 1
        np_d = cum[np.clip(x0 + w, 0, lx + 1)] - cum[np.clip(x0, 0, lx + 1)]
        tot_np += int(np_d.sum()); tot_slots += int((64 * ((np_d + 63) // 64)).sum()); tot_cells += int(w.sum())
    return {"busy_lane_fraction": tot_np / max(tot_slots, 1), "cell_paths_per_cell": tot_np / max(tot_cells, 1),
            "cell_paths_per_diagonal": tot_np / max(sum(len(j["ref"]) + len(j["events"]) for j in job_list), 1),
            "sample": "%d reads; band geometry from sa_plan_describe, path counts from the reference windows" % len(job_list)}


def measure(args, ctx, compact=False):
    """One workload through phase 1 (kernels on a resident batch) and phase 2 (fresh batches through the whole boundary);
    returns the result record on rank 0, None elsewhere.  compact: a secondary workload of the default run -- few steps, no
    serial-cycle samples, no CPU baseline."""
    import signalalign_amd as sa
    from signalalign_amd import synth
    dist, rank, world, device, backend = ctx["dist"], ctx["rank"], ctx["world"], ctx["device"], ctx["backend"]
    gold = os.path.join(ROOT, "tests", "golden", "models")
    model_path, nhdp, ambig, read_kw, wl_name = MODEL, None, None, {}, "BASELINE configs[1]: R9.4 6-mer Gaussian HMM"
    if args.workload == "cpg":
        model_path = os.path.join(gold, "testModelR9.4_450bps.cpg.6mer.template.model")
        ambig, read_kw = sa.default_ambig({"X": "CE"}), {"cpg_ambiguous": True}
        if args.cpg_every > 1:   # sparse variant positions: only every n-th CpG cytosine is ambiguous
            read_kw["cpg_every"] = int(args.cpg_every)
        wl_name = "BASELINE configs[2]: R9.4 6-mer CpG model (ACEGT), every CpG cytosine C/E"
        if args.cpg_every > 1:
            wl_name = ("R9.4 6-mer CpG model (ACEGT), every %d-th CpG cytosine ambiguous (C/E): sparse variant positions "
                       "(not a BASELINE config)" % args.cpg_every)
    elif args.workload == "scaling":
        wl_name = "BASELINE configs[4], one GPU's slice of 8 (R9.4 6-mer Gaussian HMM)"
    elif args.workload == "hdp_dense":
        nhdp, dense_pool = dense_nhdp()
        wl_name = "HDP emissions, dense .nhdp built by buildHdpUtil from the bundled assignments (3183 observed processes x 400 grid points)"
    elif args.workload in ("hdp", "hdp_cpg", "hdp_realistic"):
        model_path = os.path.join(gold, "testModelR73_acegot_template.model")
        nhdp = os.path.join(gold, "templateSingleLevelFixed.nhdp")
        wl_name = "BASELINE configs[3]: HDP emissions (templateSingleLevelFixed.nhdp, R7.3 ACEGOT)"
        if args.workload == "hdp_cpg":     # the reference's methylation-calling workflow: --sm3Hdp with variant positions
            ambig = sa.default_ambig({"X": "CE"})
            wl_name += ", every CpG cytosine C/E"
        if args.workload == "hdp_realistic":
            wl_name += ", anchors of a guide alignment"
    alpha, k, t10, tab = synth.parse_model_table(model_path)
    pm = sa.Model.load(model_path, nhdp)
    if nhdp:
        pm.set_to_hdp_expected_values()
    params = sa.default_params(threshold=args.threshold, expansion=50, trace_back=100)
    spec = dict(kind="gauss", model=model_path, events=args.events, kw=read_kw)
    if nhdp:
        # events drawn from the densities the aligner itself uses, over windows of the sequence the bundled .nhdp was trained
        # on (synth.make_read_hdp); the table's level means (after set_to_hdp_expected_values) enter the event normalisation
        spec = dict(kind="hdp", model=model_path, nhdp=nhdp, events=args.events, table5=np.array(pm.table5()),
                    ref_pool=dense_pool if args.workload == "hdp_dense" else os.path.join(ROOT, "tests", "golden", "npReads", "ZymoRef.txt"))
    # spawned numpy-only workers; identical to the serial loop (several ranks on one host share its CPUs: fewer workers each)
    gen_workers = None if world == 1 else max(1, min(4, int(os.environ.get("SA_HOST_THREADS", "2"))))
    def mark_cpg(job_list):   # hdp_cpg: the C of every CG becomes X (the reads' references come from a pool of real sequence)
        for job in job_list:
            job["ref"] = job["ref"].replace("CG", "XG")

    def make_many(idx):
        idx = [int(i) for i in idx]
        key = (args.workload, args.events, tuple(idx[:2]), idx[-1] if idx else -1, len(idx))
        memo = ctx.setdefault("reads_memo", {})
        if memo.get("workload") != args.workload:    # (one workload's sets at a time: a 10k-event slice is 4 GB)
            memo.clear()
            memo["workload"] = args.workload
        if key not in memo:
            memo[key] = synth.make_reads_parallel(spec, idx, workers=gen_workers)
        return [dict(j_) for j_ in memo[key]]        # (callers thin anchors / narrow events in their copies)
    # reads are independent: the global read list is dealt to the ranks (no collective on the data path)
    from signalalign_amd import shard
    mine = shard.shard_indices([args.events] * (world * args.reads), rank, world)
    jobs = make_many(mine)
    if args.workload == "hdp_cpg":
        mark_cpg(jobs)
    def thin_like_a_guide_alignment(job_list, indices):
        # the anchors a real guide alignment leaves: the run structure of the reference's own example cigar (an indel every
        # 10-50 bases), 14 bases trimmed off both ends of every match run as signalMachine -m 14 does
        toks = open(os.path.join(ROOT, "tests", "golden", "cigars", "ecoli_minus_strand.cigar")).read().split()[10:]
        runs = [(toks[i], int(toks[i + 1])) for i in range(0, len(toks), 2)]
        for idx, job in zip(indices, job_list):
            keep = np.zeros(len(job["ax"]), dtype=bool)
            pos, r = 0, (7 * int(idx)) % len(runs)
            while pos < len(keep):
                op, ln = runs[r % len(runs)]
                r += 1
                if op == "M":
                    if ln > 28:
                        keep[pos + 14: min(pos + ln - 14, len(keep))] = True
                    pos += ln
                elif op == "D":
                    pos += ln
            job["ax"], job["ay"] = job["ax"][keep], job["ay"][keep]

    if args.workload in ("realistic", "hdp_realistic"):
        thin_like_a_guide_alignment(jobs, [int(i) for i in mine])
    if args.workload == "realistic":
        wl_name = ("BASELINE configs[1] reads with the anchor density of a real guide alignment "
                   "(tests/golden/cigars/ecoli_minus_strand.cigar, -m 14: a sixth of the bases)")
    # ---- read sets: every timed step aligns reads the library has not seen in the step before ----
    # (a compact secondary leg of BASELINE size -- 10 000 reads with several paths per cell, 5000 HDP reads, 12 500 10k-event
    # reads -- cycles ONE read set: generating a second one costs more than the leg's steps, and the library keeps nothing of a
    # batch's inputs between batches: every step checks, packs, uploads and plans them again)
    big_compact = compact and args.reads * args.events >= 2.5e7
    n_sets = 1 if args.kernels_only or big_compact else (2 if args.workload == "scaling" or args.reads > 4000 or compact else 3)
    sets = [jobs]
    for q in range(1, n_sets):
        more = shard.shard_indices([args.events] * (world * args.reads), rank, world)
        extra = make_many([int(i) + q * world * args.reads for i in more])
        if args.workload in ("realistic", "hdp_realistic"):
            thin_like_a_guide_alignment(extra, [int(i) + q * world * args.reads for i in more])
        if args.workload == "hdp_cpg":
            mark_cpg(extra)
        sets.append(extra)
    if args.event_stride == 1:
        for js in sets:
            for j_ in js:
                if np.asarray(j_["events"]).ndim == 2:
                    j_["events"] = np.ascontiguousarray(np.asarray(j_["events"])[:, 0])
    # marshalled once: a C caller holds sa_job_t arrays anyway.  --inputs host-block: the event records and anchors live in one
    # page-locked block per read set (sa_host_alloc), as a caller that reads its inputs into such a block has them
    # auto: host-block when this rank has few host threads to pack pageable inputs with (ranks of an 8-GPU run under a 16-CPU
    # quota get two: 16.4 against 11.0 ms per step) and the block is of moderate size; pageable otherwise (faster
    # when host threads are plentiful: a third of the bytes cross PCIe)
    from signalalign_amd._capi import jobs_bytes_in_block
    host_threads = int(os.environ.get("SA_HOST_THREADS") or 0)
    in_block = args.inputs == "host-block" or (args.inputs == "auto" and 0 < host_threads <= 3 and
                                               max(jobs_bytes_in_block(js) for js in sets) <= (2 << 30))
    try:
        arrays = [sa.JobArray(js, host_block=in_block) for js in sets]
    except sa.SaError:
        if args.inputs != "auto":
            raise
        in_block = False
        arrays = [sa.JobArray(js) for js in sets]
    xflags = sa.FLAG_INPUTS_IN_HOST_BLOCK if in_block else 0
    if getattr(args, "pairs8", False):   # 8-byte result records (x, y, probability): one path per cell only
        xflags |= sa.FLAG_PAIRS8
    inputs_used = "host-block" if in_block else "pageable"
    n_events_total = sum(len(j["events"]) for j in jobs)

    def sync():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # ---- phase 1 (not the headline): the kernels alone, on a batch whose plan and inputs are resident in HBM ----
    t_create = time.perf_counter()
    batch = sa.Batch(pm, params, arrays[0], ambig=ambig, device=device, flags=xflags)
    t_create = time.perf_counter() - t_create
    st0 = batch.stats()
    cells = st0.cells_forward + st0.cells_backward
    KR = max(3, min(args.steps, 10))
    for _ in range(max(1, min(args.warmup, 3))):
        batch.run()
    t0 = time.perf_counter()
    ms_f = ms_b = ms_fold = 0.0
    for _ in range(KR):
        batch.run()
        s_ = batch.stats()
        ms_f += s_.ms_forward
        ms_b += s_.ms_backward
        ms_fold += s_.ms_fold
    dt_resident = (time.perf_counter() - t0) / KR
    ms_f, ms_b, ms_fold = ms_f / KR, ms_b / KR, ms_fold / KR
    n_pairs = int(batch.results_view()[1][-1])
    # what a caller pays who wants sa_pair_t rows instead of the packed records the step ends at: sa_batch_pairs_all (host threads)
    unpack_ms = None
    if not compact and 0 < n_pairs < 5e7:
        from signalalign_amd._capi import PAIR_DTYPE
        rows_buf = np.empty(n_pairs, dtype=PAIR_DTYPE)
        t_u = []
        for _ in range(4):
            tu0 = time.perf_counter()
            batch.pairs_all(rows_buf)
            t_u.append((time.perf_counter() - tu0) * 1e3)
        unpack_ms = {"first_call_with_page_faults": t_u[0], "steady": min(t_u[1:])}
        del rows_buf
    batch.close()
    # The roofline's kernel times: the same resident batch planned with ONE backward launch per forward-storage pass
    # (SA_GROUPS=1), so that a stage is one launch of the dominant kernel alone on its stream, timed by the library's HIP
    # events on that stream -- and `rocprofv3 --kernel-trace --stats` over `bench.py --workload W --kernels-only` under
    # SA_GROUPS=1 (profiles/r04_W_kernel_stats.csv) shows the same average duration.  (The eight result groups of the phase
    # above overlap pairwise on two streams; their stage time also holds fold / finalisation kernels of earlier groups.)
    single = None
    if int(st0.n_groups) > int(st0.n_chunks) and not os.environ.get("SA_GROUPS"):
        os.environ["SA_GROUPS"] = "1"
        try:
            b1 = sa.Batch(pm, params, arrays[0], ambig=ambig, device=device, flags=xflags)
        finally:
            del os.environ["SA_GROUPS"]
        s1 = b1.stats()
        b1.run()
        K1 = 3 if compact else 5
        f1 = bk1 = 0.0
        for _ in range(K1):
            b1.run()
            s_ = b1.stats()
            f1 += s_.ms_forward
            bk1 += s_.ms_backward
        single = {"ms_forward": f1 / K1, "ms_backward": bk1 / K1, "launches_per_pass": int(s1.n_groups) // max(int(s1.n_chunks), 1),
                  "passes": int(s1.n_chunks)}
        b1.close()
    elif int(st0.n_groups) == int(st0.n_chunks):
        single = {"ms_forward": ms_f, "ms_backward": ms_b, "launches_per_pass": 1, "passes": int(st0.n_chunks)}

    # serial cycles (median of five, outside the timed region): what a caller without overlap pays per batch in steady state
    cycle = {"create": 0.0, "run": dt_resident * 1e3, "destroy": 0.0}
    if not args.kernels_only:
        samples = []
        for q in range(1 if args.workload == "scaling" or compact else 5):
            tc0 = time.perf_counter()
            bb = sa.Batch(pm, params, arrays[(q + 1) % n_sets], ambig=ambig, device=device, flags=xflags)
            tc1 = time.perf_counter()
            bb.run()
            tc2 = time.perf_counter()
            bb.close()
            tc3 = time.perf_counter()
            samples.append(((tc1 - tc0) * 1e3, (tc2 - tc1) * 1e3, (tc3 - tc2) * 1e3))
        med = sorted(samples, key=lambda t_: sum(t_))[len(samples) // 2]
        cycle = {"create": med[0], "run": med[1], "destroy": med[2]}

    # ---- phase 2 (the headline): one step = one batch of FRESH reads through the whole boundary -- sa_batch_create (checks,
    # planning, upload), run (forward, backward/posterior, fold, finalisation, result copy to the host), results read,
    # sa_batch_destroy -- with the next batch being created while the current one is on the GPU (sa_batch_start/wait) ----
    # batches in flight: as many as asked for, as long as their forward storage (24 B per band cell, the bulk of a batch)
    # fits HBM together with room to spare -- a 10k-event slice or 10k reads with several paths per cell take half of
    # the card alone and go one at a time
    # (a rank's share of the card: the whole of it on a real node; in a rehearsal with several ranks on one GPU -- gloo backend,
    # device = local_rank % device_count -- the ranks must not each size a pipeline for all of it)
    hbm_bytes = float(sa.device_memory(device)[1]) / max(1, ctx.get("ranks_per_device", 1))
    if ctx.get("ranks_per_device", 1) > 1:
        sa.pool_configure(device_limit_bytes=int(0.5 * hbm_bytes))   # what may stay parked between batches: this rank's share too
    depth = max(1, min(args.in_flight, int(0.6 * hbm_bytes / max(1.25 * st0.f_bytes, 1.0)),
                       int(0.8 * hbm_bytes / max(st0.device_bytes, 1.0))))   # (all working storage: candidate slots weigh as much as the planes with HDP models)
    if args.workload == "scaling":
        depth = 1
    # ... and only while a batch's pairs are a small part of the traffic: the HDP workload returns 4.3 GB of pairs per batch
    # (77 ms of PCIe against 28 ms of kernels) and is faster one batch at a time (155 against 220 ms per step with three in
    # flight: the result copies of the neighbours queue behind each other on the one copy engine)
    depth = max(1, min(depth, int(6e9 / max(24.0 * n_pairs, 1.0))))
    cells_done = [0.0]
    done_at = []          # completion time of every batch (results in the caller's hands): per-step times for the median beside the mean
    groups_seen = [0]
    pairs_seen = [0]
    first_buf = np.zeros(len(jobs) + 1, dtype=np.int64)

    carry = {}   # one batch at a time: the batch whose first half was made during the previous call's last step
    step_log = []   # pipelined loop: (step ms, create ms, wait ms) of every step
    wait_ms = [0.0]

    def step_stats(log):
        """p10 / p50 / p90 / max of the steps of a run of the pipelined loop and, for the slowest three, which stage stretched"""
        if len(log) < 5:
            return None
        tot = sorted(x[0] for x in log)
        q = lambda f: tot[min(len(tot) - 1, int(f * len(tot)))]
        med_c = sorted(x[1] for x in log)[len(log) // 2]
        med_w = sorted(x[2] for x in log)[len(log) // 2]
        slow = sorted(log, key=lambda x: -x[0])[:3]
        return {"p10": round(q(0.1), 3), "p50": round(q(0.5), 3), "p90": round(q(0.9), 3), "max": round(tot[-1], 3),
                "median_create_ms": round(med_c, 3), "median_wait_ms": round(med_w, 3),
                "slowest": [{"ms": round(x[0], 2), "create_ms": round(x[1], 2), "wait_ms": round(x[2], 2)} for x in slow]}

    def stream(n_steps, first, leave_next=False):
        flying = []
        if depth == 1:
            # one batch at a time on the device (its forward storage takes most of the card): the next batch's reads are
            # checked, packed, uploaded and planned (sa_batch_create_deferred) while the current one runs; the rest of its
            # creation -- working buffers sized for the device to itself -- follows when the current one has been destroyed
            def make(s_):
                return sa.Batch(pm, params, arrays[(first + s_) % n_sets], ambig=ambig, device=device, deferred=True,
                                flags=sa.FLAG_DEVICE_TO_ITSELF | xflags)
            # (steady state across calls: with leave_next the last step prepares the batch the NEXT call starts with, so that every
            # call of K steps holds K first halves and K runs -- the timed region neither gets one for free nor pays one alone)
            nxt = carry.pop("nxt", None)
            if nxt is None and n_steps > 0:
                nxt = make(0)
            dbg = os.environ.get("SA_BENCH_DEBUG")
            for s in range(n_steps):
                cur = nxt
                t_a = time.perf_counter()
                cur.start()
                nxt = make(s + 1) if (s + 1 < n_steps or leave_next) else None
                if nxt is not None and not os.environ.get("SA_BENCH_NO_PREPARE"):
                    nxt.prepare()   # (its plan and launch lists while `cur` runs: sa_batch_prepare)
                t_b = time.perf_counter()
                cur.wait()
                t_c = time.perf_counter()
                stc = cur.stats()
                cells_done[0] += stc.cells_forward + stc.cells_backward
                groups_seen[0] = int(stc.n_groups)
                pairs_seen[0] += int(cur.results_view(first_buf)[1][-1])
                done_at.append(time.perf_counter())
                cur.close()
                if dbg:
                    print("[bench] step %d: next batch's first half %.1f ms, then waited %.1f ms, collect %.1f ms; device %.1f ms"
                          % (s, (t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (time.perf_counter() - t_c) * 1e3, stc.ms_total_device),
                          file=sys.stderr)
            if nxt is not None:
                carry["nxt"] = nxt
            return
        dbg = os.environ.get("SA_BENCH_DEBUG")
        defer = bool(os.environ.get("SA_BENCH_DEFER"))   # (experiment hook)

        def retire(old):
            t_w = time.perf_counter()
            old.wait()
            wait_ms[0] = (time.perf_counter() - t_w) * 1e3
            if defer:
                stc = old.stats()
                cells_done[0] += stc.cells_forward + stc.cells_backward
                groups_seen[0] = int(stc.n_groups)
            # the step ends where the results are the caller's: every job's packed 16-byte records, in place in the batch's
            # pinned block (sa_batch_pairs16_all: one call, nothing copied)
            pairs_seen[0] += int(old.results_view(first_buf)[1][-1])
            done_at.append(time.perf_counter())
            old.close()

        for s in range(n_steps):
            t_a = time.perf_counter()
            # (creation in one piece, also with host-block inputs: created in two halves the batches start in pairs, their
            # block transfers queue behind each other and their kernels share the chip -- 19.6 against 12 ms per step)
            cur = sa.Batch(pm, params, arrays[(first + s) % n_sets], ambig=ambig, device=device, flags=xflags, deferred=defer)
            if not defer:
                stc = cur.stats()
                cells_done[0] += stc.cells_forward + stc.cells_backward
                groups_seen[0] = int(stc.n_groups)
            t_b = time.perf_counter()
            cur.start()
            flying.append(cur)
            wait_ms[0] = 0.0
            if len(flying) >= depth:
                retire(flying.pop(0))
            # per step: whole step, sa_batch_create (host fan-out: checks, packing, upload, device planner), the wait for the oldest
            # batch in flight; the rest is start + results in place + destroy
            step_log.append(((time.perf_counter() - t_a) * 1e3, (t_b - t_a) * 1e3, wait_ms[0]))
            if dbg:
                thr = open("/sys/fs/cgroup/cpu.stat").read().split() if os.path.exists("/sys/fs/cgroup/cpu.stat") else []
                nthr = thr[thr.index("nr_throttled") + 1] if "nr_throttled" in thr else "?"
                print("[bench] step %d: create %.1f ms, step %.1f ms, cgroup nr_throttled %s"
                      % (s, (t_b - t_a) * 1e3, (time.perf_counter() - t_a) * 1e3, nthr), file=sys.stderr)
        for old in flying:
            retire(old)

    steps_timed = None
    if args.kernels_only:
        dt, cells_done[0] = dt_resident * args.steps, cells * args.steps
    else:
        # Setup, before the W warm-up steps: the library's caching allocators start empty and hand out a block of a new size
        # only after a hipMalloc / hipHostMalloc (45 ms for a batch's pinned result block).  A pipeline `depth` deep needs depth + 1
        # sets of blocks; the phases above had one batch alive at a time.  Whatever W the caller passes, the allocators have
        # seen the pipeline's working set before the first warm-up step (a long-running aligner is in that state for good).
        priming = depth + 2 if depth > 1 else 0
        stream(priming, 0)
        stream(args.warmup, priming, leave_next=args.warmup > 0)
        sync()
        cells_done[0] = 0.0
        pairs_seen[0] = 0
        del done_at[:]
        del step_log[:]
        t0 = time.perf_counter()
        stream(args.steps, priming + args.warmup, leave_next=args.warmup > 0)
        sync()
        dt = time.perf_counter() - t0
        steps_timed = step_stats(step_log)
        if carry.get("nxt") is not None:
            carry.pop("nxt").close()
    cells_streamed = cells_done[0]
    pairs_timed = pairs_seen[0]   # (of the K timed steps; the long run below counts on)
    # Beside the mean (which is what `value` is by contract): for workloads that run one batch at a time -- where every step waits
    # for its batch -- the median step: a stall of the host (a page-locked allocation, a throttled cgroup) moves the mean, not the
    # median.  (With several batches in flight completions are observed in bursts; the 200-step long run is the steadier figure.)
    median_step_ms = None
    if not args.kernels_only and depth == 1 and len(done_at) >= 3:
        gaps = sorted(b_ - a_ for a_, b_ in zip(done_at[:args.steps - 1], done_at[1:args.steps]))
        median_step_ms = gaps[len(gaps) // 2] * 1e3
    long_run = None
    if (not compact and not args.kernels_only and world == 1 and args.workload == "gaussian" and not args.no_secondary
            and args.long_steps > args.steps):
        # K = 20 steps are 0.2 s: a longer sample of the same loop beside it (NOT `value`: the contract times exactly K steps)
        cells_done[0] = 0.0
        del step_log[:]
        tl0 = time.perf_counter()
        stream(args.long_steps, priming + args.warmup + args.steps)
        dtl = time.perf_counter() - tl0
        long_run = {"steps": args.long_steps, "seconds": dtl, "ms_per_step": dtl / args.long_steps * 1e3,
                    "value": cells_done[0] / dtl, "step_ms": step_stats(step_log),
                    "note": "same pipelined loop as the timed steps, run once more for longer"}
    if dist is not None:
        import torch
        tdev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([cells_streamed, float(n_events_total) * args.steps], dtype=torch.float64, device=tdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        cells_all, events_all = float(tot[0].item()), float(tot[1].item())
    else:
        cells_all, events_all = cells_streamed, float(n_events_total) * args.steps

    if rank == 0:
        K = args.steps
        # dominant kernel of rank 0, timed with HIP events on the library's own streams.  k_bwd_fast runs as
        # n_groups launches per step that overlap pairwise on two streams: its duration here is
        # the wall time of the whole traceback stage (end of forward -> end of the last k_bwd_fast), which also
        # contains the fold/finalisation kernels of the earlier groups, and the bytes are those of all launches.
        fam = "fast" if st0.n_fast_regions == st0.n_regions else ("ring" if st0.n_ring_regions > 0 else "generic")
        if fam == "ring" and 2 * st0.n_strip_regions > st0.n_ring_regions:
            fam = "strip"
        sfx = "_hdp" if (args.workload.startswith("hdp") and fam == "fast") else ""
        rf_f, rf_b = (single["ms_forward"], single["ms_backward"]) if single else (ms_f, ms_b)
        if ms_b >= ms_f:
            dom, dom_ms, dom_cells = "k_bwd_" + fam + sfx, rf_b, st0.cells_backward
            if fam == "strip" and os.environ.get("SA_STRIP_PASSES") != "2":
                dom = "k_bwd_strip1"   # the one-pass sweep (round 4; SA_STRIP_PASSES=2: k_bwd_strip)
            dom_parts = [dom]
        else:
            dom, dom_ms, dom_cells = "k_fwd_" + fam + sfx, rf_f, st0.cells_forward
            dom_parts = [dom]
            if sfx:   # the HDP forward stage is two kernels: the emission plane (k_emit_hdp), then the sweep that reads it
                dom_parts = ["k_emit_hdp", dom]
                dom = "k_emit_hdp + " + dom
        achieved = ALGO_BYTES_PER_CELL * dom_cells / (dom_ms * 1e-3) / 1e9
        # HBM bytes per step of the dominant kernel from the rocprofv3 --pmc passes (profiles/traffic.json, written by
        # probes/profile_r02.sh + probes/traffic_from_pmc.py: FETCH_SIZE and WRITE_SIZE in separate passes, fetch doubled as
        # MI355X_MICROARCH.md prescribes for gfx950); collected at the default size of each workload only
        traffic = None
        profiles_meta = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        default_size = (args.reads == {"scaling": 12500, "hdp": 5000, "hdp_dense": 5000, "cpg": 10000}.get(args.workload, 2000) and   # (sizes of the counter passes)
                        args.events == (10000 if args.workload == "scaling" else 5000))
        if os.path.exists(tp) and default_size:
            try:
                tj = json.load(open(tp))
                tparts = [tj.get(args.workload, {}).get(q, {}).get("bytes_per_step") for q in dom_parts]
                traffic = sum(tparts) if all(tparts) else None
                profiles_meta = tj.get("_meta", {}).get(args.workload)   # commit and date of the counter passes (not this run)
            except Exception:
                traffic = None
        # instruction issue of the dominant kernel: wave-instructions per step from the rocprofv3 --pmc passes
        # (profiles/instr_mix.json, probes/profile_final.sh + probes/instr_from_pmc.py) over the stage time measured here.
        # Peak: one fp64 VALU instruction per SIMD every 1.83 ns (probes/issue_probe.hip on this part) x 1024 SIMDs.
        issue = None
        ip = os.path.join(ROOT, "profiles", "instr_mix.json")
        if os.path.exists(ip) and default_size:
            try:
                recs = [json.load(open(ip)).get(args.workload, {}).get(q) for q in dom_parts]
                rec = None
                if all(recs):
                    rec = dict(recs[-1])
                    for kk in ("valu_per_step", "instructions_per_step"):
                        rec[kk] = sum(r_.get(kk) or 0.0 for r_ in recs)
            except Exception:
                rec = None
            if rec and rec.get("valu_per_step"):
                valu_rate = rec["valu_per_step"] / (dom_ms * 1e-3)
                issue = {"kernel": dom, "valu_per_step": rec["valu_per_step"],
                         "instructions_per_step": rec.get("instructions_per_step"),
                         "achieved": valu_rate, "peak": ISSUE_PEAK_VALU_PER_S, "unit": "fp64 VALU wave-instructions/s",
                         "frac": valu_rate / ISSUE_PEAK_VALU_PER_S,
                         "all_instructions_per_s": (rec.get("instructions_per_step") or 0.0) / (dom_ms * 1e-3),
                         "wave_wait_frac": rec.get("SQ_WAIT_ANY_frac_of_wave_cycles"),
                         "source": "profiles/instr_mix.json (SQ_INSTS_* per kernel, rocprofv3 --pmc over bench.py --kernels-only)"}
        lane_use = None
        if ambig is not None and st0.n_ring_regions > st0.n_strip_regions:
            try:
                amb_opts = {"X": 2}
                lane_use = lane_use_of_ring_regions(sa, pm, params, jobs[:8], ambig, k, lambda c: amb_opts.get(c, 1))
            except Exception as ex:   # (never the line's problem)
                lane_use = {"failed": "%s: %s" % (type(ex).__name__, ex)}
        out = {
            "metric": "dp_cell_updates_per_s",
            "value": cells_all / dt,
            "unit": "cell_updates/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3,
            "median_ms_per_step_one_batch_at_a_time": median_step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s, %d x %d-event synthetic reads per GPU, band 50, threshold %g" % (wl_name, args.reads, args.events, args.threshold),
                "reads_per_gpu": args.reads, "events_per_read": args.events, "event_stride": args.event_stride,
                "inputs": inputs_used if args.inputs != "auto" else inputs_used + " (auto: host-block when SA_HOST_THREADS <= 3)",
                "host_threads": os.environ.get("SA_HOST_THREADS"),
                "events_per_s": events_all / dt,
                "cells_per_event": cells / max(n_events_total, 1),
                "pairs_rank0": n_pairs, "pairs_per_event": n_pairs / max(n_events_total, 1),
                "regions_on_register_kernels": "%d/%d" % (st0.n_fast_regions, st0.n_regions),
                "regions_on_ring_kernels": "%d/%d" % (st0.n_ring_regions - st0.n_strip_regions, st0.n_regions),
                "regions_on_strip_kernels": "%d/%d" % (st0.n_strip_regions, st0.n_regions),
                "ring_kernels_lane_use": lane_use,
                "forward_storage_passes": int(st0.n_chunks),
                "result_groups": {"resident_batch_phase": int(st0.n_groups), "timed_pipeline": groups_seen[0] or None},
                "step": "one batch of fresh reads through the whole boundary: sa_batch_create (checks, planning, upload) + run + "
                        "results on the host + sa_batch_destroy; %s" % ("%d batches in flight (sa_batch_start / sa_batch_wait)" % depth
                                                                        if depth > 1 else "one batch on the device at a time, the next one checked, "
                                                                        "packed, uploaded and planned meanwhile (sa_batch_create_deferred)"),
                "result_record_bytes": 8 if getattr(args, "pairs8", False) else 16,
                "results": {"step_ends_at": "every job's pairs as packed 16-byte records (sa_pair16_t, include/signalalign_hip.h; 8-byte sa_pair8_t with --pairs8) in "
                                            "the batch's pinned host block, read in place through sa_batch_pairs16_all",
                            "pairs_seen_in_timed_steps": pairs_timed,
                            "expand_to_sa_pair_t_ms": unpack_ms,
                            "note": "sa_batch_pairs_all (24-byte sa_pair_t rows, host threads) is NOT inside the timed step; "
                                    "its cost per batch is expand_to_sa_pair_t_ms"},
                "read_sets_cycled": n_sets, "batches_in_flight": depth, "long_run": long_run, "step_ms": steps_timed,
                "allocator_priming_batches_before_warmup": (depth + 2 if depth > 1 else 0) if not args.kernels_only else 0,
                "first_batch_create_s": t_create,
                "serial_cycle_ms": cycle,
                "value_serial_cycle": cells / (sum(cycle.values()) * 1e-3),
                "kernels_only_resident_inputs": {"value": cells / dt_resident, "ms_per_step": dt_resident * 1e3,
                                                 "note": "sa_batch_run repeated on one planned, HBM-resident batch "
                                                         "(the round-1 headline)"},
                "kernel_ms": {"forward": ms_f, "backward_posterior": ms_b, "fold_and_finalize": ms_fold},
                "kernel_cell_updates_per_s": {"forward": st0.cells_forward / (ms_f * 1e-3),
                                              "backward_posterior": st0.cells_backward / (ms_b * 1e-3)},
            },
            # `frac` follows SURVEY section 8(d): algorithmic bytes (24 B per cell update) over the stage time against HBM peak;
            # `frac_by_counters` are the bytes the counters saw.  `bound` names what the evidence says limits the kernel: the
            # sweeps issue instructions during most of their cycles while HBM idles ("issue"; HBM figures kept beside it)
            "roofline": {"bound": "issue" if issue and issue["frac"] > (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else 0.0)
                         else "hbm", "bound_of_the_formula": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_over_algorithmic": (traffic / (ALGO_BYTES_PER_CELL * dom_cells)) if traffic else None,
                         "frac_by_counters": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "limiter": "instruction issue (VALU + SALU of the serial per-diagonal chain), not HBM: see "
                                    "profiles/ and DESIGN.md section 4",
                         "counters_collected_at": profiles_meta,
                         "kernel_passes_phase1": (max(1, min(args.warmup, 3)) + KR),
                         "algorithmic_bytes_per_step": ALGO_BYTES_PER_CELL * dom_cells,
                         "stage_ms": dom_ms,
                         "stage_ms_is": ("the dominant kernel's launches of one step, each alone on its stream (resident batch planned "
                                         "with SA_GROUPS=1), HIP events on that stream: launches_per_step launches, one per "
                                         "forward-storage pass" if single else "stage wall of the grouped launches"),
                         "stage_ms_of_the_grouped_launches": ms_b if dom.startswith("k_bwd") else ms_f,
                         "launches_per_step": (single["passes"] * single["launches_per_pass"] if single else
                                               (int(st0.n_groups) if dom.startswith("k_bwd") else int(st0.n_chunks))),
                         "avg_launch_ms": dom_ms / max(1, (single["passes"] * single["launches_per_pass"]) if single else
                                                       (int(st0.n_groups) if dom.startswith("k_bwd") else int(st0.n_chunks)))},
        }
        if issue:
            out["issue_roofline"] = issue
        if not args.no_cpu_baseline and world == 1:   # (rank 0 at N = 1 only)
            # the CPU restatement on a bounded sample of THIS workload: same generator, model (HDP loaded where the workload has
            # one), ambiguity table, threshold and anchors
            from oracle import sa_oracle_py as oracle          # timed here as the baseline, nothing else
            om = oracle.Model(alpha, k, t10, tab)
            if nhdp:
                om.load_hdp(nhdp)
                om.set_to_hdp_expected_values()
            op_ = oracle.default_params(threshold=args.threshold, expansion=50, trace_back=100)

            def make_sample(n_, first_):
                idx_ = list(range(first_, first_ + n_))
                js_ = synth.make_reads_parallel(spec, idx_, workers=1)
                if args.workload in ("realistic", "hdp_realistic"):
                    thin_like_a_guide_alignment(js_, idx_)
                if args.workload == "hdp_cpg":
                    mark_cpg(js_)
                return js_
            out["cpu_baseline"] = cpu_baseline(om, op_, make_sample, args.events, args.cpu_reads_per_thread, 10 ** 6,
                                               ambig=oracle.ambig_map({"X": "CE"}) if args.workload in ("cpg", "hdp_cpg") else None,
                                               what={"cpg": ", every CpG cytosine C/E", "hdp": ", HDP emissions", "hdp_dense": ", HDP emissions (dense .nhdp)",
                                                     "hdp_cpg": ", HDP emissions, every CpG cytosine C/E",
                                                     "hdp_realistic": ", HDP emissions, anchors of a guide alignment",
                                                     "realistic": ", anchors of a guide alignment"}.get(args.workload, ""))
        return out
    return None


def scaling_job(args, ctx):
    """BASELINE configs[4] as a JOB (strong scaling): `--job-reads` (100 000) synthetic `--job-events` (10 000)-event R9.4 reads,
    dealt to the ranks by signalalign_amd/shard.py (the one-process-per-read pool of src/signalalign/signalAlignment.py:694-737
    is what this replaces), every rank aligning ITS share in slices of `--job-slice` reads through the pipelined boundary
    (sa_batch_create of the next slices while earlier ones are on the GPU, as the timed loop of the headline does).  No collective
    on the data path; the barrier and the max-over-ranks of the wall time are the only exchanges.  value = cell updates of ALL
    ranks / the slowest rank's wall time.  Reads are generated for the first `job_sets` slices of a rank only and cycled (the
    library keeps nothing of a slice between slices: every slice is checked, packed, uploaded and planned anew; 100 000 reads
    would be 32 GB of host arrays and minutes of generation) -- the line says so.  Returns the record on rank 0."""
    import signalalign_amd as sa
    from signalalign_amd import synth, shard
    from signalalign_amd._capi import jobs_bytes_in_block
    dist, rank, world, device, backend = ctx["dist"], ctx["rank"], ctx["world"], ctx["device"], ctx["backend"]
    total, ev, sl = int(args.job_reads), int(args.job_events), int(args.job_slice)
    if ctx.get("ranks_per_device", 1) > 1:   # (rehearsal: several ranks share one card -- every rank's working set must fit its share)
        sl = max(250, sl // int(ctx["ranks_per_device"]))
    alpha, k, t10, tab = synth.parse_model_table(MODEL)
    pm = sa.Model.load(MODEL)
    params = sa.default_params(threshold=0.01, expansion=50, trace_back=100)
    spec = dict(kind="gauss", model=MODEL, events=ev, kw={})
    t_part = time.perf_counter()
    mine = shard.shard_indices([ev] * total, rank, world)
    t_part = time.perf_counter() - t_part
    n_mine = len(mine)
    sizes = shard.slice_sizes(n_mine, sl)
    n_slices = len(sizes)
    n_sets = max(1, min(int(args.job_sets), n_slices))
    gen_workers = None if world == 1 else max(1, min(4, int(os.environ.get("SA_HOST_THREADS", "2"))))
    memo = ctx.setdefault("reads_memo", {})
    sets = []
    t_gen = time.perf_counter()
    starts = [0]
    for z in sizes:
        starts.append(starts[-1] + z)
    for q in range(n_sets):
        idx = [int(i) for i in mine[starts[q]:starts[q + 1]]]
        key = ("scaling", ev, tuple(idx[:2]), idx[-1] if idx else -1, len(idx))   # (the scaling_slice leg's reads, when they are the same)
        if memo.get("workload") != "scaling":
            memo.clear()
            memo["workload"] = "scaling"
        if key not in memo:
            memo[key] = synth.make_reads_parallel(spec, idx, workers=gen_workers)
        sets.append(memo[key])
    t_gen = time.perf_counter() - t_gen
    host_threads = int(os.environ.get("SA_HOST_THREADS") or 0)
    in_block = args.inputs == "host-block" or (args.inputs == "auto" and 0 < host_threads <= 3 and
                                               max(jobs_bytes_in_block(js) for js in sets) <= (2 << 30))
    try:
        arrays = [sa.JobArray(js, host_block=in_block) for js in sets]
    except sa.SaError:
        in_block = False
        arrays = [sa.JobArray(js) for js in sets]
    xflags = sa.FLAG_INPUTS_IN_HOST_BLOCK if in_block else 0
    partial = {}   # a last slice shorter than the others: the first reads of its set

    def array_of(q):
        if sizes[q] == len(sets[q % n_sets]):
            return arrays[q % n_sets]
        if sizes[q] not in partial:
            partial[sizes[q]] = sa.JobArray(sets[q % n_sets][:sizes[q]], host_block=in_block)
        return partial[sizes[q]]

    def sync():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # pipeline depth from a slice's own storage, as the headline loop sizes it (this rank's share of the card in a rehearsal)
    probe = sa.Batch(pm, params, arrays[0], ambig=None, device=device, flags=xflags)
    st0 = probe.stats()
    probe.run()
    probe.close()
    hbm_bytes = float(sa.device_memory(device)[1]) / max(1, ctx.get("ranks_per_device", 1))
    if ctx.get("ranks_per_device", 1) > 1:
        sa.pool_configure(device_limit_bytes=int(0.9 * hbm_bytes))   # (a slice's blocks stay parked for the next slice of this rank)
    depth = max(1, min(args.in_flight, int(0.6 * hbm_bytes / max(1.25 * st0.f_bytes, 1.0)), int(0.8 * hbm_bytes / max(st0.device_bytes, 1.0))))
    cells_done, events_done, pairs_done, reads_done = [0.0], [0.0], [0], [0]
    first_buf = np.zeros(sl + 1, dtype=np.int64)

    def run_slices(qs):
        flying = []

        def retire(item):
            b, q = item
            b.wait()
            pairs_done[0] += int(b.results_view(first_buf)[1][sizes[q]])
            b.close()
        if depth == 1 and ctx.get("ranks_per_device", 1) > 1:
            # a rehearsal with several ranks on one card (gloo): no rank has the device to itself -- one slice after the other,
            # each planned into what is free at that moment
            for q in qs:
                cur = sa.Batch(pm, params, array_of(q), ambig=None, device=device, flags=xflags)
                stc = cur.stats()
                cur.run()
                cells_done[0] += stc.cells_forward + stc.cells_backward
                pairs_done[0] += int(cur.results_view(first_buf)[1][sizes[q]])
                events_done[0] += float(sizes[q]) * ev
                reads_done[0] += sizes[q]
                cur.close()
            return
        if depth == 1:
            def make(q):
                return sa.Batch(pm, params, array_of(q), ambig=None, device=device, deferred=True, flags=sa.FLAG_DEVICE_TO_ITSELF | xflags)
            nxt = make(qs[0]) if qs else None
            for i, q in enumerate(qs):
                cur = nxt
                cur.start()
                nxt = make(qs[i + 1]) if i + 1 < len(qs) else None
                if nxt is not None:
                    nxt.prepare()
                cur.wait()
                stc = cur.stats()
                cells_done[0] += stc.cells_forward + stc.cells_backward
                pairs_done[0] += int(cur.results_view(first_buf)[1][sizes[q]])
                events_done[0] += float(sizes[q]) * ev
                reads_done[0] += sizes[q]
                cur.close()
            return
        for q in qs:
            cur = sa.Batch(pm, params, array_of(q), ambig=None, device=device, flags=xflags)
            stc = cur.stats()
            cells_done[0] += stc.cells_forward + stc.cells_backward
            events_done[0] += float(sizes[q]) * ev
            reads_done[0] += sizes[q]
            cur.start()
            flying.append((cur, q))
            if len(flying) >= depth:
                retire(flying.pop(0))
        for item in flying:
            retire(item)

    # before the job: the caching allocators see the pipeline's working set (a long-running aligner is in that state for good)
    run_slices([0] * (depth + 2 if depth > 1 else 2))
    # The job, R times inside one launch (one 1-second measurement is host jitter as much as anything): every
    # repetition is bracketed by the barrier on both sides; its wall is the MAX over ranks; a rank's idle time at the closing
    # barrier is that wall minus its own.  value = the job's cell updates / the MEDIAN repetition's wall; min and max beside it.
    reps = max(1, int(args.job_reps))
    rep_rows = []   # per repetition: [wall (max over ranks), per-rank own walls]
    cells_all = events_all = pairs_all = reads_all = 0.0
    for rep in range(reps):
        sync()
        cells_done[0] = events_done[0] = 0.0
        pairs_done[0] = reads_done[0] = 0
        t0 = time.perf_counter()
        run_slices(list(range(n_slices)))
        my_wall = time.perf_counter() - t0
        sync()
        dt = time.perf_counter() - t0
        own = [my_wall]
        cells_all, events_all, pairs_all, reads_all = cells_done[0], events_done[0], float(pairs_done[0]), float(reads_done[0])
        if dist is not None:
            import torch
            tdev = "cuda" if backend == "nccl" else "cpu"
            t = torch.tensor([dt], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            rows = torch.zeros(world, dtype=torch.float64, device=tdev)
            rows[rank] = my_wall
            dist.all_reduce(rows, op=dist.ReduceOp.SUM)
            own = [float(v) for v in rows.cpu().tolist()]
            tot = torch.tensor([cells_done[0], events_done[0], float(pairs_done[0]), float(reads_done[0])], dtype=torch.float64, device=tdev)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            cells_all, events_all, pairs_all, reads_all = [float(v) for v in tot.cpu().tolist()]
        rep_rows.append([dt, own])
    n_mine_all = [float(n_mine)]
    if dist is not None:
        import torch
        tdev = "cuda" if backend == "nccl" else "cpu"
        rows = torch.zeros(world, dtype=torch.float64, device=tdev)
        rows[rank] = float(n_mine)
        dist.all_reduce(rows, op=dist.ReduceOp.SUM)
        n_mine_all = [float(v) for v in rows.cpu().tolist()]
    if rank != 0:
        return None
    assert int(reads_all) == total, (reads_all, total)
    walls = sorted(r_[0] for r_ in rep_rows)
    med = walls[len(walls) // 2]
    med_row = min(rep_rows, key=lambda r_: abs(r_[0] - med))
    return {
        "workload": "BASELINE configs[4]: %d x %d-event reads over %d GPU%s, batches of <= %d" % (total, ev, world, "" if world == 1 else "s", sl),
        "scaling": "strong", "n_gpus": world, "total_reads": total, "events_per_read": ev,
        "value": cells_all / med, "unit": "cell_updates/s", "wall_s": med, "repetitions": reps,
        "wall_s_min": walls[0], "wall_s_max": walls[-1], "wall_spread": (walls[-1] - walls[0]) / med,
        "wall_s_all": [r_[0] for r_ in rep_rows],
        "events_per_s": events_all / med, "reads_per_s": reads_all / med, "pairs": pairs_all, "cell_updates": cells_all,
        "slice_reads": sizes[0], "slice_sizes_rank0": sorted(set(sizes), reverse=True), "slices_rank0": n_slices,
        "batches_in_flight": depth, "forward_storage_passes_per_slice": int(st0.n_chunks),
        "read_sets_cycled": n_sets, "reads_generated_per_rank": sum(len(s_) for s_ in sets),
        "inputs": "host-block" if in_block else "pageable", "host_threads_per_rank": os.environ.get("SA_HOST_THREADS"),
        # of the median repetition: a rank's own wall and what it waited at the closing barrier
        "per_rank": [{"rank": r_, "reads": int(n_mine_all[r_]), "wall_s": w_, "idle_at_barrier_s": med_row[0] - w_}
                     for r_, w_ in enumerate(med_row[1])],
        "partition_s": t_part, "generation_s": t_gen,
        "timed": "R repetitions of the whole job; each: barrier, every rank's batches through create / start / wait / results in "
                 "place / destroy, barrier; wall = max over ranks; value from the median repetition",
    }


LEG_COLUMNS = ["value", "ms_per_step", "roofline_frac", "frac_by_counters", "cpu_value", "kernels_only_value", "stage_ms",
               "dominant_kernel"]


def _r(x, nd=4):
    """numbers to nd significant digits (the line must stay inside the driver's 8 KB stdout tail)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def compact_line(out, full_path):
    """The ONE line of the contract, short: every BASELINE config with its fractions and CPU figure in `config.legs` (columns:
    LEG_COLUMNS) right in front of `roofline`, prose cut.  The complete record (every note, per-stage time and sample description)
    goes to `full_path` when that can be written (profiles/bench_r06_*.json are copies of it)."""
    c = dict(out.get("config") or {})
    if full_path:
        try:
            os.makedirs(os.path.dirname(full_path), exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(out, f)
        except OSError:
            full_path = None
    head = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    cfg = {"workload": str(c.get("workload", ""))[:140]}
    for k in ("reads_per_gpu", "events_per_read", "batches_in_flight", "events_per_s", "pairs_per_event", "result_record_bytes",
              "forward_storage_passes", "kernel_ms", "step_ms", "host_threads", "wall_s_whole_run"):
        if c.get(k) is not None:
            cfg[k] = c[k]
    if c.get("inputs"):
        cfg["inputs"] = str(c["inputs"]).split(" ")[0]
    if c.get("kernels_only_resident_inputs"):
        cfg["kernels_only_value"] = c["kernels_only_resident_inputs"]["value"]
    if c.get("regions_on_register_kernels"):
        cfg["regions_register_ring_strip"] = [c.get("regions_on_register_kernels"), c.get("regions_on_ring_kernels"),
                                              c.get("regions_on_strip_kernels")]
    if c.get("ring_kernels_lane_use") and "busy_lane_fraction" in c["ring_kernels_lane_use"]:
        cfg["ring_busy_lane_fraction"] = c["ring_kernels_lane_use"]["busy_lane_fraction"]
    if c.get("long_run"):
        cfg["long_run"] = {k: c["long_run"].get(k) for k in ("steps", "value", "ms_per_step", "step_ms")}
    job = c.get("scaling_job")
    if isinstance(job, dict):
        if "value" in job:
            cfg["scaling_job"] = {k: job.get(k) for k in ("value", "n_gpus", "total_reads", "events_per_read", "repetitions", "wall_s",
                                                          "wall_s_min", "wall_s_max", "wall_spread", "slice_sizes_rank0", "slices_rank0",
                                                          "batches_in_flight", "over_scaling_slice_value", "per_rank")}
            cfg["scaling_job"]["scaling"] = "strong"
        else:
            cfg["scaling_job"] = job
    sec = c.get("secondary")
    if isinstance(sec, dict):
        legs = {}
        for name, r in sec.items():
            if not isinstance(r, dict) or "value" not in r:
                legs[name] = ("skipped: " + str(r.get("skipped"))[:40]) if isinstance(r, dict) and r.get("skipped") else str(r)[:80]
                continue
            legs[name] = [r.get("value"), r.get("ms_per_step"), r.get("roofline_frac"), r.get("roofline_frac_by_counters"),
                          (r.get("cpu_baseline") or {}).get("value"), r.get("kernels_only_value"), r.get("stage_ms"),
                          r.get("dominant_kernel")]
        cfg["legs_columns"] = LEG_COLUMNS
        cfg["legs"] = legs   # (last key of config: right in front of `roofline`)
    if full_path:
        cfg["full_record"] = os.path.relpath(full_path, ROOT)
    head["config"] = _r(cfg)
    rf = out.get("roofline") or {}
    head["roofline"] = _r({k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_by_counters",
                                                  "traffic_over_algorithmic", "stage_ms", "launches_per_step", "avg_launch_ms",
                                                  "algorithmic_bytes_per_step", "bound_of_the_formula") if k in rf}, 5)
    meta = rf.get("counters_collected_at")
    if isinstance(meta, dict):
        head["roofline"]["counters_collected_at"] = {k: meta.get(k) for k in ("head", "tag", "collected") if k in meta}
    elif rf.get("source"):
        head["roofline"]["source"] = str(rf["source"])[:120]
    if out.get("issue_roofline"):
        ir = out["issue_roofline"]
        head["issue_roofline"] = _r({k: ir.get(k) for k in ("kernel", "frac", "valu_per_step", "instructions_per_step", "wave_wait_frac")})
    cb = out.get("cpu_baseline")
    if cb:
        head["cpu_baseline"] = _r({k: cb.get(k) for k in ("value", "unit", "cores", "kind", "one_thread_value",
                                                           "single_socket_linear_extrapolation", "socket_model",
                                                           "physical_cores_per_socket", "cgroup_cpu_quota") if k in cb})
        head["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    return json.dumps(head)


def main():
    # A streaming caller keeps three batches in flight, each with two compute streams, a copy stream and the upload stream: more
    # hardware queues than the runtime's default of four keep one batch's short kernels from queueing behind another's long
    # sweeps (12.7-12.9 against 12.6-13.9 ms per step; must be set before the first HIP call; ranks inherit it)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--in-flight", type=int, default=4, help="batches in flight in the timed pipeline (create of the next "
                                                            "overlaps the runs of the previous ones)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (default: 2000 = BASELINE configs[1]; "
                                                             "scaling: 12500 = configs[4]'s 100k reads / 8 GPUs)")
    ap.add_argument("--events", type=int, default=None, help="events per read (default 5000; scaling: 10000)")
    ap.add_argument("--workload", choices=["gaussian", "scaling", "scaling_job", "cpg", "hdp", "hdp_cpg", "hdp_realistic", "hdp_dense", "realistic", "event_align",
                                           "mea", "expectations", "expectations_cpg"],
                    default="gaussian",
                    help="gaussian = BASELINE configs[1] (the headline); scaling = configs[4]'s per-GPU slice (12500 "
                         "10k-event reads per GPU, several forward-storage passes); cpg = configs[2] (ACEGT model, every "
                         "CpG cytosine ambiguous C/E); hdp = configs[3] (HDP emissions); realistic = configs[1] reads with "
                         "the sparse anchors of a real guide alignment; event_align, mea = the steps either side of the "
                         "pair-HMM.")
    ap.add_argument("--threshold", type=float, default=None, help="posterior threshold (default 0.01; hdp: 0.1, what the "
                                                                  "reference's own HDP test uses, tests/stateMachineTests.c:912)")
    ap.add_argument("--event-stride", type=int, choices=[1, 4], default=4,
                    help="layout of the events a job hands over: 4 = the reference's NB_EVENT_PARAMS records (mean, noise, "
                         "duration, start: what signalMachine holds; the library gathers the means), 1 = a dense vector of means "
                         "(sa_job_t.event_stride; a quarter of the host memory traffic of sa_batch_create)")
    ap.add_argument("--inputs", choices=["auto", "pageable", "host-block"], default="auto",
                    help="host-block: the reads' event records and anchors are handed over in one page-locked block from "
                         "sa_host_alloc (SA_FLAG_INPUTS_IN_HOST_BLOCK): sent as they are, checked and packed by a kernel instead "
                         "of by host threads.  auto (default): host-block when the rank has at most three host threads "
                         "(SA_HOST_THREADS, set from the CPU quota for multi-rank runs), pageable otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernels-only", action="store_true", help="phase 1 only (sa_batch_run on one resident batch): for "
                                                                "profiler runs that count per-kernel launches")
    ap.add_argument("--cpg-every", type=int, default=1, help="--workload cpg: only every n-th CpG cytosine is ambiguous (sparse "
                                                                 "variant positions; default 1 = BASELINE configs[2])")
    ap.add_argument("--no-secondary", action="store_true", help="gaussian at N = 1 also measures `realistic` and `cpg` compactly "
                                                                "(config.secondary) and a long steady-state run (config.long_run); "
                                                                "this skips both")
    ap.add_argument("--long-steps", type=int, default=200, help="steps of the long steady-state sample reported beside the "
                                                                "K timed steps (config.long_run)")
    ap.add_argument("--cpu-reads-per-thread", type=int, default=30)
    ap.add_argument("--secondary-cpu-reads", type=int, default=6, help="reads per thread of the CPU baselines of the secondary legs")
    ap.add_argument("--job-reads", type=int, default=100000, help="config.scaling_job (BASELINE configs[4]): reads of the whole job")
    ap.add_argument("--job-events", type=int, default=10000)
    ap.add_argument("--job-slice", type=int, default=2000, help="reads per batch of the scaling job")
    ap.add_argument("--job-sets", type=int, default=2, help="distinct slice-sized read sets a rank generates and cycles")
    ap.add_argument("--full-record", default=os.path.join(ROOT, "gpurun_out", "bench_full_record.json"),
                    help="where the complete record goes (the printed line is its short form); '' = nowhere")
    ap.add_argument("--no-scaling-job", action="store_true", help="skip config.scaling_job")
    ap.add_argument("--job-reps", type=int, default=5, help="repetitions of the scaling job inside one launch (median, min, max reported)")
    ap.add_argument("--pairs8", action="store_true", help="SA_FLAG_PAIRS8: the batches hold 8-byte result records (x, y, probability) "
                                                          "instead of 16-byte ones -- workloads with one path per cell")
    ap.add_argument("--legs", default=None, help="comma-separated names of the config.secondary legs to run (default: all) -- for "
                                                 "looking at one leg in the context of the default run")
    ap.add_argument("--secondary-budget-s", type=float, default=330.0,
                    help="a secondary leg is not started once the run has taken this long (the default line must stay well inside "
                         "the driver's time limit)")
    args = ap.parse_args()
    if args.reads is None:
        # the sizes BASELINE.json names: configs[2] (cpg) 10 000 reads, configs[3] (hdp) 5000, configs[4]'s slice 12 500
        args.reads = {"scaling": 12500, "hdp": 5000, "hdp_dense": 5000, "hdp_cpg": 2000, "hdp_realistic": 2000, "cpg": 10000}.get(args.workload, 2000)
    if args.events is None:
        args.events = 10000 if args.workload == "scaling" else 5000
    if args.threshold is None:
        args.threshold = 0.1 if args.workload.startswith("hdp") else 0.01
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become one (nothing in this process has touched or will touch the GPU)
        sys.exit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    device = 0
    backend = os.environ.get("SA_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 path on a box with one GPU
    if world > 1:
        import torch
        import torch.distributed as dist
        device = local_rank % max(torch.cuda.device_count(), 1)  # identity on a full node
        n_dev = max(torch.cuda.device_count(), 1)
        ranks_per_device = (int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) + n_dev - 1) // n_dev
        torch.cuda.set_device(device)
        dist.init_process_group(backend=backend)  # RCCL; only used for the barrier and the max-over-ranks
    if world > 1:
        # Ranks of one node share the host (and, in a container, one CPU quota): the library sizes its host fan-out for a process
        # that has the machine to itself, so every rank gets its share here.  (SA_HOST_THREADS / SA_PLAN_THREADS: DESIGN.md.)
        info = host_cpu_info()
        cpus = int(info["cgroup_cpu_quota"]) if info["cgroup_cpu_quota"] else len(info["allowed"])
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        share = max(2, min(16, cpus // max(local_world, 1) - 2))
        os.environ.setdefault("SA_HOST_THREADS", str(share))
        os.environ.setdefault("SA_PLAN_THREADS", str(share))
    import signalalign_amd as sa
    from signalalign_amd import synth

    if args.workload in ("mea", "event_align", "expectations", "expectations_cpg"):
        # the two "next" rows are single-GPU side benchmarks: under a multi-rank launch only rank 0 runs them
        if world > 1:
            dist.barrier()
            if rank != 0:
                dist.destroy_process_group()
                return
        res = (bench_mea(args) if args.workload == "mea" else
               bench_expectations(args) if args.workload.startswith("expectations") else bench_event_align(args))
        if world > 1:
            dist.destroy_process_group()
        return res
    ctx = dict(dist=dist, rank=rank, world=world, device=device, backend=backend, ranks_per_device=ranks_per_device if world > 1 else 1,
               t_start=T_START)
    if args.workload == "scaling_job":   # the job alone: its own line (strong scaling)
        job = scaling_job(args, ctx)
        if job is not None:
            print(json.dumps({"metric": "dp_cell_updates_per_s", "value": job["value"], "unit": "cell_updates/s", "n_gpus": world,
                              "steps": job["slices_rank0"], "warmup": 0, "ms_per_step": job["wall_s"] / max(job["slices_rank0"], 1) * 1e3,
                              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": job}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    out = measure(args, ctx)
    if world > 1 and args.workload == "gaussian" and not args.kernels_only and not args.no_scaling_job:
        # BASELINE configs[4] on the multi-GPU line: the 100 000-read job, strong scaling, beside the weak-scaling headline
        sa.lib().sa_pool_release()
        try:
            job = scaling_job(args, ctx)
        except Exception as ex:   # (every rank raises or none: the job's collectives are behind the slices)
            job = {"failed": "%s: %s" % (type(ex).__name__, ex)}
        if out is not None:
            out["config"]["scaling_job"] = job
    if out is not None and world == 1 and args.workload == "gaussian" and not args.kernels_only and not args.no_secondary:
        # Every other BASELINE config and the two workloads beside them, compactly, on the same line: configs[2] (cpg) at its
        # 10 000 reads, configs[3] (hdp) at its 5000 reads and the reference's threshold 0.1 with the 0.01 figure beside it, one
        # GPU's slice of configs[4] (scaling), reads with the anchors of a real guide alignment (realistic) and the expectation
        # pass -- each with its own roofline fractions and its own CPU baseline (the restatement on a bounded sample of THAT
        # workload).  A leg that would start after --secondary-budget-s seconds of wall time is skipped and says so.
        import copy
        import signalalign_amd as sa
        sec = out["config"]["secondary"] = {}
        t_start = ctx["t_start"]

        only = None if not args.legs else set(args.legs.split(","))

        def leg(name, wl, reads, events, threshold, steps, warmup, cpu=True):
            if only is not None and name not in only:
                sec[name] = {"skipped": "--legs"}
                return None
            if time.perf_counter() - t_start > args.secondary_budget_s:
                sec[name] = {"skipped": "wall-time budget of the default run (%d s) reached" % args.secondary_budget_s}
                return None
            t_leg = time.perf_counter()
            a2 = copy.copy(args)
            a2.workload, a2.reads, a2.events, a2.threshold = wl, reads, events, threshold
            # (warm-up: the batches in flight plus two must have been through the caching allocators, which start empty -- the
            # blocks the previous workload left parked have other sizes)
            a2.steps, a2.warmup, a2.no_cpu_baseline, a2.cpu_reads_per_thread = steps, warmup, not cpu, args.secondary_cpu_reads
            # (device blocks only: page-locked blocks stay parked.  Measured: after an 8 GB page-locked block has been freed, a newly
            # pinned 5 GB block is filled by the copy engine at 30 instead of 57 GB/s -- the pages the runtime gets back are
            # scattered --, and the 8-byte-record leg behind the 16-byte one ran at 135 instead of 79 ms per step)
            sa.lib().sa_pool_release_device()
            try:
                r2 = measure(a2, ctx, compact=True)
            except Exception as ex:   # (a leg must not take the headline line down with it)
                sec[name] = {"failed": "%s: %s" % (type(ex).__name__, ex)}
                return None
            rf, cb = r2["roofline"], r2.get("cpu_baseline")
            sec[name] = {
                "workload": r2["config"]["workload"], "value": r2["value"], "ms_per_step": r2["ms_per_step"], "steps": r2["steps"],
                "median_ms_per_step_one_batch_at_a_time": r2.get("median_ms_per_step_one_batch_at_a_time"),
                # (one batch at a time: the same figure from the median step -- a single stalled step, e.g. a hipMalloc of tens of
                # GB that took seconds, moves `value` of a 4-step leg by an order of magnitude and this one not at all)
                "value_from_median_step": (None if not r2.get("median_ms_per_step_one_batch_at_a_time") else
                                           r2["value"] * r2["ms_per_step"] / r2["median_ms_per_step_one_batch_at_a_time"]),
                "events_per_s": r2["config"]["events_per_s"],
                "kernels_only_value": r2["config"]["kernels_only_resident_inputs"]["value"],
                "kernel_ms": r2["config"]["kernel_ms"], "pairs_per_event": r2["config"]["pairs_per_event"],
                "batches_in_flight": r2["config"]["batches_in_flight"], "read_sets_cycled": r2["config"]["read_sets_cycled"],
                "forward_storage_passes": r2["config"]["forward_storage_passes"],
                "ring_kernels_lane_use": r2["config"].get("ring_kernels_lane_use"),
                "dominant_kernel": rf["kernel"], "roofline_frac": rf["frac"], "roofline_frac_by_counters": rf["frac_by_counters"],
                "roofline_bound": rf["bound"], "stage_ms": rf["stage_ms"], "launches_per_step": rf["launches_per_step"],
                "issue_frac": (r2.get("issue_roofline") or {}).get("frac"),
                "cpu_baseline": None if cb is None else {"value": cb["value"], "cores": cb["cores"], "kind": cb["kind"],
                                                         "one_thread_value": cb["one_thread_value"], "sample": cb["sample"]},
                "gpu_over_cpu_baseline": None if cb is None else r2["value"] / cb["value"],
                "leg_wall_s": time.perf_counter() - t_leg}
            return r2
        ks = max(4, min(args.steps, 10))
        leg("realistic", "realistic", 2000, 5000, 0.01, ks, max(5, args.in_flight + 3))
        leg("cpg", "cpg", 10000, 5000, 0.01, 5, 2)
        if leg("hdp", "hdp", 5000, 5000, 0.1, ks, max(5, args.in_flight + 3)) is not None or (only and any(x_.startswith("hdp_threshold_0.01") for x_ in only)):
            r3 = leg("hdp_threshold_0.01", "hdp", 5000, 5000, 0.01, 3, 1, cpu=True)
            args_p8 = args.pairs8
            args.pairs8 = True
            r3b = leg("hdp_threshold_0.01_pairs8", "hdp", 5000, 5000, 0.01, 3, 1, cpu=True)
            args.pairs8 = args_p8
            if r3b is not None:
                sec["hdp_threshold_0.01_pairs8"]["note"] = ("the same step with SA_FLAG_PAIRS8: 8-byte records (x, y, probability; the "
                                                            "k-mer of a pair is the reference's at x), half the bytes over PCIe")
            if r3 is not None:
                sec["hdp_threshold_0.01"]["note"] = ("the bundled .nhdp is flat (every process: mean 59.8, sd 15.5 pA): 17.8 pairs "
                                                     "per event at 0.01, the step is their PCIe transfer")
        leg("hdp_dense", "hdp_dense", 5000, 5000, 0.1, ks, max(5, args.in_flight + 3))
        if only is not None and "expectations" not in only:
            sec["expectations"] = {"skipped": "--legs"}
        elif time.perf_counter() - t_start <= args.secondary_budget_s:
            t_leg = time.perf_counter()
            a3 = copy.copy(args)
            a3.workload, a3.reads, a3.events, a3.threshold, a3.steps, a3.warmup = "expectations", 2000, 5000, 0.01, ks, 2
            sa.lib().sa_pool_release()
            try:
                r4 = bench_expectations(a3, compact=True)
                rf4, cb4 = r4["roofline"], r4.get("cpu_baseline")
                sec["expectations"] = {"workload": r4["config"]["workload"], "value": r4["value"], "ms_per_step": r4["ms_per_step"],
                                       "steps": r4["steps"], "dominant_kernel": rf4["kernel"], "roofline_frac": rf4.get("frac"),
                                       "roofline_frac_by_counters": None, "stage_ms": rf4.get("stage_ms"),
                                       "roofline_source": rf4.get("source"),
                                       "cpu_baseline": None if cb4 is None else {"value": cb4["value"], "cores": cb4["cores"],
                                                                                 "kind": cb4["kind"], "sample": cb4["sample"]},
                                       "gpu_over_cpu_baseline": None if cb4 is None else r4["value"] / cb4["value"],
                                       "leg_wall_s": time.perf_counter() - t_leg}
            except Exception as ex:
                sec["expectations"] = {"failed": "%s: %s" % (type(ex).__name__, ex)}
        else:
            sec["expectations"] = {"skipped": "wall-time budget of the default run (%d s) reached" % args.secondary_budget_s}
        leg("scaling_slice", "scaling", 12500, 10000, 0.01, 4, 3)
        # ... and the whole configs[4] job at this N (what `--gpus N` reports as config.scaling_job): 100 000 reads in slices
        if args.no_scaling_job:
            out["config"]["scaling_job"] = {"skipped": "--no-scaling-job"}
        elif time.perf_counter() - t_start > args.secondary_budget_s:
            out["config"]["scaling_job"] = {"skipped": "wall-time budget of the default run (%d s) reached" % args.secondary_budget_s}
        else:
            t_leg = time.perf_counter()
            sa.lib().sa_pool_release()
            try:
                out["config"]["scaling_job"] = scaling_job(args, ctx)
                out["config"]["scaling_job"]["leg_wall_s"] = time.perf_counter() - t_leg
                sl_ = sec.get("scaling_slice") or {}
                if sl_.get("value"):
                    out["config"]["scaling_job"]["over_scaling_slice_value"] = out["config"]["scaling_job"]["value"] / sl_["value"]
                if sl_.get("value_from_median_step"):
                    out["config"]["scaling_job"]["over_scaling_slice_value_from_median_step"] = (
                        out["config"]["scaling_job"]["value"] / sl_["value_from_median_step"])
            except Exception as ex:
                out["config"]["scaling_job"] = {"failed": "%s: %s" % (type(ex).__name__, ex)}
        out["config"]["wall_s_whole_run"] = time.perf_counter() - t_start
    if out is not None:
        print(compact_line(out, args.full_record))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
