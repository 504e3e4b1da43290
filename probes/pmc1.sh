cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/pmc1","gpurun_out/pmc2"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in agg.items():
            if "fast" in k or "fold" in k: print(k, dict(v))
PY
