# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof2.log 2>&1
python3 - <<'PY'
import csv,glob,collections,json
out={}
for d in ("gpurun_out/pmc_fetch","gpurun_out/pmc_write"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        for k,v in agg.items():
            for c,val in v.items():
                out.setdefault(k,{})[c]={"sum":val,"launches":cnt[(k,c)]}
print(json.dumps(out,indent=1))
json.dump(out,open("gpurun_out/pmc_traffic_raw.json","w"),indent=1)
PY
find gpurun_out/prof2 -name "*kernel_stats.csv" | head -1 | xargs cat
