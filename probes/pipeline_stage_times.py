import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import signalalign_amd as sa
from signalalign_amd import synth
ROOT=os.getcwd()
gold=os.path.join(ROOT,"tests","golden","models")
mp=os.path.join(gold,"testModelR9.4_450bps.nucleotide.6mer.template.model")
alpha,k,t10,tab=synth.parse_model_table(mp)
pm=sa.Model.load(mp)
params=sa.default_params(threshold=0.01, expansion=50, trace_back=100)
sets=[[synth.make_read(i+q*2000,5000,alpha,k,tab) for i in range(2000)] for q in range(3)]
arrays=[sa.JobArray(js) for js in sets]
depth=int(sys.argv[1]) if len(sys.argv)>1 else 3
flying=[]; f=[];b_=[];tot=[]
t0=None
for s in range(30):
    if s==8: t0=time.perf_counter()
    cur=sa.Batch(pm,params,arrays[s%3],device=0)
    cur.start(); flying.append(cur)
    if len(flying)>=depth:
        old=flying.pop(0); old.wait(); st=old.stats()
        if s>=8: f.append(st.ms_forward); b_.append(st.ms_backward); tot.append(st.ms_total_device)
        old.n_pairs(0); old.close()
for old in flying:
    old.wait(); old.close()
dt=(time.perf_counter()-t0)/22
print("depth",depth,"ms/step %.2f"%(dt*1e3),"fwd %.2f bwd %.2f total_device %.2f"%(np.mean(f),np.mean(b_),np.mean(tot)))
