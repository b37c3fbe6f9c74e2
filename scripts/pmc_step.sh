# SQ counters of every kernel of the bf16 training step (two passes of 8 counters): bash scripts/pmc_step.sh <out-name>
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
rm -rf $O; mkdir -p $O
ARGS="--dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-second-mode --no-other-configs --no-kernel-timer"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/a -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/b -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $O/b.log 2>&1
python3 - <<PY
import csv, glob, os, collections
O="$O"
tab=collections.defaultdict(dict)
for d in ("a","b"):
    f=glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True)
    if not f: print(d,"no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        for c,vals in v.items(): tab[k][c]=sum(vals)/len(vals)
        tab[k]["launches"]=len(next(iter(v.values())))
with open(O+"/summary.csv","w") as f:
    cols=["launches","GRBM_GUI_ACTIVE","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_VALU_MFMA_BUSY_CYCLES","SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_INSTS_SALU","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE"]
    f.write("kernel,"+",".join(cols)+"\n")
    for k,v in sorted(tab.items(), key=lambda kv:-kv[1].get("GRBM_GUI_ACTIVE",0)*kv[1].get("launches",0)):
        f.write('"'+k+'",'+",".join(str(round(v.get(c,0))) for c in cols)+"\n")
print(open(O+"/summary.csv").read()[:6000])
PY
