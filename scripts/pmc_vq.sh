cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_vq
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/a -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/one_vq_f32.py > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace -d $O/b -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/one_vq_f32.py > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
O="$O"
for d in ("a","b"):
    f=glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "vq_forward_kernel" in r["Kernel_Name"]:
            k=r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d,k,{c: round(sum(x)/len(x)) for c,x in v.items()}, len(next(iter(v.values()))))
PY
