cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_patch
mkdir -p $O
for W in 3x3 s2; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/${W}_a -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/conv_only.py $W 12 > $O/${W}_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace -d $O/${W}_b -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/conv_only.py $W 12 > $O/${W}_b.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_patch"
for d in sorted(glob.glob(O+"/*_[ab]")):
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not f: print(d,"no counters"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "patch_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in agg.items()}, "launches", {k:len(v) for k,v in agg.items()}.get("SQ_WAVE_CYCLES") )
PY
