cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_patch2
rm -rf $O; mkdir -p $O
W=${1:-3x3}
i=0
for SET in "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" "TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace -d $O/p$i -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/conv_only.py $W 8 > $O/p$i.log 2>&1 || echo "pass $i failed: $SET"
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_patch2"
for d in sorted(glob.glob(O+"/p*")):
    if not os.path.isdir(d): continue
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not f: print(os.path.basename(d),"no counters:", open(d+".log").read()[-300:].replace("\n"," | ")); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "patch_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
