set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_large
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --kernel-trace -d "$O/$C" -o c -- python3 "$R/tools/large_pmc.py" > "$O/$C.log" 2>&1 || echo "$C failed"
done
python3 $R/tools/large_pmc.py --collect $O
