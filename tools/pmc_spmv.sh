cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_spmv
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 280 rocprofv3 --pmc $set -d gpurun_out/pmc_spmv/$tag -o r -- python tools/spmv_sweep.py 1024 > gpurun_out/pmc_spmv/$tag.log 2>&1
  db=$(find gpurun_out/pmc_spmv/$tag -name "*.db" | head -1)
  [ -n "$db" ] && python profiles/pmc_rocpd.py $db k_spmv_fused | head -8
done
