cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, env
  name=$1; shift; envs=$1; shift
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum"; do
    env $envs rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$name -- python $R/scripts/conv_layers.py 6e_7x1 wgrad 3 > /dev/null 2>$R/gpurun_out/pmc_$name.err || echo "failed: $grp"
  done
}
rm -rf $R/gpurun_out/pmc_*
run load "IFCBK_WGRAD_DEBUG=1"
run full "IFCBK_WGRAD_DEBUG=0"
