#!/bin/bash
# Development aid: SQ counters of the lane kernel's general forms (tools/probes/laneh_probe.py under rocprofv3 --pmc, one pass per set)
export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/lh_pmc_$name -o pmc --output-format csv -- python3 tools/probes/laneh_probe.py > gpurun_out/lh_pmc_$name.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/lh_pmc_$name | grep -A 12 "wfa_lane_kernel"
  rm -rf gpurun_out/lh_pmc_$name
}
pass sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
