# SQ counter passes of one tools/gpu_perf.py configuration (development aid): bash tools/probes/tile_pmc.sh <config> [tag]
CFG=${1:-C4xs}; TAG=${2:-tile}
OUT=gpurun_out; P=$OUT/${TAG}_${CFG}
mkdir -p $OUT
export TMPDIR=/tmp NO_CPU=1 BRIEF=1
pass() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d ${P}_pmc_$name -o pmc --output-format csv -- python3 tools/gpu_perf.py $CFG > ${P}_pmc_$name.log 2>&1
  python3 tools/pmc_summary.py ${P}_pmc_$name > ${P}_pmc_$name.txt 2>&1
  grep -A12 "tile_kernel" ${P}_pmc_$name.txt
  rm -rf ${P}_pmc_$name
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH
pass c SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
