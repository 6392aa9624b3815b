#!/bin/bash
# Counter passes for the C2 step (development aid; run on the GPU box through gpurun):
#   bash tools/gpu_counters.sh <tag>      -> gpurun_out/<tag>_*
# Every rocprofv3 pass is its own run (--pmc only with --kernel-trace), the program directly after `--`.
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import bench; print(bench.kernel_source_hash())" > $OUT/${TAG}_kernel_source_hash.txt
hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o /tmp/issue_rate && timeout 120 /tmp/issue_rate > $OUT/${TAG}_issue_rate.txt 2>&1
rocprofv3 -L > $OUT/${TAG}_counters_list.txt 2>&1
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs"
pass() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/${TAG}_pmc_$name -o pmc --output-format csv -- python3 $ARGS > $OUT/${TAG}_pmc_$name.log 2>&1
  python3 tools/pmc_summary.py $OUT/${TAG}_pmc_$name > $OUT/${TAG}_pmc_$name.txt 2>&1
}
pass sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
pass fetch FETCH_SIZE
pass write WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs > $OUT/${TAG}_stats.log 2>&1
find $OUT/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
# keep the merged-back volume small
find $OUT -name "*.csv" -size +8M -delete
