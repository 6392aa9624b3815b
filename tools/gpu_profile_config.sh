#!/bin/bash
# rocprofv3 passes for ONE configuration of tools/gpu_perf.py (development aid; run on the GPU box through gpurun):
#   bash tools/gpu_profile_config.sh <tag> <config>       e.g.  r03 C3      -> gpurun_out/<tag>_<config>_*
# --kernel-trace --stats, the two SQ passes, FETCH_SIZE and WRITE_SIZE, each its own run (--pmc only with --kernel-trace), the
# program directly after `--`.  tools/make_profiles.py turns the outputs into the files kept under profiles/.
TAG=${1:-r06}; CFG=${2:-C3}
OUT=gpurun_out
P=$OUT/${TAG}_${CFG}
mkdir -p $OUT
export TMPDIR=/tmp NO_CPU=1 BRIEF=1
python3 -c "import bench; print(bench.kernel_source_hash())" > $OUT/${TAG}_kernel_source_hash.txt
pass() {  # name, counters...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d ${P}_pmc_$name -o pmc --output-format csv -- python3 tools/gpu_perf.py $CFG > ${P}_pmc_$name.log 2>&1
  python3 tools/pmc_summary.py ${P}_pmc_$name > ${P}_pmc_$name.txt 2>&1
}
timeout 900 rocprofv3 --kernel-trace --stats -d ${P}_stats -o stats --output-format csv -- python3 tools/gpu_perf.py $CFG > ${P}_stats.log 2>&1
find ${P}_stats -name "*kernel_stats.csv" -exec cp {} ${P}_kernel_stats.csv \;
pass sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
pass fetch FETCH_SIZE
pass write WRITE_SIZE
# (which of the L2's fabric requests go to the memory controller?  TCC_EA0_RDREQ_DRAM counts the requests routed to
# DRAM — the Infinity Cache sits behind that interface, memory-side, so its hits are not told apart here; kept for the record)
if [ "$CFG" = "C3x8k" ]; then pass dram TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum; fi
# keep the merged-back volume small: the per-dispatch CSVs are summarised above
find $OUT -name "*.csv" -size +4M -delete
rm -rf ${P}_stats
