#!/bin/bash
# After tools/gpu_final.sh <tag> came back through gpurun: turn gpurun_out/<tag>_* into the files kept under profiles/ (development aid).
TAG=${1:-r06}
cp gpurun_out/${TAG}_lane_regions.txt profiles/${TAG}_lane_regions.txt
python3 tools/make_traffic.py gpurun_out $TAG > /dev/null
python3 tools/lane_mix.py > /dev/null
python3 tools/make_traffic.py gpurun_out $TAG | grep -E '"frac"|kernel_source_hash'
for p in "C1 C1" "C3 C3" "C4abig C4-adaptive" "C4x4k C4-exact" "C3x8k exact-10kb-score" "C3xf8k exact-10kb-full" "B10k BiWFA-10kb"; do set -- $p; python3 tools/make_profiles.py $TAG $1 $2 | grep -E "hbm_bytes_per_pair"; done
cp gpurun_out/${TAG}_other_configs.jsonl profiles/${TAG}_other_configs.jsonl
[ -f gpurun_out/${TAG}_C3x8k_pmc_dram.txt ] && cp gpurun_out/${TAG}_C3x8k_pmc_dram.txt profiles/${TAG}_C3x8k_pmc_dram.txt
[ -f gpurun_out/${TAG}_gputests.log ] && cp gpurun_out/${TAG}_gputests.log profiles/${TAG}_gputests.txt
[ -s gpurun_out/${TAG}_bench.json ] && cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
