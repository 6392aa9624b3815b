#!/bin/bash
# The evidence of a round, in one gpurun call (development aid):  bash tools/gpu_final.sh <tag>
#   profiles (tools/gpu_profiles_all.sh), the bench line, the other configurations with the reference beside them, the GPU suite.
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
bash tools/gpu_profiles_all.sh $TAG > $OUT/${TAG}_all.log 2>&1
python3 bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python3 tools/gpu_perf.py C1 C1big E5 E10 A150 A150m A150l EF150 S150 X150 X150l X20 X20l C3 C3big C4a C4abig C4bench C4ax5 LEVf LINf INDf X100kb C3x8k C3xf8k C4x4k C3x20 X30k X30kf X100k B10k B1k BH10k C5 C5a8k C5as8k C3x8 C4x8 C3a8 M5 M5f LEV LIN MN1 MN1f C4am5 C3m5 C3xm5 > $OUT/${TAG}_other_configs.jsonl 2> $OUT/${TAG}_other_configs.err
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=25 > $OUT/${TAG}_gputests.log 2>&1
tail -30 $OUT/${TAG}_gputests.log; wc -c $OUT/${TAG}_bench.json; wc -l $OUT/${TAG}_other_configs.jsonl
