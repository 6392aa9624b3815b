#!/bin/bash
# fixed cost of a run by stage (development aid): tools/probes/midbatch_probe.py with stages switched off
cd "$(dirname "$0")/../.."
for v in "" "WFA_HIP_NO_BAND=1" "WFA_HIP_NO_BAND=1 WFA_HIP_NO_WIDE=1" "WFA_HIP_NO_BAND=1 WFA_HIP_NO_WIDE=1 WFA_HIP_FAST_STAGES=1" "WFA_HIP_NO_BAND=1 WFA_HIP_NO_WIDE=1 WFA_HIP_NO_FAST=1"; do
  echo "== $v"; env $v python3 tools/probes/midbatch_probe.py 2>&1 | grep "n=   65536"
done
