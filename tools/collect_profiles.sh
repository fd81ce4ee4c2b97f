#!/bin/bash
# GPU box: the rocprofv3 evidence of a round.  usage: tools/collect_profiles.sh <tag>   (writes under gpurun_out/prof_<tag>/)
# One program per rocprofv3 run; counters in their own passes (--pmc with --kernel-trace only).
set -u
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "[prof] bench under --stats" >> $out/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.json 2>> $out/log.txt
export TRAIN_ONLY=1
for B in 32 256; do
  echo "[prof] train-only stats B=$B" >> $out/log.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_b$B -- python3 tools/time_unet2d.py $B 10 >> $out/log.txt 2>&1
done
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  echo "[prof] pmc $c B=256" >> $out/log.txt
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 tools/time_unet2d.py 256 1 >> $out/log.txt 2>&1
done
echo "[prof] pmc LDS B=256" >> $out/log.txt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/pmc_LDS -- python3 tools/time_unet2d.py 256 1 >> $out/log.txt 2>&1
echo "[prof] done" >> $out/log.txt
find $out -name "*.csv" | head -40
