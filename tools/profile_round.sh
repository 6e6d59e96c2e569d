#!/bin/bash
# Everything profiles/<tag>_* is made from, in two calls on the GPU box (a call is limited to 20 minutes):
#   bash tools/profile_round.sh r03 a    the default bench, its kernel trace, the counter passes
#   bash tools/profile_round.sh r03 b    the pose-graph back-end, the multi-GPU legs, the other configs, the marker trace
# (then, back in the build container: python tools/make_profiles.py r03)
set -e
tag=$1
part=${2:-ab}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
if [[ $part == *a* ]]; then
python bench.py > $O/${tag}_bench_final.log 2> $O/${tag}_bench_final.err
echo "bench done: $(tail -c 300 $O/${tag}_bench_final.log | head -c 120)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- python3 $R/bench.py --timed-only > $O/${tag}_prof_f.log 2>&1
cp $(ls /tmp/prof_f/*/*kernel_stats.csv | head -1) $O/${tag}_f_kernel_stats.csv
echo "kernel trace done"
bash $R/tools/pmc_run.sh $tag
fi
if [[ $part != *b* ]]; then exit 0; fi
# pose-graph back-end: configs[2] and the 4K / 200-tag shape, kernel trace and the MFMA counters of the 4K run
cd $R
python tools/run_config3.py > $O/${tag}_config3.json 2> /dev/null
python tools/run_config3.py --width 3840 --height 2160 --tags 200 --frames 24 > $O/${tag}_config5.json 2> /dev/null
cd /tmp
rm -rf /tmp/prof_gn /tmp/pmc_gn
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gn -- python3 $R/tools/run_config3.py --width 3840 --height 2160 --tags 200 --frames 24 > /dev/null 2>&1
cp $(ls /tmp/prof_gn/*/*kernel_stats.csv | head -1) $O/${tag}_gn_kernel_stats.csv
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_gn -- python3 $R/tools/run_config3.py --width 3840 --height 2160 --tags 200 --frames 24 > $O/${tag}_pmc_gn.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmc_gn $O/${tag}_pmc_gn.csv
echo "gn done"
# the multi-GPU step: the exchange at N = 1 (RCCL, the collective degenerates to a copy) and a 2-rank gloo rehearsal on this one GPU
cd $R
python bench.py --exchange --timed-only > $O/${tag}_exchange_n1.json 2> /dev/null
python bench.py --gpus 2 --rehearse --steps 8 --warmup 3 --gn-every 4 --no-cpu-baseline > $O/${tag}_rehearse_gpus2.json 2> /dev/null
# the other BASELINE.json configs through the same harness (graph update and pose-graph LM inside the timed region)
python bench.py --workload configs2 --timed-only > $O/${tag}_bench_configs2.json 2> /dev/null
python bench.py --workload configs4 --timed-only > $O/${tag}_bench_configs4_n1.json 2> /dev/null
python bench.py --workload configs4 --gpus 2 --rehearse --steps 6 --warmup 2 --batch 64 --no-cpu-baseline > $O/${tag}_bench_configs4_rehearse_gpus2.json 2> /dev/null
echo "multi-gpu legs done"
# ROCTX ranges of the stage groups (host-side enqueue spans) next to the kernel trace, one short run
cd /tmp
rm -rf /tmp/prof_mk
# (one rank under a process group, so that the exchange is a real RCCL call with its ROCTX range around it)
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 ASL_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --output-format csv -d /tmp/prof_mk -- python3 $R/bench.py --exchange --timed-only --steps 2 --warmup 1 --pipeline 1 > $O/${tag}_prof_mk.log 2>&1 || true
ls /tmp/prof_mk/*/ > $O/${tag}_prof_mk_files.txt 2>&1 || true
cp $(ls /tmp/prof_mk/*/*marker_api_trace.csv 2>/dev/null | head -1) $O/${tag}_marker_trace.csv 2>/dev/null || true
echo "marker trace done"
