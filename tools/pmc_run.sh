#!/bin/bash
# Counter passes over the default bench (B = 1024, one synchronous launch per kernel per step).  Run on the GPU box:
#   bash tools/pmc_run.sh <tag> [passes]   -> gpurun_out/pmc_<tag>_{fetch,write,sq1,sq2[,sq3]}.csv (per-kernel averages, tools/pmc_summary.py)
# Counters are collected in their own runs (no trace domains), as the pool requires.
set -e
tag=$1
want=${2:-"fetch write sq1 sq2 sq3"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" \
            "sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "sq3 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VALU" \
            "sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  set -- $pass
  name=$1; shift
  case " $want " in *" $name "*) ;; *) continue ;; esac
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --pipeline 1 --timed-only > $R/gpurun_out/pmc_${tag}_$name.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_$name $R/gpurun_out/pmc_${tag}_$name.csv
  echo "pass $name done"
done
