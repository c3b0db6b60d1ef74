#!/bin/bash
# SQ counter passes (instruction mix, issue / wait cycles, MFMA busy) for the three kernels the judge named: mlp_tile_kernel<true>,
# gemm_tn_split_kernel, kp1_step_kernel.  8 SQ slots per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots"); --pmc only ever with
# --kernel-trace; the program goes directly after `--`.
# Usage (GPU box): bash tools/sq_passes.sh <outdir-under-gpurun_out>         -> <out>/bench/<pass>/..., <out>/env/<pass>/... ; summarise with
#                  python3 tools/sq_summary.py gpurun_out/<outdir> profiles/r03
out=gpurun_out/$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"
P3="SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT"
P4="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH"
i=0
for c in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/bench/p$i -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $out/bench.p$i.log 2>&1 || { echo "bench pass $i FAILED"; tail -3 $out/bench.p$i.log; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/env/p$i -o pmc -- python3 tools/env_kernel_bench.py --launches 60 > $out/env.p$i.log 2>&1 || { echo "env pass $i FAILED"; tail -3 $out/env.p$i.log; exit 1; }
  echo "sq pass $i ok"
done
