#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04x; mkdir -p $O
bash tools/pmc_once.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" --config C3 > $O/pmc1.log 2>&1; tail -2 $O/pmc1.log
bash tools/pmc_once.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" --config C3 > $O/pmc2.log 2>&1; tail -2 $O/pmc2.log
bash tools/pmc_once.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" --config C3 --debug-flags 512 > $O/pmc3.log 2>&1; tail -2 $O/pmc3.log
bash tools/pmc_once.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" --config C3 --debug-flags 512 > $O/pmc4.log 2>&1; tail -2 $O/pmc4.log
