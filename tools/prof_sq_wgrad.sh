#!/bin/bash
# usage (GPU box): tools/prof_sq_wgrad.sh  -> gpurun_out/r05_sq_wgrad/{a,b}: two --pmc passes of 8 SQ counters each over tools/probe_wgrad_sparse.py
# (dense / sparse-from-dout / sparse-from-pooled launches of conv4's weight gradient); tools/pmc_sq_wgrad.py turns them into profiles/r05_wgrad_sq.json
set -u
cd $GRAFT_REPO_ROOT
bash tools/prof_pmc_any.sh r05_sq_wgrad/a tools/probe_wgrad_sparse.py "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
bash tools/prof_pmc_any.sh r05_sq_wgrad/b tools/probe_wgrad_sparse.py "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE"
python tools/pmc_sq_wgrad.py gpurun_out/r05_sq_wgrad profiles/r05_wgrad_sq.json
cp profiles/r05_wgrad_sq.json gpurun_out/r05_sq_wgrad/
