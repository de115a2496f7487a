#!/bin/bash
# Round 6 evidence set on ONE box:  bash scripts/evidence_r06.sh r06z     (outputs under gpurun_out/$TAG*)
#   profile_round.sh (default bench, rocprofv3 stats of the same command, PMC passes of one chunk), hash-grid PMC passes (float2 /
#   half2 table), rocprofv3 kernel stats of the training step per route and loss, the kernel trace of one captured replay, counter
#   passes over the training step with the default planes and with MNRF_DW_PLANES_HALF=1, the GPU suite.
TAG=${1:-r06z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash scripts/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
PMC_OUT=gpurun_out/${TAG}_pmc_tcnn bash scripts/pmc_tcnn.sh > gpurun_out/${TAG}_pmc_tcnn.log 2>&1
MNRF_TCNN_TABLE_F16=1 PMC_OUT=gpurun_out/${TAG}_pmc_tcnn_f16 bash scripts/pmc_tcnn.sh > gpurun_out/${TAG}_pmc_tcnn_f16.log 2>&1
bash scripts/prof_train_routes.sh ${TAG}_train color_mask > /dev/null 2>&1
ROUTES=graph bash scripts/prof_train_routes.sh ${TAG}_train total > /dev/null 2>&1
bash scripts/trace_train_graph.sh ${TAG}_trace > /dev/null 2>&1
PMC_OUT=gpurun_out/${TAG}_pmc_train bash scripts/pmc_train.sh > /dev/null 2>&1
MNRF_DW_PLANES_HALF=1 PMC_OUT=gpurun_out/${TAG}_pmc_train_half bash scripts/pmc_train.sh > /dev/null 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${TAG}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/${TAG}_gpu_tests.log 2>&1
find gpurun_out/${TAG}* -name "*.csv" -size +2M -delete
find gpurun_out/${TAG}* -name "*.db" -delete
du -sh gpurun_out/${TAG}* | tail -20
python scripts/bench_summary.py gpurun_out/$TAG/bench.json
tail -3 gpurun_out/${TAG}_gpu_tests.log
