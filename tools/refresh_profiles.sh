#!/bin/bash
# Regenerates the round's measured evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh r2
# Writes gpurun_out/<tag>_*; big rocprofv3 databases stay under /tmp on the box.  Copy the summaries into profiles/.
set -x
TAG=${1:-r2}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
cut -c1-300 gpurun_out/${TAG}_bench_n1.json
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-vae --no-precise --also-clips 0"
# kernel trace of the graph-replayed run
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $BENCH --steps 60 --warmup 5 > /tmp/kt.log 2>&1)
KT=$(find /tmp/prof_kt -name "*.db" | head -1)
python tools/prof_summary.py $KT > gpurun_out/${TAG}_kernel_trace_bench.md
(cd tools && python step_timeline.py $KT --steps 50) > gpurun_out/${TAG}_step_timeline.md
head -20 gpurun_out/${TAG}_step_timeline.md
# PMC: three separate passes over eager steps
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$n -o p -- $BENCH --no-graph --steps 3 --warmup 1 > /tmp/pmc_$n.log 2>&1)
done
M=$(find /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db" | head -1)
F=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1)
W=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
python tools/pmc_step.py $M $F $W --skip 1 > gpurun_out/${TAG}_pmc_step.md
cat gpurun_out/${TAG}_pmc_step.md
python tools/pmc_traffic.py $F $W gpurun_out/pmc_traffic.json && cat gpurun_out/pmc_traffic.json
# every GEMM-family launch of the step against the vendor library on the same shape (yardstick only)
python tools/step_vs_blas.py 2>/dev/null > gpurun_out/${TAG}_step_vs_blas.txt
tail -1 gpurun_out/${TAG}_step_vs_blas.txt
# the in-tolerance mode (per-layer precision plan): kernel trace + steady-state timeline of the graph-replayed step
(cd /tmp && AVSD_PRECISION_PLAN=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_plan -o kt -- $BENCH --steps 40 --warmup 5 > /tmp/kt_plan.log 2>&1)
KP=$(find /tmp/prof_plan -name "*.db" | head -1)
python tools/prof_summary.py $KP > gpurun_out/${TAG}_plan_trace.md
(cd tools && python step_timeline.py $KP --steps 30) > gpurun_out/${TAG}_plan_timeline.md
head -12 gpurun_out/${TAG}_plan_timeline.md
# GroupNorm family: three PMC passes over tools/pmc_gn.py
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" \
         "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_gn_$i -o p -- python $GRAFT_REPO_ROOT/tools/pmc_gn.py > /tmp/pmc_gn_$i.log 2>&1)
  G=$(find /tmp/pmc_gn_$i -name "*.db" | head -1)
  (cd tools && python pmc_table.py $G gn_) > gpurun_out/${TAG}_pmc_gn_pass$i.txt
done
wc -l gpurun_out/${TAG}_pmc_gn_pass*.txt
