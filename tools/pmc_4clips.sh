cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-vae --no-precise --also-clips 0 --clips-per-gpu 4"
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc4_$n -o p -- $BENCH --no-graph --steps 3 --warmup 1 > /tmp/pmc4_$n.log 2>&1)
done
M=$(find /tmp/pmc4_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db" | head -1)
F=$(find /tmp/pmc4_FETCH_SIZE -name "*.db" | head -1)
W=$(find /tmp/pmc4_WRITE_SIZE -name "*.db" | head -1)
python tools/pmc_step.py $M $F $W --skip 1 > gpurun_out/r4_pmc_step_4clips.md
cat gpurun_out/r4_pmc_step_4clips.md
