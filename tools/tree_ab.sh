#!/bin/bash
# Whole-tree A/B on ONE box: the previous round's tree (built beside HEAD under _ab/<tag>/, see below) against this tree — step, 4 clips per
# forward, VAE decode, cfg 4 — alternating, so that a regression no switch-level A/B can see (round 4: an allocator change that cost the VAE
# decode 12 %) shows up before the round ends.  Run through gpurun from the repo root:
#     bash tools/tree_ab.sh r4 [reps]
# Prepare the other tree HERE (CPU box) first; _ab/ is git-ignored but travels to the GPU box:
#     mkdir -p _ab/r4 && git archive <previous round's commit> | tar -x -C _ab/r4 && (cd _ab/r4 && python -m asva_amd.build && rm -rf asva_amd/csrc/_obj*)
# Each tree runs its OWN bench.py / tools with its OWN library and tile table.
TAG=${1:-r4}
REPS=${2:-3}
cd "$GRAFT_REPO_ROOT" || exit 1
OLD=$GRAFT_REPO_ROOT/_ab/$TAG
[ -f $OLD/asva_amd/libavsd_hip.so ] || { echo "no built tree under _ab/$TAG"; exit 1; }
export TMPDIR=/tmp
row() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d.get('batched',{}); v=d.get('vae_decode',{})
print('$1 step %.2f steps/s (%.3f ms)   4 clips %s clip-steps/s   VAE %s clips/s' % (d['value'], d['ms_per_step'], b.get('value'), v.get('clips_per_s')))"; }
for i in $(seq $REPS); do
  for t in new old; do
    if [ $t = old ]; then D=$OLD; else D=$GRAFT_REPO_ROOT; fi
    (cd $D && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-precise --also-clips 4 2>/dev/null | row "$t($( [ $t = old ] && echo $TAG || echo HEAD))")
  done
done
for t in new old; do
  if [ $t = old ]; then D=$OLD; else D=$GRAFT_REPO_ROOT; fi
  (cd $D && python tools/cfg4_run.py 2>/dev/null | tail -1 | sed "s/^/$t cfg4: /")
done
