set -x
cd /root/repo
python tools/export_plan.py --out /tmp/avsync15 --steps 50 2>&1 | tail -3
ls -la /tmp/avsync15 /tmp/avsync15/clip.plan.d | head -30
asva_amd/plan_host asva_amd/libavsd_hip.so /tmp/avsync15/clip.plan /tmp/avsync15/program.txt
python tools/export_plan.py --check /tmp/avsync15
PLAN_HOST_GRAPH=1 asva_amd/plan_host asva_amd/libavsd_hip.so /tmp/avsync15/clip.plan /tmp/avsync15/program.txt
python tools/export_plan.py --check /tmp/avsync15
