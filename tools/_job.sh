set -x
cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "copy_and or pack_kv" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_plan_gpu.py -x -q -m gpu -s 2>&1 | tail -40
