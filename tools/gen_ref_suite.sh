#!/bin/bash
# Regenerates tests/golden/ref_gpu_suite.tar.gz: the task graphs of the reference's OWN GPU test suite
# (/root/reference/unittests/test_gpu_bfv.py, test_gpu_ckks.py: 35 BFV + 28 CKKS graph shapes x every level x 3-4 parameter
# sets = 1252 task directories), produced by running those generator tests unmodified with the reference's pure-Python
# frontend.  Build container only (needs /root/reference); the output is data (mega_ag.json + task_signature.json per task)
# and is all that travels.  Node ids are random (custom_task.py:148-154): regenerating changes ids, not graphs.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
cat > $W/test_config.py <<PY
TEST_BASE_DIR = "$W/out"
CPU_OUTPUT_BASE_DIR = "$W/out/cpu_tests"
GPU_OUTPUT_BASE_DIR = "$W/out/gpu_tests"
FPGA_OUTPUT_BASE_DIR = "$W/out/fpga_tests/noc_config_16c_3"
PY
cd $W
PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=$W python -m pytest /root/reference/unittests/test_gpu_bfv.py /root/reference/unittests/test_gpu_ckks.py -q -p no:cacheprovider | tail -2
cd $W/out/gpu_tests
find . -type f ! -name mega_ag.json ! -name task_signature.json -delete
tar --sort=name --mtime='2026-01-01' --owner=0 --group=0 -czf $ROOT/tests/golden/ref_gpu_suite.tar.gz .
echo "$(find . -name mega_ag.json | wc -l) tasks -> tests/golden/ref_gpu_suite.tar.gz ($(stat -c %s $ROOT/tests/golden/ref_gpu_suite.tar.gz) bytes)"
rm -rf $W
